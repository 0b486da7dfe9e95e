#!/usr/bin/env python3
"""Did the step launches of a rocprofv3 --kernel-trace run overlap?  Reads <dir>/**/*kernel_trace.csv, takes the ge_step_kernel
rows, and prints: launches, mean duration, and - over the busiest window of back-to-back launches (gaps < 50 us) - the span, the
sum of durations, sum / span (> 1 = launches ran concurrently) and the time two or more launches were in flight.
    python tools/trace_overlap.py <dir> [bytes_per_full_launch [threads_of_a_full_launch]]"""
import csv, glob, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p, newline="") as f:
        for r in csv.DictReader(f):
            if "ge_step_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), r["Queue_Id"] if "Queue_Id" in r else "?"))
rows.sort()
if not rows:
    raise SystemExit("no ge_step_kernel rows")
# windows of back-to-back launches
wins, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if r[0] - end > 50_000:
        wins.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
wins.append(cur)
w = max(wins, key=len)
span = max(r[1] for r in w) - w[0][0]
dur = sum(r[1] - r[0] for r in w)
ev = sorted([(r[0], 1) for r in w] + [(r[1], -1) for r in w])
depth = multi = 0
last = ev[0][0]
for t, d in ev:
    if depth >= 2:
        multi += t - last
    depth += d; last = t
grids = sorted({r[2] for r in w})
queues = sorted({r[3] for r in w})
print(f"{len(rows)} step launches; busiest window: {len(w)} launches, span {span / 1e3:.1f} us, sum of durations {dur / 1e3:.1f} us "
      f"(x{dur / span:.2f}), >= 2 in flight for {100 * multi / span:.1f} % of the span; mean duration {dur / len(w) / 1e3:.2f} us; grids {grids}; queues {queues}")
if len(sys.argv) > 2:
    full = int(sys.argv[3]) if len(sys.argv) > 3 else max(grids)
    units = sum(r[2] for r in w) / full                       # launches of the whole batch the window amounts to
    print(f"  = {units:.1f} whole-batch launches, {span / units / 1e3:.2f} us each, {float(sys.argv[2]) * units / span:.1f} GB/s = "
          f"{float(sys.argv[2]) * units / span / 80:.1f} % of 8 TB/s by kernel-trace timestamps")
