#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile.sh into small committed files:
   profiles/<tag>_kernel_stats.csv   (the --stats table)
   profiles/<tag>_counters.json      (per-launch medians for the step kernel, fused launches)
   profiles/pmc_traffic.json         (HBM bytes per fused launch, corrected as the microarch guide says:
                                      FETCH_SIZE is in KiB and reads HALF the bytes of a 16 B/lane
                                      coalesced stream on gfx950 -> x2; WRITE_SIZE is exact)"""
import csv, glob, json, os, shutil, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "profiles_out")
os.makedirs(prof, exist_ok=True)

def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None

ks = one("kt/**/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(prof, f"{tag}_kernel_stats.csv"))

def counters(sub):
    f = one(f"{sub}/**/*_counter_collection.csv")
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if "ge_step_kernel" not in r["Kernel_Name"]:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        out.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), dur))
    return out

res = {"tag": tag, "kernel": "ge_step_kernel", "note": "medians over the fused launches (dispatches lasting >= half the longest one)"}
allc = {}
for sub in ("fetch", "write", "sq"):
    allc.update(counters(sub))
for name, vals in allc.items():
    dmax = max(d for _, d in vals)
    big = [x for x in vals if x[1] >= 0.5 * dmax]                   # the fused launches (bench.py also runs a short un-fused probe)
    res[name] = {"median": statistics.median(v for v, _ in big), "n": len(big),
                 "median_duration_ns_profiled": statistics.median(d for _, d in big)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    fetch_b = res["FETCH_SIZE"]["median"] * 1024 * 2          # gfx950: x2 for wide coalesced reads
    write_b = res["WRITE_SIZE"]["median"] * 1024
    res["hbm_bytes_per_launch"] = {"read": fetch_b, "write": write_b, "total": fetch_b + write_b}
    with open(os.path.join(prof, "pmc_traffic.json"), "w") as f:
        # the launch's floor: every room record read once and written once (65 536 x 32 B for the default bench)
        floor = float(os.environ.get("GE_STATE_BYTES", 2 * 65536 * 32))
        json.dump({"tag": tag, "bytes_per_launch": fetch_b + write_b, "read": fetch_b, "write": write_b,
                   "state_bytes_read_plus_written": floor,
                   "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count for 16 B/lane streams), WRITE_SIZE KiB x1024"}, f)
for b in ("bench_kt.json",):
    p = os.path.join(src, b)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
with open(os.path.join(prof, f"{tag}_counters.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
