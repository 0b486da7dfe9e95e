#!/usr/bin/env python3
"""Condenses a tools/abn.sh log: per shape, the median us/turn of each variant (all values) and its change against the first variant.  python tools/ab_summary.py <log>"""
import collections, re, statistics, sys
rows = collections.OrderedDict(); order = []; v = None
for ln in open(sys.argv[1]):
    if ln.startswith("== "):
        v = ln[3:].strip().split("/")[-1]
        if v not in order: order.append(v)
        continue
    m = re.match(r"\s*(\S+)\s+fuse=(\d+)\s+kernel\s+([0-9.]+) us/turn", ln)
    if m and v:
        rows.setdefault((m.group(1), int(m.group(2))), collections.defaultdict(list))[v].append(float(m.group(3)))
    elif ln.startswith("#") or "passed" in ln or "failed" in ln:
        print(ln.rstrip())
print(f"{'shape':>26} {'fuse':>5} | " + " | ".join(f"{o:>22}" for o in order))
for (shape, fuse), d in rows.items():
    base = statistics.median(d[order[0]])
    print(f"{shape:>26} {fuse:>5} | " + " | ".join((f"{statistics.median(d[o]):8.3f} {100 * (statistics.median(d[o]) / base - 1):+5.1f}% [{min(d[o]):.3f}]" if d.get(o) else "-").rjust(22) for o in order))
