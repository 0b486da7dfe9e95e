#!/bin/bash
# SQ counter pass for the bench command at a given batch size: tools/sq_profile.sh <tag> <rooms>
set -u
TAG=$1; ROOMS=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$OUT/a" -- python3 bench.py --no-cpu-baseline --no-unfused --no-other-shapes --no-from-init --fuse 64 --warmup 64 --steps 512 --rooms $ROOMS > "$OUT/a.json" 2> "$OUT/a.err" || echo a failed
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/b" -- python3 bench.py --no-cpu-baseline --no-unfused --no-other-shapes --no-from-init --fuse 64 --warmup 64 --steps 512 --rooms $ROOMS > "$OUT/b.json" 2> "$OUT/b.err" || echo b failed
python3 - "$OUT" $ROOMS <<'PY'
import csv, glob, sys, statistics
out, rooms = sys.argv[1], int(sys.argv[2])
waves = (rooms + 63) // 64
for sub in ("a", "b"):
    fs = glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True)
    if not fs: print(sub, "no csv"); continue
    d = {}
    for r in csv.DictReader(open(fs[0])):
        if "ge_step_kernel" not in r["Kernel_Name"]: continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), dur))
    for k, v in sorted(d.items()):
        dmax = max(x[1] for x in v); big = [x for x in v if x[1] >= 0.5 * dmax]
        med = statistics.median(x[0] for x in big)
        print(f"{k:24s} per launch {med:14.0f}   per wave-turn {med / waves / 64:10.1f}   (launch {statistics.median(x[1] for x in big)/1e3:.1f} us, n={len(big)})")
PY
