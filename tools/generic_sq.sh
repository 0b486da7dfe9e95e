#!/bin/bash
# SQ counters of the shipped game against the same rules in generic form (tools/generic_probe.py's `same`): where the GENERIC
# builds' extra time per turn goes.   tools/generic_sq.sh <ww|tt> <players> <rooms>
set -u
GAME=$1; N=$2; ROOMS=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/generic_one.py <<'PY'
import copy, json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from game_engine_amd import GameTable, RoomBatch
game, n, rooms, form = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
name = {"ww": "werewolf-(mafia)", "tt": "two-truths-and-a-lie"}[game]
dsl = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dsl", name + ".json"), encoding="utf-8"))
if form == "same":
    extra = " and player.selected_target_id >= 0" if game == "ww" else " and player.total_score >= 0"
    for ph in dsl["phases"].values():
        cc = ph.get("completion_criteria") or {}
        if cc.get("type") == "player_action":
            cc["target_players"]["condition"] += extra
with RoomBatch([(GameTable(dsl, 2 if game == "tt" else 1), n, rooms)], seed=0xC0FFEE, max_fuse=64, restart=True) as b:
    b.step(256); b.step(1024); b.sync()
PY
for form in shipped same; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
    OUT=gpurun_out/gsq_${GAME}${N}_${form}_$(echo $set | cut -c1-12 | tr ' ' _); rm -rf "$OUT"; mkdir -p "$OUT"
    rocprofv3 --pmc $set --output-format csv -d "$OUT" -- python3 /tmp/generic_one.py $GAME $N $ROOMS $form > "$OUT/out.txt" 2> "$OUT/err.txt" || echo "$form failed"
    python3 - "$OUT" $ROOMS $form <<'PY'
import csv, glob, sys, statistics
out, rooms, form = sys.argv[1], int(sys.argv[2]), sys.argv[3]
waves = (rooms + 63) // 64
fs = glob.glob(f"{out}/**/*_counter_collection.csv", recursive=True)
d = {}
for r in csv.DictReader(open(fs[0])) if fs else []:
    if "ge_step_kernel" not in r["Kernel_Name"]: continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), dur))
for k, v in sorted(d.items()):
    dmax = max(x[1] for x in v); big = [x for x in v if x[1] >= 0.5 * dmax]
    med = statistics.median(x[0] for x in big)
    print(f"{form:8s} {k:22s} per wave-turn {med / waves / 64:10.1f}   (launch {statistics.median(x[1] for x in big)/1e3:.1f} us, n={len(big)})")
PY
  done
done
