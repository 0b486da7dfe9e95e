#!/bin/bash
# A/B of launch chains for single-turn stepping (GE_CHAINS = number of independent launch chains a batch's blocks are cut into,
# parallel branches of the captured hipGraph; csrc/ge_step.hip graph_for), on one box, interleaved.  First a parity check of
# every setting: 48 single-turn launches == 48 fused turns (summary checksum + a window of rooms).
#   tools/chains_ab.sh "ww:8:1048576 ww:12:2097152 tt:4:1048576 ww:8:524288+tt:4:524288 ww:8:33554432"
# NOTE: GE_CHAINS existed at commit 3731afc only (the variant was measured and removed; profiles/r05_ab_launch_chains.txt).
SHAPES=${1:-"ww:8:1048576 ww:12:2097152 tt:4:1048576 ww:8:524288+tt:4:524288 ww:8:33554432"}
for ch in 1 2 4 8; do
  GE_CHAINS=$ch timeout -k 10 300 python - <<'PY' || exit 1
import json, os, sys
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
def dsl(g): return json.load(open(f"tests/golden/dsl/{g}.json", encoding="utf-8"))
ww, tt = GameTable(dsl("werewolf-(mafia)")), GameTable(dsl("two-truths-and-a-lie"))
for segs in ([(ww, 8, 1048576 + 77)], [(ww, 12, 600001)], [(tt, 4, 1048576)], [(tt, 9, 524289)], [(ww, 8, 300001), (tt, 4, 400000), (ww, 12, 200003)]):
    r = sum(x[2] for x in segs)
    with RoomBatch(segs, seed=5, max_fuse=1, restart=True) as a, RoomBatch(segs, seed=5, max_fuse=48, restart=True) as f:
        a.step(48); f.step(48); a.step(7); f.step(7)
        sa, sf = a.summary(), f.summary()
        assert sa == sf, (segs, sa, sf)
        assert a.read_rooms(r - 1000, 1000).tobytes() == f.read_rooms(r - 1000, 1000).tobytes()
        assert a.read_rooms(r // 2, 1000).tobytes() == f.read_rooms(r // 2, 1000).tobytes()
print("parity ok at GE_CHAINS =", os.environ["GE_CHAINS"], flush=True)
PY
done
for rep in 1 2 3; do
  for ch in 1 2 4 8; do
    GE_CHAINS=$ch timeout -k 10 300 python tools/k1_probe.py $SHAPES || exit 1
  done
done
