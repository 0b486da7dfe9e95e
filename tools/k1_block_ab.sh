#!/bin/bash
# A/B of the single-turn launches' block size (GE_SINGLE_BLOCK = rooms per block of a large batch's single-turn launch), on one
# box, interleaved; first a parity check of every setting: 48 single-turn launches == 48 fused turns (summary checksum)
#   tools/k1_block_ab.sh "ww:8:1048576 ww:12:2097152 tt:4:1048576"
SHAPES=${1:-"ww:8:1048576 ww:12:2097152 tt:4:1048576"}
for bs in 256 512 1024; do
  GE_SINGLE_BLOCK=$bs timeout -k 10 300 python - <<'PY' || exit 1
import json, os, sys
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
def dsl(g): return json.load(open(f"tests/golden/dsl/{g}.json", encoding="utf-8"))
for g, n, r in (("werewolf-(mafia)", 8, 1048576 + 77), ("werewolf-(mafia)", 12, 600001), ("two-truths-and-a-lie", 4, 1048576), ("two-truths-and-a-lie", 9, 524289)):
    tb = GameTable(dsl(g))
    with RoomBatch([(tb, n, r)], seed=5, max_fuse=1, restart=True) as a, RoomBatch([(tb, n, r)], seed=5, max_fuse=48, restart=True) as f:
        a.step(48); f.step(48)
        sa, sf = a.summary(), f.summary()
        assert sa == sf, (g, n, sa, sf)
        assert a.read_rooms(r - 1000, 1000).tobytes() == f.read_rooms(r - 1000, 1000).tobytes()
print("parity ok at GE_SINGLE_BLOCK =", os.environ["GE_SINGLE_BLOCK"], flush=True)
PY
done
for rep in 1 2 3; do
  for bs in 256 512 1024; do
    GE_SINGLE_BLOCK=$bs timeout -k 10 300 python tools/k1_probe.py $SHAPES || exit 1
  done
done
