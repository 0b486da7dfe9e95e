#!/usr/bin/env python3
"""Diagnostic: where the cycles of a werewolf turn go (GE_STAMPS build of libge_step.so, see ge_device.h).
   cp game_engine_amd/ab/stamps.so game_engine_amd/libge_step.so; python tools/stamps.py [rooms]
Segments per wave-turn (s_memtime ticks = shader cycles): 0 = [results in registers .. next turn's row in registers]
(merge, phase decision, effects, restart, condition), 1 = [.. first queue slot in registers] (scan, ctx, slot writes,
shadow work, first LDS round trip), 2 = [.. results in registers] (queue rounds, atomics, second round trip)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "stamps.jsonl")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["GE_STAMPS_OUT"] = out
from game_engine_amd import GameTable, RoomBatch
rooms = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dsl = json.load(open(os.path.join(ROOT, "tests", "golden", "dsl", "werewolf-(mafia).json"), encoding="utf-8"))
b = RoomBatch([(GameTable(dsl), 8, rooms)], seed=0xC0FFEE, max_fuse=1024, restart=True)
b.step(1024); b.sync()
b.set_timing(True); b.kernel_time(reset=True)
b.step(4096); b.sync()
ms, _ = b.kernel_time(reset=True)
b.close()
d = json.loads(open(out).read().strip().splitlines()[-1])
wt = d["wave_turns"]
print(f"rooms {rooms}: {ms * 1e3 / 4096:.3f} us/turn with stamps; cycles per wave-turn by segment:",
      [round(x / wt, 1) for x in d["seg"]], "sum", round(sum(d["seg"]) / wt, 1))
