#!/usr/bin/env python3
"""What the kernel event trace (GE_FLAG_TRACE: one 16-byte record per room and turn) costs:
device time per turn with and without it, and the trace bytes written per second.
python tools/trace_probe.py [game:n:rooms ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from game_engine_amd import GameTable, RoomBatch

SHORT = {"ww": "werewolf-(mafia)", "tt": "two-truths-and-a-lie"}
for spec in sys.argv[1:] or ["ww:8:65536", "ww:8:1048576", "tt:4:1048576"]:
    g, n, rooms = spec.split(":"); n, rooms = int(n), int(rooms)
    with open(os.path.join(ROOT, "tests", "golden", "dsl", SHORT[g] + ".json"), encoding="utf-8") as f:
        tb = GameTable(json.load(f))
    res = {}
    for trace in (False, True):
        with RoomBatch([(tb, n, rooms)], seed=0xC0FFEE, max_fuse=64, restart=True, trace=trace) as b:
            for _ in range(4):
                b.step(64)                       # (a traced step may not exceed max_fuse turns)
            b.sync()
            b.set_timing(True); b.kernel_time(reset=True)
            for _ in range(8):
                b.step(64)
            b.sync()
            ms, launches = b.kernel_time(reset=True)
            res[trace] = ms * 1e3 / (8 * 64)
    extra = res[True] / res[False] - 1.0
    print(f"{spec:>16}  no trace {res[False]:8.3f} us/turn   traced {res[True]:8.3f} us/turn  (+{100 * extra:.1f} %)   "
          f"trace writes {16 * rooms / res[True] / 1e3:7.1f} GB/s", flush=True)
