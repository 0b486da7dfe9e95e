#!/usr/bin/env python3
"""Turn cost while all rooms are still in the same phase (straight after a reset) against steady state, where the 64 rooms of a
wavefront sit in 64 different phases: the upper bound of what regrouping rooms by phase could save.  python tools/coherence_probe.py"""
import json, os, sys
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
tb = GameTable(json.load(open("tests/golden/dsl/werewolf-(mafia).json")))
for rooms in (65536, 1 << 20):
    b = RoomBatch([(tb, 8, rooms)], seed=0xC0FFEE, max_fuse=8, restart=True)
    b.set_timing(True); b.kernel_time(reset=True)
    out = []
    for i in range(40):
        b.step(8); b.sync(); ms, _ = b.kernel_time(reset=True); out.append(ms * 1e3 / 8)
    b.step(4096); b.sync(); b.kernel_time(reset=True)
    st = []
    for i in range(8):
        b.step(8); b.sync(); ms, _ = b.kernel_time(reset=True); st.append(ms * 1e3 / 8)
    print(rooms, "us/turn in launches of 8 turns from reset:", " ".join("%.2f" % x for x in out))
    print(rooms, "steady state:", " ".join("%.2f" % x for x in st), flush=True)
    b.close()
