#!/usr/bin/env python3
"""Quick device-time probe of the step kernels over batch shapes (not the bench contract).
python tools/perf_probe.py [game:n:rooms[+game:n:rooms...] ...]   ('+' joins the segments of one mixed batch)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from game_engine_amd import GameTable, RoomBatch

def dsl(game):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        return json.load(f)

SHORT = {"ww": "werewolf-(mafia)", "tt": "two-truths-and-a-lie"}
specs = sys.argv[1:] or ["ww:8:65536", "ww:8:1048576", "ww:12:2097152", "tt:4:1048576"]
for spec in specs:
    segs = []
    for part in spec.split("+"):
        g, n, r = part.split(":")
        segs.append((GameTable(dsl(SHORT[g])), int(n), int(r)))
    rooms = sum(r for _, _, r in segs)
    # PROBE_FUSE="1024:4096,64:1024": fuse:steps pairs (default: 64 turns per launch over 1 024 turns, and single-turn launches)
    pairs = [tuple(int(x) for x in p.split(":")) for p in os.environ.get("PROBE_FUSE", "64:1024,1:256").split(",")]
    for fuse, steps in pairs:
        b = RoomBatch(segs, seed=0xC0FFEE, max_fuse=fuse, restart=True)
        b.step(256); b.sync()
        b.set_timing(True); b.kernel_time(reset=True)
        t0 = time.perf_counter(); b.step(steps); b.sync(); wall = time.perf_counter() - t0
        ms, launches = b.kernel_time(reset=True)
        alg = 2 * sum(b.bytes_per_room(i) * r for i, (_, _, r) in enumerate(segs))
        per_turn_us = ms * 1e3 / steps
        print(f"{spec:>16} fuse={fuse:<3} kernel {per_turn_us:8.3f} us/turn  {rooms*steps/(ms*1e-3):.3e} steps/s (device)  "
              f"{rooms*steps/wall:.3e} (wall)  alg {alg/per_turn_us/1e3:8.1f} GB/s = {alg/per_turn_us/1e3/80:.1f}% of 8 TB/s", flush=True)
        b.close()
