#!/usr/bin/env python3
"""Single-turn launches (max_fuse = 1) in their sustained regime: device time of a replayed hipGraph of back-to-back launches
(HIP events around the replay on the launch stream), per launch - the figure bench.py's hbm_streaming block reports.
    python tools/k1_probe.py [game:n:rooms ...]         (launch knobs such as GE_SINGLE_BLOCK are read once per process)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from game_engine_amd import GameTable, RoomBatch

SHORT = {"ww": "werewolf-(mafia)", "tt": "two-truths-and-a-lie"}


def dsl(game):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        return json.load(f)


stream = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:] or ["ww:8:1048576", "ww:12:2097152", "tt:4:1048576"]:
    g, n, r = spec.split(":")
    n, r = int(n), int(r)
    launches = max(16, min(256, int(2e10 // (r * 64))))            # ~a few hundred ms of replays at most
    b = RoomBatch([(GameTable(dsl(SHORT[g])), n, r)], seed=0xC0FFEE, max_fuse=1, restart=True)
    bpr = b.bytes_per_room(0)
    b.step(256, stream); b.step(launches, stream); b.sync()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.step(launches, stream); e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / launches
        best = t if best is None else min(best, t)
    b.close()
    gbs = 2 * bpr * r / best / 1e3
    print(f"{spec:>16} block={os.environ.get('GE_SINGLE_BLOCK', 'default'):>7} {best:9.3f} us/launch sustained ({launches} per replay)  "
          f"{gbs:7.1f} GB/s = {gbs / 80:5.1f}% of 8 TB/s", flush=True)
