#!/usr/bin/env python3
"""Single-turn launches (max_fuse = 1) in their sustained regime: device time of a replayed hipGraph of back-to-back launches
(HIP events around the replay on the launch stream), per launch - the figure bench.py's hbm_streaming block reports.
    python tools/k1_probe.py [game:n:rooms[+game:n:rooms] ...]     (launch knobs such as GE_SINGLE_BLOCK / GE_CHAINS are read once per
    process; `+` joins the segments of a mixed batch).  K1_PREROLL = untimed turns before the window (default 1 024, as in bench.py: Two-Truths rooms
    start in step and are still loosely in phase after 256 turns, where a mixed Werewolf + Two-Truths launch times 5 % faster than in steady state)"""
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
    segs = []
    for part in spec.split("+"):
        g, n, r = part.split(":")
        segs.append((GameTable(dsl(SHORT[g])), int(n), int(r)))
    rooms = sum(x[2] for x in segs)
    launches = max(16, min(256, int(2e10 // (rooms * 64))))        # ~a few hundred ms of replays at most
    b = RoomBatch(segs, seed=0xC0FFEE, max_fuse=1, restart=True)
    state = sum(b.bytes_per_room(k) * x[2] for k, x in enumerate(segs))
    b.step(int(os.environ.get('K1_PREROLL', '1024')), stream); b.step(launches, stream); b.sync()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.step(launches, stream); e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / launches
        best = t if best is None else min(best, t)
    b.close()
    gbs = 2 * state / best / 1e3
    print(f"{spec:>30} block={os.environ.get('GE_SINGLE_BLOCK', 'default'):>7} chains={os.environ.get('GE_CHAINS', 'default'):>7} "
          f"{best:9.3f} us/launch sustained ({launches} per replay)  {gbs:7.1f} GB/s = {gbs / 80:5.1f}% of 8 TB/s", flush=True)
