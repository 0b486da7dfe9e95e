#!/usr/bin/env python3
"""Single-turn launches over time: device us per launch in consecutive groups of G launches (one replayed hipGraph of G launches per
HIP-event pair), after an untimed pre-roll - does a shape's per-launch time depend on where in the run the window lies?
    K1_PREROLL=256 K1_GROUP=16 K1_GROUPS=64 python tools/k1_series.py game:n:rooms[+game:n:rooms] ..."""
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
P, G, NG = (int(os.environ.get(k, d)) for k, d in (("K1_PREROLL", "256"), ("K1_GROUP", "16"), ("K1_GROUPS", "64")))
for spec in sys.argv[1:]:
    segs = []
    for part in spec.split("+"):
        g, n, r = part.split(":")
        segs.append((GameTable(dsl(SHORT[g])), int(n), int(r)))
    b = RoomBatch(segs, seed=0xC0FFEE, max_fuse=1, restart=True)
    done = 0
    while done < P:                                                # pre-roll in steps of at most 256 launches (one small graph, replayed)
        k = min(256, P - done); b.step(k, stream); done += k
    b.step(G, stream); b.sync()                                    # builds the G-launch graph
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NG)]
    for e0, e1 in ev:
        e0.record(); b.step(G, stream); e1.record()
    b.sync()
    us = [e0.elapsed_time(e1) * 1e3 / G for e0, e1 in ev]
    b.close()
    print(f"{spec} preroll={P} group={G}: mean {sum(us) / len(us):.3f} min {min(us):.3f} max {max(us):.3f} us/launch")
    print("   " + " ".join(f"{u:.2f}" for u in us), flush=True)
