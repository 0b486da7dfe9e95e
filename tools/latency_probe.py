#!/usr/bin/env python3
"""Single-room (drop-in) latency: one turn of one traced room through the Python host, as the
room service does it — step(1), read the room view, read the turn's event, render the tool calls.
Prints the median and p95 per turn.  python tools/latency_probe.py [turns]"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from game_engine_amd import GameTable, RoomBatch
from game_engine_amd.toolcalls import turn_tool_calls

turns = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for game, n in (("werewolf-(mafia)", 8), ("two-truths-and-a-lie", 4)):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        tb = GameTable(json.load(f))
    with RoomBatch([(tb, n, 1)], seed=7, max_fuse=1, restart=True, trace=True) as b:
        before = b.read_rooms(0, 1)[0]
        step_t, full_t = [], []
        for t in range(turns):
            t0 = time.perf_counter()
            b.step(1); b.sync()
            t1 = time.perf_counter()
            after = b.read_rooms(0, 1)[0]
            ev = b.read_events(0, 1)[0][0]
            calls = turn_tool_calls(tb, before, after, ev) if not ev["restarted"] else []
            t2 = time.perf_counter()
            before = after
            if t >= 20:
                step_t.append((t1 - t0) * 1e6); full_t.append((t2 - t0) * 1e6)
        q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
        print(f"{game} x{n}: step+sync median {statistics.median(step_t):.1f} us (p95 {q(step_t, .95):.1f}); "
              f"step + read view + read event + render tool calls median {statistics.median(full_t):.1f} us (p95 {q(full_t, .95):.1f})")
