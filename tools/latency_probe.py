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

# the whole graph-run replacement: RoomService.continue_room = step + read-back + backend tool calls + log fold
# (playerActions / game_notes / phase_history) + the phase's frontend tool calls (ui_script)
from game_engine_amd import RoomService
for game, n in (("werewolf-(mafia)", 8), ("two-truths-and-a-lie", 4)):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        dsl = json.load(f)
    svc = RoomService(seed=7)
    svc.create_room("t", game, [{"name": f"Player {i + 1}"} for i in range(n)], dsl=dsl)
    ts = []
    for t in range(turns):
        t0 = time.perf_counter()
        out = svc.continue_room("t")
        ts.append((time.perf_counter() - t0) * 1e6)
        if out["state"].get("end_turn", -1) >= 0 and t < turns - 1:          # next game on a fresh thread
            svc.close("t")
            svc.create_room("t", game, [{"name": f"Player {i + 1}"} for i in range(n)], dsl=dsl)
    ts = ts[20:]
    print(f"{game} x{n}: RoomService.continue_room (state + toolCalls + uiCalls) median {statistics.median(ts):.1f} us "
          f"(p95 {sorted(ts)[int(.95 * (len(ts) - 1))]:.1f})")
    svc.close("t")
