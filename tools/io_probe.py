#!/usr/bin/env python3
"""Host-buffer round trips of the C ABI (ge_batch_write_rooms / ge_batch_read_rooms hand over host arrays of ge_room_view):
their cost alone, and the rate of a whole job that starts and ends in host memory - write every room, K turns, read every
room back - beside the device-resident rate bench.py reports (`value` never includes these copies).
python tools/io_probe.py [game:n:rooms ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from game_engine_amd import GameTable, RoomBatch

SHORT = {"ww": "werewolf-(mafia)", "tt": "two-truths-and-a-lie"}
def dsl(game):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        return json.load(f)

for spec in sys.argv[1:] or ["ww:8:65536", "ww:8:1048576", "ww:12:2097152", "tt:4:1048576"]:
    g, n, rooms = spec.split(":"); n, rooms = int(n), int(rooms)
    b = RoomBatch([(GameTable(dsl(SHORT[g])), n, rooms)], seed=0xC0FFEE, max_fuse=1024, restart=True)
    b.step(1024); b.sync()
    v = b.read_rooms(); b.write_rooms(0, v); b.sync()                        # warm: pinned staging, first-touch
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): v = b.read_rooms(out=v)
    t_r = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): b.write_rooms(0, v)
    b.sync(); t_w = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); b.summary(); t_s = time.perf_counter() - t0
    mb = v.nbytes / 1e6
    print(f"{spec:>16}: ge_room_view {v.itemsize} B/room ({mb:.1f} MB)  read_rooms {t_r * 1e3:7.2f} ms ({mb / t_r / 1e3:5.1f} GB/s)  "
          f"write_rooms {t_w * 1e3:7.2f} ms ({mb / t_w / 1e3:5.1f} GB/s)  summary {t_s * 1e3:.2f} ms", flush=True)
    for k in (1, 64, 1024):
        t0 = time.perf_counter()
        for _ in range(3):
            b.write_rooms(0, v); b.step(k); v = b.read_rooms(out=v)
        dt = (time.perf_counter() - t0) / 3
        b.set_timing(True); b.kernel_time(reset=True); b.step(k); b.sync(); ms, _ = b.kernel_time(reset=True); b.set_timing(False)
        print(f"{'':>16}  host -> {k:4d} turns -> host: {dt * 1e3:8.2f} ms = {rooms * k / dt:.3e} steps/s with the copies; "
              f"the {k} turns alone {ms:.3f} ms = {rooms * k / (ms * 1e-3):.3e} steps/s (device)", flush=True)
    b.close()
