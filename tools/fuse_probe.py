import json, os, sys, time
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
tb = GameTable(json.load(open("tests/golden/dsl/werewolf-(mafia).json")))
for rooms in (65536, 1048576):
    for fuse in (32, 64, 128, 256, 1024):
        b = RoomBatch([(tb, 8, rooms)], seed=0xC0FFEE, max_fuse=fuse, restart=True)
        b.step(256); b.sync(); b.set_timing(True); b.kernel_time(reset=True)
        steps = 2048
        t0 = time.perf_counter(); b.step(steps); b.sync(); wall = time.perf_counter() - t0
        ms, launches = b.kernel_time(reset=True)
        print(f"rooms {rooms} fuse {fuse:5d}: {ms*1e3/steps:.3f} us/turn device, {wall*1e6/steps:.3f} us/turn wall, {rooms*steps/wall:.3e} steps/s wall", flush=True)
        b.close()
