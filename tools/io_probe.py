import json, os, sys, time
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
tb = GameTable(json.load(open("tests/golden/dsl/werewolf-(mafia).json")))
b = RoomBatch([(tb, 8, 1 << 20)], seed=1, restart=True)
b.step(64); b.sync()
for n in (1, 4096, 1 << 18, 1 << 20):
    t0 = time.perf_counter(); v = b.read_rooms(0, n); t1 = time.perf_counter()
    b.write_rooms(0, v); t2 = time.perf_counter()
    print(f"read_rooms({n}): {(t1-t0)*1e3:.2f} ms ({n/(t1-t0)/1e6:.2f} M rooms/s)   write_rooms: {(t2-t1)*1e3:.2f} ms", flush=True)
t0 = time.perf_counter(); s = b.summary(); print(f"summary: {(time.perf_counter()-t0)*1e3:.2f} ms")
