import sys, time, json
sys.path.insert(0, '.')
from game_engine_amd import GameTable, RoomBatch
dsl = json.load(open('tests/golden/dsl/werewolf-(mafia).json'))
for rooms in (65536, 1 << 20):
    b = RoomBatch([(GameTable(dsl), 8, rooms)], seed=1, restart=True)
    b.step(128); b.sync(); b.summary()
    t0 = time.perf_counter()
    for _ in range(20): s = b.summary()
    print(rooms, "summary ms", (time.perf_counter() - t0) / 20 * 1e3, s["finished"])
