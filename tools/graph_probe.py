import json, os, sys, time
sys.path.insert(0, os.getcwd())
from game_engine_amd import GameTable, RoomBatch
tb = GameTable(json.load(open("tests/golden/dsl/werewolf-(mafia).json")))
for rooms in (1, 4096, 65536, 1048576):
    for fuse in (1, 4):
        b = RoomBatch([(tb, 8, rooms)], seed=0xC0FFEE, max_fuse=fuse, restart=True)
        b.step(256); b.sync()
        t0 = time.perf_counter()
        for _ in range(8): b.step(256)
        b.sync(); wall = time.perf_counter() - t0
        print(f"graph={'off' if os.environ.get('GE_NO_GRAPH') else 'on '} rooms {rooms:8d} max_fuse {fuse}: {wall*1e6/2048:.2f} us/turn wall ({wall*1e6/2048*fuse:.2f} us/launch)", flush=True)
        b.close()
