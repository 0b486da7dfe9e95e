#!/usr/bin/env python3
"""Lone wavefronts holding 33 - 63 rooms (single-game batches of 32 769 - 65 535 rooms, fused launches): every room against the oracle, and device us per turn
against GE_HALF_WAVES=0 (64 rooms per wavefront).  python tools/rpb_check.py [check|time]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_dsl
from game_engine_amd import GameTable, RoomBatch
from oracle.oracle import Oracle
from parity_util import assert_views_equal, oracle_rooms_as_views

mode = sys.argv[1] if len(sys.argv) > 1 else "check"
for game, n in (("werewolf-(mafia)", 8), ("werewolf-(mafia)", 12), ("two-truths-and-a-lie", 4), ("two-truths-and-a-lie", 7)):
    dsl = load_dsl(game)
    for R in (32769, 40000, 49152, 57000, 65535):
        tb = GameTable(dsl)
        if mode == "check":
            orc = Oracle(dsl, n); st = orc.init_rooms(R)
            with RoomBatch([(tb, n, R)], seed=1234567, first_room=77, max_fuse=64, restart=True) as b:
                done = 0
                for turns in (1, 64, 70, 33):
                    b.step(turns); orc.run(st, 1234567, 77, done, turns, threads=0, restart=True); done += turns
                    assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, st), f"{game} x{n} R={R} after {done}")
            print(f"ok {game} x{n} rooms={R}", flush=True)
        else:
            with RoomBatch([(tb, n, R)], seed=0xC0FFEE, max_fuse=1024, restart=True) as b:
                b.step(1024); b.sync(); b.set_timing(True); b.kernel_time(reset=True)
                b.step(4096); b.sync(); ms, _ = b.kernel_time(reset=True)
            print(f"{game} x{n} rooms={R}: {ms * 1e3 / 4096:.3f} us/turn (GE_HALF_WAVES={os.environ.get('GE_HALF_WAVES', 'default')})", flush=True)
