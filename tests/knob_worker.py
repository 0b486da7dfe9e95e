"""Child process of tests/test_gpu_knobs.py: the launch knobs (GE_LOWOCC_ROOMS, GE_BLOCK_THREADS) are read once per
process, so each setting needs a fresh one.  Runs one scenario against the oracle and exits 0, or raises."""
import sys

import numpy as np

from conftest import load_dsl
from game_engine_amd import GameTable, RoomBatch
from oracle.oracle import Oracle
from parity_util import assert_views_equal, oracle_batch, oracle_events, oracle_rooms_as_views


def trace(n_rooms):
    """the event trace, turn by turn, single-turn launches and fused ones, Werewolf x8 / x12 and Two-Truths x4"""
    for game, n, restart in (("werewolf-(mafia)", 8, True), ("werewolf-(mafia)", 12, True), ("two-truths-and-a-lie", 4, True)):
        dsl = load_dsl(game)
        seed, first = 11, 1 << 20
        orc = Oracle(dsl, n)
        rooms = orc.init_rooms(n_rooms)
        with RoomBatch([(GameTable(dsl), n, n_rooms)], seed=seed, first_room=first, max_fuse=16, restart=restart, trace=True) as b:
            t = 0
            for chunk in (1, 16, 7, 1, 16, 3, 16):
                b.step(chunk)
                ev = b.read_events()
                assert ev.shape == (n_rooms, chunk)
                for k in range(chunk):
                    orc.run(rooms, seed, first, t, 1, threads=0, restart=restart)
                    want = oracle_events(orc, rooms, t)
                    got = np.ascontiguousarray(ev[:, k])
                    for f in ("turn", "from_phase_id", "to_phase_id", "acted_now", "restarted", "choice"):
                        assert (got[f] == want[f]).all(), (game, n, f, t)
                    t += 1
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"{game} x{n} after the traced turns")


def humans(n_rooms):
    """host-driven seats (human_mask): the bot policy skips them, batched injection applies their actions"""
    for game, n, mask in (("werewolf-(mafia)", 8, 0b101), ("two-truths-and-a-lie", 4, 0b11)):
        dsl = load_dsl(game)
        orc = Oracle(dsl, n)
        seed, first = 33, 1 << 20
        rng = np.random.default_rng(n * 7 + mask)
        rooms = orc.init_rooms(n_rooms)
        seats = [i + 1 for i in range(n) if (mask >> i) & 1]
        applied = 0
        with RoomBatch([(GameTable(dsl), n, n_rooms, mask)], seed=seed, first_room=first, max_fuse=4) as b:
            t = 0
            for _ in range(12):
                k = 4000
                rr = rng.integers(0, n_rooms, size=k).astype(np.uint64)
                pl = rng.choice(seats, size=k).astype(np.uint32)
                ch = rng.integers(0, n + 2, size=k).astype(np.uint32)
                want = np.array([0 if orc.inject(rooms, int(r), int(p), int(c)) else -1 for r, p, c in zip(rr, pl, ch)], dtype=np.int32)
                got = b.inject_actions(rr, pl, ch)
                assert got.tolist() == want.tolist(), (game, t)
                applied += int((got == 0).sum())
                turns = 1 + t % 4                              # single-turn and fused launches
                b.step(turns)
                orc.run(rooms, seed, first, t, turns, threads=0, human_mask=mask)
                t += turns
                assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"{game} turn {t}")
        assert applied > 200


def mixed(n_rooms):
    """a mixed batch in steady state (restart): Werewolf x8 / x12 and Two-Truths x4 / x7 segments in the same launches"""
    ww, tt = load_dsl("werewolf-(mafia)"), load_dsl("two-truths-and-a-lie")
    segs = [(ww, 8, n_rooms), (tt, 4, n_rooms // 2 + 1), (ww, 12, n_rooms // 3 + 5), (tt, 7, n_rooms // 4 + 3)]
    seed, first, turns = 0xC0FFEE, 1 << 33, 90
    with RoomBatch([(GameTable(d), n, r) for d, n, r in segs], seed=seed, first_room=first, restart=True, max_fuse=32) as b:
        b.step(turns)
        b.step(1)                                              # and a single-turn launch of the mixed kernel
        got = b.read_rooms()
    lo = 0
    for d, n, r in segs:
        assert_views_equal(got[lo:lo + r], oracle_batch(Oracle(d, n), r, seed, first + lo, turns + 1, restart=True), f"segment n={n} at {lo}")
        lo += r


if __name__ == "__main__":
    {"trace": trace, "humans": humans, "mixed": mixed}[sys.argv[1]](int(sys.argv[2]))
    print("knob scenario ok:", sys.argv[1:], flush=True)
