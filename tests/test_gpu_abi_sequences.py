"""The C ABI driven in the ORDERS a host really uses (-m gpu): a human's message is logged at the start of the NEXT graph
run (agent/tools/utils.py:310-358), so the live sequence of a served room is read -> inject -> step -> read, not "inject
before the first read".  Round 3 shipped a use-after-free that only that order reaches (inject_impl's regrow freed the pinned
staging buffer of read_rooms / write_rooms and left the pointer); these tests pin the orders:

* the regression itself: read (allocates the pinned staging) -> small inject -> large inject (regrows the device scratch)
  -> write / read, against the oracle, at a size whose ranges do not fit the 4 KB stack path;
* the one-room sequence of the crash (gpurun_out/segv.log of round 3) at one room and at a range just past 4 KB;
* a state-machine fuzz: read / write / inject / inject_actions / step / reset / set_turn / summary / read_events
  interleaved at random, batch sizes on both sides of the 4 KB stack / pinned threshold and action counts on both sides of
  the injection scratch's first size, every observable compared with the oracle model after every operation.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import load_dsl
from game_engine_amd import GameTable, GeError, RoomBatch
from oracle.oracle import Oracle
from parity_util import assert_views_equal, oracle_events, oracle_rooms_as_views, views_as_oracle_rooms

pytestmark = pytest.mark.gpu

WW, TT = "werewolf-(mafia)", "two-truths-and-a-lie"


def test_read_then_inject_regrow_then_write_read(dsl_ww):
    """4 096 Werewolf x 8 rooms (128 KB of records: the pinned path) with a host-driven seat."""
    R, n, mask, seed, first = 4096, 8, 0b1, 17, 1 << 22
    orc = Oracle(dsl_ww, n)
    rooms = orc.init_rooms(R)
    rng = np.random.default_rng(5)
    with RoomBatch([(GameTable(dsl_ww), n, R, mask)], seed=seed, first_room=first, max_fuse=1) as b:
        b.step(3)
        orc.run(rooms, seed, first, 0, 3, threads=0, human_mask=mask)
        assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), "first read (allocates the staging buffer)")
        for k in (100, 5000):                                   # 2.4 KB of scratch, then 120 KB: the regrow
            rr = rng.integers(0, R, size=k).astype(np.uint64)
            pl = np.ones(k, dtype=np.uint32)
            ch = rng.integers(0, n + 2, size=k).astype(np.uint32)
            want = [0 if orc.inject(rooms, int(r), 1, int(c)) else -1 for r, c in zip(rr, ch)]
            assert b.inject_actions(rr, pl, ch).tolist() == want
        views = oracle_rooms_as_views(orc, rooms)
        assert_views_equal(b.read_rooms(), views, "read after the regrow")
        b.write_rooms(0, views[::-1].copy())                    # the same staging buffer, host -> device
        assert_views_equal(b.read_rooms(), views[::-1], "write / read after the regrow")
        b.write_rooms(0, views)
        b.step(5)
        orc.run(rooms, seed, first, 3, 5, threads=0, human_mask=mask)
        assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), "stepped on")


@pytest.mark.parametrize("R", [1, 129, 600])
def test_read_inject_step_read_one_seat(dsl_ww, R):
    """The crashing sequence of round 3: read -> first inject -> step -> read, turn after turn, for one room (the 4 KB
    stack path) and for ranges just past it (129 x 32 B: the pinned path from the first read on)."""
    from oracle import dsl_table
    from oracle.human_script import scripted_human
    n, mask, first = 8, 0b1, 12345
    orc = Oracle(dsl_ww, n)
    otb = dsl_table.compile_dsl(dsl_ww)
    injected = 0
    for seed in (0, 1, 0xC0FFEE):                                   # (one room alone may lose its seat-1 player in the first night)
        rooms = orc.init_rooms(R)
        with RoomBatch([(GameTable(dsl_ww), n, R, mask)], seed=seed, first_room=first, max_fuse=1) as b:
            for t in range(45):
                got = b.read_rooms()
                assert_views_equal(got, oracle_rooms_as_views(orc, rooms), f"R={R} seed {seed} before turn {t}")
                for r in range(R):
                    act = scripted_human(otb, t, orc.project(rooms[r]), n)
                    if act:
                        assert orc.inject(rooms, r, act[0], act[1])
                        b.inject_action(r, act[0], act[1])
                        injected += 1
                b.step(1)
                orc.run(rooms, seed, first, t, 1, threads=0, human_mask=mask)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"R={R} seed {seed} at the end")
    assert injected >= R


class _Model:
    """The oracle side of a batch: one room array per segment + the turn counter."""

    def __init__(self, segs, seed, first, restart):
        self.segs, self.seed, self.first, self.restart = segs, seed, first, restart
        self.orcs = [Oracle(load_dsl(g), n) for g, n, _, _ in segs]
        self.lo = np.cumsum([0] + [r for _, _, r, _ in segs]).tolist()
        self.total = self.lo[-1]
        self.reset()

    def reset(self):
        self.rooms = [o.init_rooms(r) for o, (_, _, r, _) in zip(self.orcs, self.segs)]
        self.turn = 0
        self.last_events = None

    def seg_of(self, room):
        k = int(np.searchsorted(self.lo, room, side="right")) - 1
        return k, room - self.lo[k]

    def views(self, lo=0, hi=None):
        hi = self.total if hi is None else hi
        parts = []
        for k, o in enumerate(self.orcs):
            a, z = max(lo, self.lo[k]), min(hi, self.lo[k + 1])
            if a < z:
                parts.append(oracle_rooms_as_views(o, self.rooms[k][a - self.lo[k]: z - self.lo[k]]))
        return np.concatenate(parts)

    def step(self, k):
        ev = []
        for t in range(k):                                          # turn by turn: the event trace is per turn
            row = []
            for s, o in enumerate(self.orcs):
                o.run(self.rooms[s], self.seed, self.first + self.lo[s], self.turn, 1, threads=0, restart=self.restart,
                      human_mask=self.segs[s][3])
                row.append(oracle_events(o, self.rooms[s], self.turn))
            ev.append(np.concatenate(row))
            self.turn += 1
        self.last_events = np.stack(ev, axis=1)                     # [room, turn]

    def inject(self, room, player, choice):
        if room >= self.total:
            return -6
        k, r = self.seg_of(room)
        return 0 if self.orcs[k].inject(self.rooms[k], r, player, choice) else -1

    def write(self, lo, views):
        for k, o in enumerate(self.orcs):
            a, z = max(lo, self.lo[k]), min(lo + len(views), self.lo[k + 1])
            if a < z:
                self.rooms[k][a - self.lo[k]: z - self.lo[k]] = views_as_oracle_rooms(o, views[a - lo: z - lo])

    def summary(self):
        s = {"rooms": self.total, "turn": self.turn, "finished": 0, "village_wins": 0, "wolf_wins": 0, "alive_players": 0,
             "sum_end_turn": 0, "end_turn_hist": np.zeros(16, dtype=np.int64), "score_hist": np.zeros(16, dtype=np.int64),
             "games_recycled": 0}
        for o, rooms, (_, n, _, _) in zip(self.orcs, self.rooms, self.segs):
            fin = rooms["end_turn"] >= 0
            s["finished"] += int(fin.sum())
            s["sum_end_turn"] += int(rooms["end_turn"][fin].sum())
            s["end_turn_hist"] += np.bincount(np.minimum(rooms["end_turn"][fin] // 8, 15), minlength=16)
            s["games_recycled"] += int(rooms["games"].sum())
            if o.table.pack == 1:
                alive = rooms["p"][:, :n, 2]
                wolves = ((rooms["p"][:, :n, 1] == 2) & (alive == 1)).sum(axis=1)
                s["alive_players"] += int(alive.sum())
                s["village_wins"] += int((fin & (wolves == 0)).sum())
                s["wolf_wins"] += int((fin & (wolves > 0)).sum())
            else:
                s["alive_players"] += n * len(rooms)                 # nobody is eliminated in Two-Truths
                s["score_hist"] += np.bincount(np.minimum(rooms["p"][:, :n, 7].ravel(), 15), minlength=16)
        s["end_turn_hist"], s["score_hist"] = s["end_turn_hist"].tolist(), s["score_hist"].tolist()
        return s


# rooms per segment straddle the 4 KB stack / pinned threshold of rooms_io: 32-B records 128 | 129, 24-B 170 | 171, 40-B 102 | 103
SCENARIOS = [
    ("ww8-128-trace", [(WW, 8, 128, 0b1)], 1, False, True),
    ("ww8-129-trace", [(WW, 8, 129, 0b1)], 4, True, True),
    ("ww8-4096", [(WW, 8, 4096, 0b101)], 4, True, False),
    ("mixed-small", [(WW, 12, 102, 0b1), (TT, 4, 170, 0b11), (WW, 8, 128, 0b10)], 8, True, False),
    ("mixed-large", [(WW, 12, 103, 0b100000000001), (TT, 4, 171, 0b1), (WW, 8, 3000, 0)], 2, False, True),
    ("tt4-5000-trace", [(TT, 4, 5000, 0b11)], 3, True, True),
    # single-game Werewolf x 12 in single-turn launches: the side plane of prepared deals (both kernel builds) under resets,
    # turn changes and overwritten rooms
    ("ww12-300-single", [(WW, 12, 300, 0b1)], 1, True, True),
    ("ww12-70000-single", [(WW, 12, 70000, 0)], 1, True, False),
]


@pytest.mark.parametrize("name,segs,fuse,restart,trace", SCENARIOS, ids=[s[0] for s in SCENARIOS])
@pytest.mark.parametrize("fuzz_seed", range(int(os.environ.get("GE_SEQ_FUZZ_SEEDS", "2"))))     # (a soak run sets more)
def test_abi_state_machine_fuzz(name, segs, fuse, restart, trace, fuzz_seed):
    rng = np.random.default_rng(zlib.crc32(name.encode()) + fuzz_seed)
    seed, first = 777 + fuzz_seed, (1 << 30) + 3
    m = _Model(segs, seed, first, restart)
    tables = {g: GameTable(load_dsl(g)) for g in {s[0] for s in segs}}
    ops_done = {}
    with RoomBatch([(tables[g], n, r, mask) for g, n, r, mask in segs], seed=seed, first_room=first, max_fuse=fuse,
                   restart=restart, trace=trace) as b:
        nmax = max(n for _, n, _, _ in segs)
        snapshots = []                                               # (views, turn) taken along the way, written back later
        for i in range(70):
            op = rng.choice(["step", "step", "read", "read_part", "write", "inject", "inject_many", "reset", "set_turn",
                             "summary", "events"], p=[.2, .1, .1, .12, .1, .08, .14, .02, .04, .05, .05])
            ops_done[op] = ops_done.get(op, 0) + 1
            what = f"{name} fuzz {fuzz_seed} op {i} {op}"
            if op == "step":
                k = int(rng.integers(1, fuse + 1)) if trace else int(rng.choice([1, 2, fuse, 3 * fuse + 1]))
                b.step(k)
                m.step(k)
            elif op == "read":
                assert_views_equal(b.read_rooms(), m.views(), what)
                if rng.random() < .5:
                    snapshots.append((m.views(), m.turn))
            elif op == "read_part":
                lo = int(rng.integers(0, m.total))
                cnt = int(rng.integers(0, m.total - lo + 1))
                if cnt:
                    assert_views_equal(b.read_rooms(lo, cnt), m.views(lo, lo + cnt), what)
                else:
                    assert len(b.read_rooms(lo, 0)) == 0
            elif op == "write":
                if snapshots and rng.random() < .7:                  # an earlier state of a sub-range (no turn change)
                    views, _ = snapshots[int(rng.integers(len(snapshots)))]
                else:
                    views = m.views()
                lo = int(rng.integers(0, m.total))
                cnt = int(rng.integers(1, m.total - lo + 1))
                part = np.ascontiguousarray(views[lo: lo + cnt])
                b.write_rooms(lo, part)
                m.write(lo, part)
                if rng.random() < .3:                                # and a refused write changes nothing
                    bad = part.copy()
                    j = int(rng.integers(cnt))
                    kind = int(rng.integers(3))
                    if kind == 0:
                        bad["n_players"][j] += 1
                    elif kind == 1:
                        bad["phase_id"][j] = 4242
                    else:
                        bad["pack"][j] ^= 3
                    with pytest.raises(GeError) as e:
                        b.write_rooms(lo, bad)
                    assert e.value.status == -1
            elif op == "inject":
                room, pl, ch = int(rng.integers(0, m.total)), int(rng.integers(0, nmax + 2)), int(rng.integers(0, nmax + 2))
                want = m.inject(room, pl, ch)
                try:
                    b.inject_action(room, pl, ch)
                    got = 0
                except GeError as e:
                    got = e.status
                assert got == want, what
            elif op == "inject_many":
                # 170 | 171 actions: 4 084 | 4 108 bytes of scratch, either side of its first size; 3 000: a regrow
                k = int(rng.choice([1, 50, 170, 171, 3000]))
                rr = rng.integers(0, m.total + (3 if rng.random() < .3 else 0), size=k).astype(np.uint64)
                pl = rng.integers(0, nmax + 2, size=k).astype(np.uint32)
                ch = rng.integers(0, nmax + 2, size=k).astype(np.uint32)
                want = [m.inject(int(r), int(p), int(c)) for r, p, c in zip(rr, pl, ch)]
                assert b.inject_actions(rr, pl, ch).tolist() == want, what
            elif op == "reset":
                b.reset()
                m.reset()
            elif op == "set_turn":
                t = int(rng.integers(0, 500))
                b.set_turn(t)
                m.turn = t
                assert b.turn == t
            elif op == "summary":
                got = b.summary()
                want = m.summary()
                for key, v in want.items():
                    assert got[key] == v, (what, key)
            elif op == "events":
                if not trace:
                    with pytest.raises(GeError) as e:
                        b.read_events()
                    assert e.value.status == -7
                    continue
                if m.last_events is None:
                    continue
                lo = int(rng.integers(0, m.total))
                cnt = int(rng.integers(1, m.total - lo + 1))
                ev = b.read_events(lo, cnt)
                want = m.last_events[lo: lo + cnt]
                assert ev.shape == want.shape, what
                for f in ("turn", "from_phase_id", "to_phase_id", "acted_now", "restarted", "choice"):
                    assert (ev[f] == want[f]).all(), (what, f)
            if op in ("reset",):
                snapshots = []
        assert_views_equal(b.read_rooms(), m.views(), f"{name} fuzz {fuzz_seed}: final read")
        got, want = b.summary(), m.summary()
        for key, v in want.items():
            assert got[key] == v, (name, key)
