"""GPU parity (-m gpu): the HIP path, called through the C ABI, against
  (1) the committed golden vectors (reference node coroutines under the fixed policy),
  (2) the oracle on the same seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Integer path: every comparison is bit-exact."""
import numpy as np
import pytest

from conftest import golden_dsl, golden_files, human_files, load_dsl, load_golden, restart_files
from game_engine_amd import GameTable, GeError, RoomBatch
from game_engine_amd.stepper import project_view
from parity_util import assert_views_equal, oracle_batch, oracle_rooms_as_views

pytestmark = pytest.mark.gpu
SEEDS = [0, 1, 0xC0FFEE]


def _oracle(dsl, n, rounds=1):
    from oracle.oracle import Oracle
    return Oracle(dsl, n, rounds=rounds)


@pytest.mark.parametrize("name", golden_files())
def test_golden_trajectories_turn_by_turn(name):
    """Every turn of every committed reference trajectory, one launch per turn."""
    g = load_golden(name)
    tb = GameTable(golden_dsl(g), rounds=g["rounds"])
    for case in g["cases"]:
        with RoomBatch([(tb, g["n_players"], 1)], seed=case["seed"], first_room=case["room"], max_fuse=1) as b:
            for t, want in enumerate(case["turns"]):
                b.step(1)
                got = project_view(b.read_rooms(0, 1)[0], tb)
                assert got == want, f"{name} seed={case['seed']:#x} room={case['room']} turn={t}"


@pytest.mark.parametrize("name", golden_files())
def test_golden_trajectories_fused(name):
    """Same end states when all turns run inside one launch (state kept in registers)."""
    g = load_golden(name)
    tb = GameTable(golden_dsl(g), rounds=g["rounds"])
    for case in g["cases"]:
        T = len(case["turns"])
        with RoomBatch([(tb, g["n_players"], 1)], seed=case["seed"], first_room=case["room"], max_fuse=T) as b:
            b.step(T)
            assert project_view(b.read_rooms(0, 1)[0], tb) == case["turns"][-1]


@pytest.mark.parametrize("game,n,n_rooms,turns,rounds", [
    ("werewolf-(mafia)", 8, 65536, 64, 1),          # BASELINE config C2
    ("werewolf-(mafia)", 12, 20000, 80, 1),
    ("werewolf-(mafia)", 12, 40000, 70, 1),         # 32 769 .. 65 535 rooms: lone wavefronts of 33 .. 63 rooms (ge_step.hip launch_geometry)
    ("werewolf-(mafia)", 8, 57001, 70, 1),
    ("two-truths-and-a-lie", 7, 49152, 70, 1),
    ("werewolf-(mafia)", 4, 5000, 40, 1),
    ("werewolf-(mafia)", 9, 3001, 64, 1),
    ("draft-werewolf-(mafia)", 8, 65536, 72, 1),    # the reference's earlier Werewolf draft: own field names, two terminal phases
    ("draft-werewolf-(mafia)", 11, 3001, 90, 1),
    ("two-truths-and-a-lie", 4, 65536, 64, 1),
    ("two-truths-and-a-lie", 3, 777, 48, 1),
    ("two-truths-and-a-lie", 7, 4099, 120, 2),
    ("two-truths-and-a-lie", 12, 2048, 150, 1),
])
@pytest.mark.parametrize("seed", SEEDS)
def test_batch_equals_oracle(game, n, n_rooms, turns, rounds, seed):
    dsl = load_dsl(game)
    first = 123456789 if seed else 0
    with RoomBatch([(GameTable(dsl, rounds), n, n_rooms)], seed=seed, first_room=first) as b:
        b.step(turns)
        got = b.read_rooms()
    want = oracle_batch(_oracle(dsl, n, rounds), n_rooms, seed, first, turns)
    assert_views_equal(got, want, f"{game} n={n} seed={seed:#x}")


@pytest.mark.parametrize("name", restart_files())
def test_restart_mode_golden(name):
    g = load_golden(name)
    tb = GameTable(golden_dsl(g))
    for case in g["cases"]:
        with RoomBatch([(tb, g["n_players"], 1)], seed=case["seed"], first_room=case["room"],
                       max_fuse=1, restart=True) as b:
            for t, want in enumerate(case["turns"]):
                b.step(1)
                assert project_view(b.read_rooms(0, 1)[0], tb) == want, f"{name} seed={case['seed']:#x} turn={t}"
            assert int(b.read_rooms(0, 1)[0]["games"]) == case["games"]


@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("werewolf-(mafia)", 12), ("two-truths-and-a-lie", 4),
                                    ("draft-werewolf-(mafia)", 8)])
def test_restart_mode_equals_oracle(game, n):
    """The bench workload: steady state, finished rooms recycled, many fused launches."""
    dsl = load_dsl(game)
    R, turns, seed = 20000, 700, 0xC0FFEE
    with RoomBatch([(GameTable(dsl), n, R)], seed=seed, restart=True) as b:
        b.step(turns)
        got = b.read_rooms()
        s = b.summary()
    want = oracle_batch(_oracle(dsl, n), R, seed, 0, turns, restart=True)
    assert_views_equal(got, want, f"restart {game} n={n}")
    assert s["games_recycled"] == int(want["games"].sum()) > R


@pytest.mark.parametrize("game,n,n_rooms", [("werewolf-(mafia)", 8, 262144), ("werewolf-(mafia)", 12, 200000),
                                            ("two-truths-and-a-lie", 4, 262144), ("draft-werewolf-(mafia)", 12, 200000)])
def test_high_occupancy_path_equals_oracle(game, n, n_rooms):
    """Above ~196 608 rooms the launch switches to its many-wavefronts-per-SIMD code path (256-thread
    blocks, predicated queue writes): compare it with the oracle too, steady state."""
    dsl = load_dsl(game)
    seed, turns = 0xC0FFEE, 96
    with RoomBatch([(GameTable(dsl), n, n_rooms)], seed=seed, restart=True) as b:
        b.step(turns)
        got = b.read_rooms()
    want = oracle_batch(_oracle(dsl, n), n_rooms, seed, 0, turns, restart=True)
    assert_views_equal(got, want, f"{game} n={n} rooms={n_rooms}")


@pytest.mark.parametrize("game,n,n_rooms,first", [("two-truths-and-a-lie", 4, 1 << 20, 0),            # C3
                                                  ("werewolf-(mafia)", 8, 1 << 20, 1 << 33),
                                                  ("werewolf-(mafia)", 12, 1 << 21, 5 << 21)])      # rank 5's share of C4
def test_full_baseline_sizes_equal_oracle(game, n, n_rooms, first):
    """The BASELINE shapes at their full per-GPU size (256-room blocks, 16+ wavefronts per SIMD), every
    room against the oracle: the oracle runs chunk by chunk on the host cores (rooms are independent
    and keyed by their global index), the GPU batch in one piece."""
    dsl = load_dsl(game)
    orc = _oracle(dsl, n)
    seed, turns, chunk = 0xC0FFEE, 80, 1 << 18
    with RoomBatch([(GameTable(dsl), n, n_rooms)], seed=seed, first_room=first, restart=True) as b:
        b.step(turns)
        s = b.summary()
        recycled = 0
        for lo in range(0, n_rooms, chunk):
            want = oracle_batch(orc, chunk, seed, first + lo, turns, restart=True)
            assert_views_equal(b.read_rooms(lo, chunk), want, f"{game} n={n} rooms {lo}..{lo + chunk}")
            recycled += int(want["games"].sum())
    assert s["games_recycled"] == recycled and s["rooms"] == n_rooms


@pytest.mark.parametrize("name", human_files())
def test_host_driven_player_golden(name):
    """Player 1 host-driven (human_mask) + ge_batch_inject_action, against reference-run vectors."""
    from oracle.human_script import scripted_human
    from oracle import dsl_table
    g = load_golden(name)
    dsl = golden_dsl(g)
    tb, otb, n = GameTable(dsl), dsl_table.compile_dsl(dsl), g["n_players"]
    for case in g["cases"]:
        with RoomBatch([(tb, n, 1, g["human_mask"])], seed=case["seed"], first_room=case["room"], max_fuse=1) as b:
            for t, want in enumerate(case["turns"]):
                act = scripted_human(otb, t, project_view(b.read_rooms(0, 1)[0], tb), n)
                if act:
                    b.inject_action(0, act[0], act[1])
                b.step(1)
                assert project_view(b.read_rooms(0, 1)[0], tb) == want, f"{name} seed={case['seed']:#x} turn={t}"


@pytest.mark.parametrize("game,n,mask", [("werewolf-(mafia)", 8, 0b1), ("werewolf-(mafia)", 12, 0b100000000101),
                                         ("two-truths-and-a-lie", 4, 0b11)])
def test_host_driven_players_random_injection_equals_oracle(game, n, mask):
    """Many rooms, several host-driven players, actions injected at random moments with random
    (valid) choices; refused injections (not a target, already acted, dead target) agree too."""
    dsl = load_dsl(game)
    orc = _oracle(dsl, n)
    R, seed, first = 300, 21, 5000
    rng = np.random.default_rng(n + mask)
    rooms = orc.init_rooms(R)
    accepted = refused = 0
    with RoomBatch([(GameTable(dsl), n, R, mask)], seed=seed, first_room=first, max_fuse=1) as b:
        for t in range(70):
            for r in rng.choice(R, size=40, replace=False):
                pl = int(rng.choice([i + 1 for i in range(n) if (mask >> i) & 1]))
                ch = int(rng.integers(0, n + 2))
                ok = orc.inject(rooms, int(r), pl, ch)
                try:
                    b.inject_action(int(r), pl, ch)
                    got_ok = True
                except GeError as e:
                    assert e.status == -1
                    got_ok = False
                assert ok == got_ok, (t, int(r), pl, ch)
                accepted += ok
                refused += not ok
            b.step(1)
            orc.run(rooms, seed, first, t, 1, threads=0, human_mask=mask)
            from parity_util import oracle_rooms_as_views
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"{game} turn {t}")
    assert accepted > 50 and refused > 50


def test_fused_equals_unfused_and_chunked(dsl_ww):
    tb = GameTable(dsl_ww)
    outs = []
    for fuse, chunks in ((1, [50]), (64, [50]), (7, [13, 1, 36]), (64, [1] * 50)):
        with RoomBatch([(tb, 8, 10000)], seed=5, max_fuse=fuse) as b:
            for c in chunks:
                b.step(c)
            assert b.turn == 50
            outs.append(b.read_rooms().tobytes())
    assert outs[0] == outs[1] == outs[2] == outs[3]


@pytest.mark.parametrize("players,sizes", [
    ((8, 4, 12, 12), (3000, 5000, 700, 1100)),               # (Werewolf, Two-Truths, Werewolf, Two-Truths) players; lone-wavefront build
    ((8, 7, 9, 4), (2000, 900, 1500, 3000)),
    ((8, 4, 12, 12), (50000, 40000, 30001, 15000)),          # large-batch build of the mixed kernel, 64-room blocks
    ((8, 7, 9, 4), (60000, 20000, 30000, 25000))])
def test_mixed_batch_one_launch(dsl_ww, dsl_tt, players, sizes):
    """BASELINE config C5 shape and beyond: Werewolf and Two-Truths rooms of several sizes (different
    phase graphs, record layouts and action paths) advanced by the same launches; global room indices
    run across segments."""
    ww, tt = GameTable(dsl_ww), GameTable(dsl_tt)
    turns, seed, first = 64, 0xC0FFEE, 1 << 33
    segs = [(dsl_ww, ww, players[0]), (dsl_tt, tt, players[1]), (dsl_ww, ww, players[2]), (dsl_tt, tt, players[3])]
    with RoomBatch([(tb, n, r) for (_, tb, n), r in zip(segs, sizes)], seed=seed, first_room=first) as b:
        b.step(turns)
        got = b.read_rooms()
        s = b.summary()
    lo = 0
    for (dsl, _, n), r in zip(segs, sizes):
        assert_views_equal(got[lo:lo + r], oracle_batch(_oracle(dsl, n), r, seed, first + lo, turns), f"segment n={n} at {lo}")
        lo += r
    assert s["rooms"] == sum(sizes) and s["finished"] == int((got["end_turn"] >= 0).sum())


def test_mixed_batch_full_c5_share_equals_oracle(dsl_ww, dsl_tt):
    """One GPU's share of BASELINE config C5 at full size: 524 288 Werewolf x 8 + 524 288 Two-Truths x 4
    rooms in the same launches (mixed kernel, 256-room blocks), steady state, every room vs the oracle."""
    half, seed, turns, first, chunk = 1 << 19, 0xC0FFEE, 80, 7 << 20, 1 << 18
    with RoomBatch([(GameTable(dsl_ww), 8, half), (GameTable(dsl_tt), 4, half)], seed=seed, first_room=first, restart=True) as b:
        b.step(turns)
        for dsl, n, base in ((dsl_ww, 8, 0), (dsl_tt, 4, half)):
            orc = _oracle(dsl, n)
            for lo in range(base, base + half, chunk):
                want = oracle_batch(orc, chunk, seed, first + lo, turns, restart=True)
                assert_views_equal(b.read_rooms(lo, chunk), want, f"n={n} rooms {lo}..{lo + chunk}")


def test_summary_matches_readback(dsl_ww, dsl_tt):
    for dsl, n, pack in ((dsl_ww, 8, 1), (dsl_tt, 4, 2)):
        with RoomBatch([(GameTable(dsl), n, 30011)], seed=1) as b:
            b.step(40)
            r = b.read_rooms()
            s = b.summary()
        fin = r["end_turn"] >= 0
        assert s["rooms"] == 30011 and s["turn"] == 40
        assert s["finished"] == int(fin.sum()) and s["sum_end_turn"] == int(r["end_turn"][fin].sum())
        hist = np.bincount(np.minimum(r["end_turn"][fin] // 8, 15), minlength=16)
        assert s["end_turn_hist"] == hist.tolist()
        if pack == 1:
            alive = r["players"][:, :n, 2]
            wolves_alive = ((r["players"][:, :n, 1] == 2) & (alive == 1)).sum(axis=1)
            assert s["alive_players"] == int(alive.sum())
            assert s["village_wins"] == int((fin & (wolves_alive == 0)).sum())
            assert s["wolf_wins"] == int((fin & (wolves_alive > 0)).sum())
            assert s["village_wins"] + s["wolf_wins"] == s["finished"]
        else:
            sc = np.bincount(np.minimum(r["players"][:, :n, 7].ravel(), 15), minlength=16)
            assert s["score_hist"] == sc.tolist()


def test_shard_invariance(dsl_ww):
    """results do not depend on how rooms are split (RNG keyed by global room index);
    summaries of shards add up to the whole-job summary, checksum included."""
    tb = GameTable(dsl_ww)
    R, turns, seed = 40000, 64, 7
    with RoomBatch([(tb, 8, R)], seed=seed, first_room=1000) as b:
        b.step(turns)
        whole, sw = b.read_rooms(), b.summary_words()
    from game_engine_amd.dist import reduce_summaries
    parts, words = [], []
    for lo, hi in ((0, 12345), (12345, 12346), (12346, R)):
        with RoomBatch([(tb, 8, hi - lo)], seed=seed, first_room=1000 + lo) as b:
            b.step(turns)
            parts.append(b.read_rooms())
            words.append(b.summary_words())
    assert np.concatenate(parts).tobytes() == whole.tobytes()
    assert reduce_summaries(np.stack(words)).tolist() == sw.tolist()      # checksum included


def test_write_read_roundtrip_and_restore(dsl_ww, dsl_tt):
    """Checkpoint/restore through canonical views INTO A FRESH BATCH: a checkpoint is (room records,
    turn); ge_batch_set_turn restores the counter the RNG is keyed by.  Same end state as the
    uninterrupted run."""
    for dsl, n in ((dsl_ww, 12), (dsl_ww, 6), (dsl_tt, 4), (dsl_tt, 8), (dsl_tt, 11)):
        tb = GameTable(dsl)
        with RoomBatch([(tb, n, 999)], seed=3, first_room=50) as a:
            a.step(23)
            mid = a.read_rooms()
            a.step(41)
            end = a.read_rooms()
        with RoomBatch([(tb, n, 999)], seed=3, first_room=50) as b:
            b.write_rooms(0, np.ascontiguousarray(mid[::-1])[::-1].copy())
            b.set_turn(23)
            assert b.turn == 23
            assert b.read_rooms().tobytes() == mid.tobytes()
            b.step(41)
            assert b.read_rooms().tobytes() == end.tobytes()


def test_invariants_at_full_sizes(dsl_ww, dsl_tt):
    """BASELINE sizes C3 (1 048 576 Two-Truths x 4) and one GPU's share of C4 (2 097 152 Werewolf
    x 12): size-independent properties instead of an oracle run."""
    with RoomBatch([(GameTable(dsl_tt), 4, 1 << 20)], seed=1) as b:
        b.step(64)
        s1 = b.summary()
        b.step(64)
        s2 = b.summary()
        sample = b.read_rooms(1 << 19, 4096)
    assert s2["finished"] >= s1["finished"] > 0 and s2["finished"] == 1 << 20   # terminal is absorbing, all done by 128
    assert sum(s2["score_hist"]) == 4 << 20
    pts = sample["players"][:, :4, 7].astype(int).sum(axis=1)
    assert (pts == 3 * 4).all()      # every vote gives exactly one point (voter if right, speaker if fooled)
    assert (sample["players"][:, :4, 8] == 1).all() and (sample["phase_id"] == 99).all()

    with RoomBatch([(GameTable(dsl_ww), 12, 1 << 21)], seed=0xC0FFEE, first_room=3 << 21) as b:
        b.step(2)
        r = b.read_rooms(0, 8192)
        roles = r["players"][:, :12, 0]
        assert ((roles == 2).sum(axis=1) == 3).all() and ((roles == 3).sum(axis=1) == 1).all()
        assert ((roles == 4).sum(axis=1) == 1).all() and ((roles == 1).sum(axis=1) == 7).all()
        b.step(62)
        s1, alive1 = b.summary(), b.read_rooms(0, 8192)["players"][:, :12, 2].copy()
        b.step(64)
        s2, r2 = b.summary(), b.read_rooms(0, 8192)
    assert s1["finished"] <= s2["finished"] and s2["village_wins"] + s2["wolf_wins"] == s2["finished"]
    assert (r2["players"][:, :12, 2] <= alive1).all()                 # the dead stay dead
    dead = r2["players"][:, :12, 2] == 0
    assert (r2["players"][:, :12, 4][dead] == 0).all() and (r2["players"][:, :12, 3][dead] == 1).all()
    assert s2["alive_players"] <= s1["alive_players"] <= 12 << 21


@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("two-truths-and-a-lie", 4)])
def test_counters_saturate_like_the_oracle(game, n):
    """Long-running steady state: end_turn saturates at 65 534 for games that end later (POLICY §1.3), the
    per-slot game counter - which also keys the role deal - at 65 535 (§2).  Rooms start with the
    counter just below its ceiling and run past turn 65 534."""
    dsl = load_dsl(game)
    orc = _oracle(dsl, n)
    R, seed, first, turns = 512, 3, 1 << 20, 65700
    rooms = orc.init_rooms(R)
    rooms["games"] = 65535 - 20 - (np.arange(R) % 7)
    with RoomBatch([(GameTable(dsl), n, R)], seed=seed, first_room=first, restart=True, max_fuse=1024) as b:
        b.write_rooms(0, oracle_rooms_as_views(orc, rooms))
        b.step(turns)
        got = b.read_rooms()
    orc.run(rooms, seed, first, 0, turns, threads=0, restart=True)
    want = oracle_rooms_as_views(orc, rooms)
    assert_views_equal(got, want, f"{game} after {turns} turns")
    assert (want["games"] == 65535).all()
    done = want["end_turn"] >= 0
    assert done.any() and (want["end_turn"][done] == 65534).all()      # whoever is finished right now finished late


def test_graph_replay_of_short_launches_equals_fused(dsl_ww, dsl_tt):
    """A step() call that needs >= 4 launches (small max_fuse) is captured into a hipGraph per n_turns and
    replayed with a device-side turn base; mixing replays, plain launches and resets must not change a bit."""
    for dsl, n in ((dsl_ww, 8), (dsl_tt, 4)):
        tb, R, seed, first = GameTable(dsl), 5000, 77, 12345
        plan = [7, 7, 7, 2, 7, 9, 1, 9, 7]
        with RoomBatch([(tb, n, R)], seed=seed, first_room=first, max_fuse=1, restart=True) as g, \
             RoomBatch([(tb, n, R)], seed=seed, first_room=first, max_fuse=64, restart=True) as f:
            for rep in range(2):
                for k in plan:
                    g.step(k)
                f.step(sum(plan))
                assert g.turn == f.turn == sum(plan)
                assert g.read_rooms().tobytes() == f.read_rooms().tobytes(), f"n={n} pass {rep}"
                if rep == 0:
                    want = oracle_batch(_oracle(dsl, n), R, seed, first, sum(plan), restart=True)
                    assert_views_equal(g.read_rooms(), want, f"graph path vs oracle n={n}")
                    g.reset(); f.reset()


def test_argument_and_range_errors(dsl_ww, dsl_tt):
    tb = GameTable(dsl_ww)
    for n in (3, 13):
        with pytest.raises(GeError):
            RoomBatch([(tb, n, 10)])
    with pytest.raises(GeError):
        RoomBatch([(tb, 8, 0)])
    with pytest.raises(GeError):
        RoomBatch([(GameTable(dsl_tt, rounds=15), 12, 4)])          # scores would not fit a byte
    with RoomBatch([(tb, 8, 10)]) as b:
        with pytest.raises(GeError) as e:
            b.read_rooms(5, 6)
        assert e.value.status == -6
        b.step(0)
        assert b.turn == 0


@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("werewolf-(mafia)", 12), ("two-truths-and-a-lie", 4),
                                    ("draft-werewolf-(mafia)", 8)])
def test_turn_counter_past_16_bits_equals_oracle(game, n):
    """The turn index is 32 bits in the RNG key but end_turn is a 16-bit field of the record that saturates at 0xFFFE
    (0xFFFF = not finished): a batch whose clock starts shortly before turn 65 536 and runs across it, steady state, every
    room against the oracle - in fused launches and one launch per turn."""
    dsl = load_dsl(game)
    orc = _oracle(dsl, n)
    R, seed, t0, turns = 6000, 0xC0FFEE, 65536 - 300, 900
    want = orc.init_rooms(R)
    orc.run(want, seed, 0, t0, turns, threads=0, restart=True)
    from parity_util import oracle_rooms_as_views
    want_v = oracle_rooms_as_views(orc, want)
    assert int(want_v["end_turn"].max()) == 0xFFFE                    # some room finished past the field's range
    for fuse in (0, 1):
        with RoomBatch([(GameTable(dsl), n, R)], seed=seed, restart=True, max_fuse=fuse) as b:
            b.set_turn(t0)
            b.step(turns if fuse == 0 else 400)
            if fuse:
                ref = orc.init_rooms(R)
                orc.run(ref, seed, 0, t0, 400, threads=0, restart=True)
                assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, ref), f"{game} n={n} one launch per turn")
            else:
                assert_views_equal(b.read_rooms(), want_v, f"{game} n={n} fused")
                assert b.turn == t0 + turns


@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("two-truths-and-a-lie", 4)])
def test_game_counter_saturates_like_the_oracle(game, n):
    """`games` (how many games a slot has completed; keys the role deal) is a 16-bit field that stops at 0xFFFF: slots
    written with counters just below it, stepped in steady state through several games, every room against the oracle."""
    from parity_util import oracle_rooms_as_views
    dsl = load_dsl(game)
    orc = _oracle(dsl, n)
    R, seed, turns = 3000, 7, 400
    rooms = orc.init_rooms(R)
    rooms["games"] = 0xFFFF - 2 - (np.arange(R) % 3)
    with RoomBatch([(GameTable(dsl), n, R)], seed=seed, restart=True) as b:
        b.write_rooms(0, oracle_rooms_as_views(orc, rooms))
        b.step(turns)
        got = b.read_rooms()
    orc.run(rooms, seed, 0, 0, turns, threads=0, restart=True)
    assert int(rooms["games"].max()) == 0xFFFF and int(rooms["games"].min()) == 0xFFFF
    assert_views_equal(got, oracle_rooms_as_views(orc, rooms), f"{game} n={n}")


def test_checkpoint_file_resumes_bit_exact(tmp_path):
    """RoomBatch.save_checkpoint / load_checkpoint: a mixed batch (both games, a host-driven seat) written to a file in the
    middle of a run and rebuilt from that file alone continues exactly like the uninterrupted batch."""
    ww, tt = load_dsl("werewolf-(mafia)"), load_dsl("two-truths-and-a-lie")
    segs = lambda: [(GameTable(ww), 8, 5000, 0b1), (GameTable(tt, 2), 5, 3000)]
    seed, first = 0xC0FFEE, 1 << 40
    with RoomBatch(segs(), seed=seed, first_room=first, restart=True) as whole:
        whole.step(37)
        path = str(tmp_path / "batch.npz")
        whole.save_checkpoint(path)
        whole.step(30)
        want, want_sum = whole.read_rooms(), whole.summary()
    with RoomBatch.load_checkpoint(path) as resumed:
        assert resumed.turn == 37 and resumed.n_rooms == 8000
        resumed.step(30)
        assert_views_equal(resumed.read_rooms(), want, "resumed from the checkpoint file")
        got_sum = resumed.summary()
    assert got_sum["checksum"] == want_sum["checksum"] and got_sum["turn"] == 67
    assert got_sum["games_recycled"] <= want_sum["games_recycled"]        # (the counter restarts with the new batch; states do not)


@pytest.mark.parametrize("game,n,n_rooms", [("werewolf-(mafia)", 8, 1 << 25),       # 1 GiB of records
                                            ("werewolf-(mafia)", 12, 1 << 24),      # the whole of C4 on one GPU, 640 MiB
                                            ("two-truths-and-a-lie", 4, 1 << 25)])  # 768 MiB
def test_single_turn_launches_beyond_the_infinity_cache_equal_fused_and_oracle(game, n, n_rooms):
    """The sizes bench.py's hbm_streaming_beyond_l3 streams (resident state larger than the 256 MiB Infinity Cache): 70
    single-turn launches == 70 fused turns (the whole summary, the checksum over every packed record included), and windows
    of rooms at both ends and in the middle equal the oracle.  70 turns cross the Werewolf x 12 side plane's refill turn and
    a restart, so prepared deals are written, read back and consumed at this size."""
    dsl = load_dsl(game)
    tb = GameTable(dsl)
    orc = _oracle(dsl, n)
    seed, first, turns, win = 0xC0FFEE, 1 << 36, 70, 4096
    with RoomBatch([(tb, n, n_rooms)], seed=seed, first_room=first, max_fuse=1, restart=True) as k1:
        k1.step(turns)
        s1 = k1.summary()
        for lo in (0, n_rooms // 2 - 77, n_rooms - win):
            assert_views_equal(k1.read_rooms(lo, win), oracle_batch(orc, win, seed, first + lo, turns, restart=True),
                               f"{game} x{n}: rooms {lo}.. of {n_rooms}, single-turn launches")
    with RoomBatch([(tb, n, n_rooms)], seed=seed, first_room=first, max_fuse=64, restart=True) as fz:
        fz.step(turns)
        sf = fz.summary()
    assert s1 == sf and s1["rooms"] == n_rooms and s1["turn"] == turns and s1["games_recycled"] > 0


def test_mixed_batch_single_turn_launches_full_c5_share(dsl_ww, dsl_tt):
    """One GPU's share of C5 (524 288 Werewolf x 8 + 524 288 Two-Truths x 4) through SINGLE-TURN launches of the mixed
    kernel's own single-turn build (one graph run = one state read + write, agent/game_agent_v2.py:1571-1587): 70 launches ==
    70 fused turns (whole summary, checksum over every record) and windows of both segments equal the oracle."""
    half, seed, first, turns, win = 1 << 19, 0xC0FFEE, 9 << 20, 70, 8192
    segs = [(GameTable(dsl_ww), 8, half), (GameTable(dsl_tt), 4, half)]
    with RoomBatch(segs, seed=seed, first_room=first, max_fuse=1, restart=True) as k1:
        k1.step(turns)
        s1 = k1.summary()
        for dsl, n, base in ((dsl_ww, 8, 0), (dsl_tt, 4, half)):
            orc = _oracle(dsl, n)
            for lo in (base, base + half // 2 - 33, base + half - win):
                assert_views_equal(k1.read_rooms(lo, win), oracle_batch(orc, win, seed, first + lo, turns, restart=True),
                                   f"C5 share, x{n}: rooms {lo}.., single-turn launches")
    with RoomBatch(segs, seed=seed, first_room=first, max_fuse=64, restart=True) as fz:
        fz.step(turns)
        sf = fz.summary()
    assert s1 == sf and s1["rooms"] == 2 * half and s1["games_recycled"] > 0


@pytest.mark.parametrize("n_rooms", [700, 140001])               # lone-wavefront and large-batch builds
def test_mixed_batch_with_werewolf_12_single_turn_launches(dsl_ww, dsl_tt, n_rooms):
    """A mixed batch whose single-turn launches take a Werewolf x 12 segment's role deals from the side plane (allocated by the
    first single-turn launch): every room of every segment vs the oracle, across the plane's refill turn and restarts; a fused
    step in between (the plane is not read) and ge_batch_set_turn (the plane is cleared) change nothing."""
    seed, first = 5, 1 << 30
    segs = [(GameTable(dsl_ww), 12, n_rooms), (GameTable(dsl_tt), 4, n_rooms // 2), (GameTable(dsl_ww), 8, n_rooms)]
    plan = [(1, 40), (64, 64), (1, 35)]
    with RoomBatch(segs, seed=seed, first_room=first, max_fuse=64, restart=True) as b:
        t = 0
        for fuse, turns in plan:
            if fuse == 1:
                for _ in range(turns):
                    b.step(1)
            else:
                b.step(turns)
            t += turns
        got = b.read_rooms()
    lo = 0
    for (tb, n, r), dsl in zip(segs, (dsl_ww, dsl_tt, dsl_ww)):
        want = oracle_batch(_oracle(dsl, n), r, seed, first + lo, t, restart=True)
        assert_views_equal(got[lo:lo + r], want, f"mixed batch segment x{n}, {n_rooms} rooms")
        lo += r


@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("werewolf-(mafia)", 12), ("two-truths-and-a-lie", 4), ("two-truths-and-a-lie", 9)])
def test_half_filled_lone_wavefronts_around_their_threshold(game, n):
    """A fused launch over at most 32 768 rooms runs 32 rooms per wavefront (csrc/ge_step.hip launch_geometry), one more room and it
    is 64 again; ragged sizes leave a last wavefront with a handful of rooms in either form.  Every room against the oracle, steady
    state, and the same batch stepped in single-turn launches (always 64 rooms per wavefront) ends in the same bytes."""
    dsl = load_dsl(game)
    tb, orc = GameTable(dsl), _oracle(dsl, n)
    for rooms in (32768, 32769, 31, 33, 32768 - 17):
        seed, first, turns = 0xBEEF + rooms, (1 << 34) + rooms, 75
        want = oracle_batch(orc, rooms, seed, first, turns, restart=True)
        with RoomBatch([(tb, n, rooms)], seed=seed, first_room=first, restart=True) as f, \
             RoomBatch([(tb, n, rooms)], seed=seed, first_room=first, restart=True, max_fuse=1) as k1:
            f.step(turns); k1.step(turns)
            got = f.read_rooms()
            assert_views_equal(got, want, f"{game} x{n}, {rooms} rooms, fused")
            assert got.tobytes() == k1.read_rooms().tobytes() and f.summary() == k1.summary()
