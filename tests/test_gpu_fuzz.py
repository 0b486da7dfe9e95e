"""GPU vs oracle from ARBITRARY states (-m gpu): rooms are overwritten with random — mostly
unreachable — canonical views (ge_batch_write_rooms), stepped, and compared with the oracle started
from the same views.  Catches any reliance on invariants that only hold along real games
(e.g. 'a player who has not acted has choice 0'), and the edge cases the reference's prose rules
imply: ties, empty target sets (dead Doctor / Detective), nobody alive on a team, all players dead."""
import numpy as np
import pytest

from conftest import load_dsl
from game_engine_amd import GameTable, RoomBatch
from game_engine_amd.stepper import ROOM_VIEW_DTYPE
from parity_util import assert_views_equal, oracle_rooms_as_views, views_as_oracle_rooms

pytestmark = pytest.mark.gpu


def _oracle(dsl, n, rounds=1):
    from oracle.oracle import Oracle
    return Oracle(dsl, n, rounds=rounds)


def _random_ww_views(orc, n, R, rng, consistent):
    v = np.zeros(R, dtype=ROOM_VIEW_DTYPE)
    ids = np.array(orc.ids)
    v["phase_id"] = ids[rng.integers(0, len(ids), R)]
    v["prev_phase_id"] = ids[rng.integers(0, len(ids), R)]
    v["phase0_done"] = rng.integers(0, 2, R)
    v["phase0_done"][v["phase_id"] != 0] = 1
    v["end_turn"] = -1
    v["n_players"], v["pack"] = n, 1
    p = v["players"]
    p[:, :n, 0] = rng.integers(0, 5, (R, n))                       # role class (0 = unassigned)
    # the packed model keeps ONE investigated_alignments memory per room (POLICY §3 assigns exactly one
    # Detective): keep the first Detective of each random room, demote the others
    det = p[:, :n, 0] == 4
    extra = det & (np.cumsum(det, axis=1) > 1)
    p[:, :n, 0][extra] = 1
    p[:, :n, 1] = rng.integers(0, 3, (R, n))                       # team
    if consistent:                                                 # team follows role, as the policy assigns them
        p[:, :n, 1] = np.where(p[:, :n, 0] == 2, 2, np.where(p[:, :n, 0] == 0, 0, 1))
    for f in (2, 3, 4, 5, 6, 7, 9):
        p[:, :n, f] = rng.integers(0, 2, (R, n))
    p[:, :n, 2] |= rng.integers(0, 2, (R, n)).astype(np.uint8)      # bias towards alive
    p[:, :n, 8] = rng.integers(0, n + 1, (R, n))                   # selected_target_id
    p[:, :n, 10] = rng.integers(0, n + 1, (R, n))                  # logged choice
    if consistent:
        p[:, :n, 10] *= p[:, :n, 9]                                # no choice without an action
    v["det"][:, :n] = rng.integers(0, 3, (R, n))
    return v


def _random_tt_views(orc, n, R, rng, rounds):
    v = np.zeros(R, dtype=ROOM_VIEW_DTYPE)
    ids = np.array(orc.ids)
    v["phase_id"] = ids[rng.integers(0, len(ids), R)]
    v["prev_phase_id"] = ids[rng.integers(0, len(ids), R)]
    v["phase0_done"] = 1
    v["end_turn"] = -1
    v["n_players"], v["pack"] = n, 2
    p = v["players"]
    for f in (0, 1, 3, 4, 6, 9):
        p[:, :n, f] = rng.integers(0, 2, (R, n))
    p[:, :n, 2] = rng.integers(0, 4, (R, n))
    p[:, :n, 5] = rng.integers(0, 4, (R, n))
    p[:, :n, 7] = rng.integers(0, 200, (R, n))
    p[:, :n, 8] = rng.integers(0, rounds + 1, (R, n))
    p[:, :n, 10] = rng.integers(0, 4, (R, n)) * p[:, :n, 9]
    return v


# R below / above the room count at which the launcher switches kernel builds (branch-lean vs LDS tables)
@pytest.mark.parametrize("n,consistent,R", [(8, True, 20000), (8, False, 20000), (12, True, 20000), (12, False, 20000),
                                            (4, True, 20000), (6, False, 20000), (8, False, 120000), (12, False, 110000)])
def test_werewolf_from_random_states(dsl_ww, n, consistent, R):
    seed, first = 99, 777
    rng = np.random.default_rng(n * 2 + consistent)
    orc = _oracle(dsl_ww, n)
    views = _random_ww_views(orc, n, R, rng, consistent)
    rooms = views_as_oracle_rooms(orc, views)
    with RoomBatch([(GameTable(dsl_ww), n, R)], seed=seed, first_room=first, max_fuse=3) as b:
        b.step(5)                                   # advance the clock: turns 5.. are the ones compared
        b.write_rooms(0, views)
        assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), "write/read of random views")
        for chunk in (1, 3, 2):
            b.step(chunk)
            orc.run(rooms, seed, first, b.turn - chunk, chunk, threads=0)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"n={n} after turn {b.turn}")


@pytest.mark.parametrize("n,R,fuse", [(8, 20000, 1), (8, 20000, 16), (8, 150000, 1), (12, 20000, 1), (12, 120000, 8), (5, 20000, 1)])
def test_werewolf_random_states_played_on_in_steady_state(dsl_ww, n, R, fuse):
    """From random (mostly unreachable) states through whole games with recycling: single-turn launches (their own kernel
    builds; for N <= 8 they prepare role deals into the records' spare half-word) mixed with fused ones, finished rooms
    restarted from the template, roles dealt again - every block of turns against the oracle."""
    seed, first = 4242, 1 << 35
    rng = np.random.default_rng(n * 31 + R % 7 + fuse)
    orc = _oracle(dsl_ww, n)
    views = _random_ww_views(orc, n, R, rng, consistent=False)
    rooms = views_as_oracle_rooms(orc, views)
    with RoomBatch([(GameTable(dsl_ww), n, R)], seed=seed, first_room=first, max_fuse=fuse, restart=True) as b:
        b.step(3)
        b.write_rooms(0, views)
        for chunk in (1, 1, 17, 1, 40, 2, 64, 1, 33):
            b.step(chunk)
            orc.run(rooms, seed, first, b.turn - chunk, chunk, threads=0, restart=True)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"n={n} fuse={fuse} after turn {b.turn}")
        assert int(rooms["games"].max()) >= 2                       # rooms did finish and start again


@pytest.mark.parametrize("n,rounds", [(4, 1), (3, 2), (8, 3), (12, 2)])
def test_two_truths_from_random_states(dsl_tt, n, rounds):
    R, seed, first = 20000, 5, 1 << 40
    rng = np.random.default_rng(n + rounds)
    orc = _oracle(dsl_tt, n, rounds)
    views = _random_tt_views(orc, n, R, rng, rounds)
    rooms = views_as_oracle_rooms(orc, views)
    with RoomBatch([(GameTable(dsl_tt, rounds), n, R)], seed=seed, first_room=first, max_fuse=4) as b:
        b.step(2)
        b.write_rooms(0, views)
        for chunk in (1, 4, 1):
            b.step(chunk)
            orc.run(rooms, seed, first, b.turn - chunk, chunk, threads=0)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"tt n={n} after turn {b.turn}")


def _edge_case_views(n=8):
    def base(phase, prev):
        v = np.zeros(1, dtype=ROOM_VIEW_DTYPE)
        v["phase_id"], v["prev_phase_id"], v["phase0_done"], v["end_turn"] = phase, prev, 1, -1
        v["n_players"], v["pack"] = n, 1
        roles = [2, 2, 3, 4, 1, 1, 1, 1]
        for i, r in enumerate(roles):
            v["players"][0, i, :9] = [r, 2 if r == 2 else 1, 1, 0, 1, int(r != 1), int(r != 1), 0, 0]
        return v

    cases = []
    v = base(15, 14)                      # day vote complete with a 3-3 tie between players 5 and 3 -> 3 dies
    v["players"][0, :n, 9] = 1
    v["players"][0, :n, 10] = [5, 5, 5, 3, 3, 3, 6, 7]
    cases.append(("tie", v))
    v = base(12, 11)                      # detective phase done; wolves chose 5, doctor protects 5 -> nobody dies at 13
    v["players"][0, 0, 7:9] = [1, 5]; v["players"][0, 1, 7:9] = [1, 5]; v["players"][0, 2, 7:9] = [1, 5]
    v["players"][0, 3, 9:11] = [1, 6]; v["players"][0, 3, 7] = 1
    cases.append(("protected", v))
    v = base(10, 9)                       # doctor is dead: phase 11 has no target players
    v["players"][0, 2, 2] = 0; v["players"][0, 2, 4] = 0
    v["players"][0, 0, 9:11] = [1, 6]; v["players"][0, 1, 9:11] = [1, 6]
    cases.append(("dead doctor", v))
    v = base(9, 16)                       # two wolves, two villagers alive -> wolves win
    v["players"][0, 4:8, 2] = 0
    cases.append(("wolves win", v))
    v = base(9, 13)                       # no wolves alive -> village wins
    v["players"][0, :2, 2] = 0
    cases.append(("village wins", v))
    v = base(9, 13)                       # follows night -> day discussion (14)
    cases.append(("follows night", v))
    v = base(9, 16)                       # follows day -> next night (10)
    cases.append(("follows day", v))
    return np.concatenate([c[1] for c in cases])


def test_hand_built_edge_cases(dsl_ww):
    """Ties -> lowest id; protected victim survives; dead Doctor -> empty target set completes at once;
    wolves >= villagers and wolves == 0 route to phase 99; follows-day / follows-night routing."""
    n = 8
    orc = _oracle(dsl_ww, n)
    tb = GameTable(dsl_ww)
    views = _edge_case_views(n)
    cases = range(len(views))
    rooms = views_as_oracle_rooms(orc, views)
    with RoomBatch([(tb, n, len(views))], seed=1, max_fuse=1) as b:
        b.step(3)
        b.write_rooms(0, views)
        traj = []
        for t in range(3, 7):
            b.step(1)
            orc.run(rooms, 1, 0, t, 1)
            got = b.read_rooms()
            assert_views_equal(got, oracle_rooms_as_views(orc, rooms), f"turn {t}")
            traj.append(got.copy())
    first = traj[0]
    assert first["phase_id"].tolist()[:1] == [16] and first["players"][0, 2, 2] == 0 and first["players"][0, 4, 2] == 1   # tie: 3 dies, 5 lives
    assert first["phase_id"][1] == 13 and first["players"][1, :n, 2].sum() == 8                                        # protected
    assert first["phase_id"][3] == 99 and first["phase_id"][4] == 99
    assert first["phase_id"][5] == 14 and first["phase_id"][6] == 10
    assert first["end_turn"][3] == 3
    dd = [int(x["phase_id"][2]) for x in traj]
    assert 11 in dd and dd[dd.index(11) + 1] == 12 if dd.index(11) + 1 < len(dd) else True   # empty target set: one turn only
