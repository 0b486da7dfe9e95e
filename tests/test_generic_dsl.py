"""Generic DSL coverage, first slice: the rest of the target-condition grammar the reference's DSL
generator is told to write (agent/prompt/dsl_phases_generation_prompt.txt:106-150) - `in [..]`, `not in`, `!=`,
`<`, `<=`, `>`, `>=` over the packs' declared `num` fields, `or` - and the three `wait_for` kinds.

* the product compiler (ge_table.cpp) and the oracle compiler (oracle/dsl_table.py) agree on the clause form of
  random conditions, and on what is an error;
* GPU (-m gpu): rooms stepped on DSL variants and on randomly conditioned DSLs equal the oracle, every room;
  the variants' reference-run goldens (traj_variant_*.json) are covered by the golden tests in test_gpu_parity.py
  and test_oracle_golden.py through conftest.golden_dsl."""
import copy
import random

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import load_dsl
from game_engine_amd import GameTable, GeError
from oracle import cond_gen, dsl_table as T, dsl_variants

GAMES = {1: "werewolf-(mafia)", 2: "two-truths-and-a-lie"}


def _oracle_clauses(p):
    return [[(1 if l.kind == "base" else 2, int(l.negate), sum(1 << b for b in l.bases), l.num, l.lo, l.hi) for l in c]
            for c in p.clauses]


def _assert_same_table(dsl, rounds=1):
    tb, ot = GameTable(dsl, rounds), T.compile_dsl(dsl, rounds)
    for r, p in zip(tb.rows(), ot.phases):
        assert (r["phase_id"], r["completion"], r["act"], r["effect"]) == (p.id, p.completion, p.act, p.effect)
        assert r["clauses"] == _oracle_clauses(p) and r["generic"] == p.generic, (p.id, r["clauses"], _oracle_clauses(p))
        if not p.generic:
            assert r["terms"] == [(t.base, int(t.negate)) for t in p.terms]
    return tb, ot


def _acts(dsl):
    return {p.id: p.act for p in T.compile_dsl(dsl).phases}


@pytest.mark.parametrize("name", sorted(dsl_variants.VARIANTS))
def test_variants_compile_to_the_same_clause_form(name):
    game, builder, rounds = dsl_variants.VARIANTS[name]
    tb, ot = _assert_same_table(builder(load_dsl(game)), rounds)
    assert any(r["generic"] for r in tb.rows()) or name == "ww_minimal_schema"      # (that one varies the schema, not the conditions)


@settings(max_examples=250, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(0, 2**32 - 1), st.sampled_from([1, 2]))
def test_random_conditions_compile_alike(seed, pack):
    """Random conditions in the grammar: both compilers accept and produce the same clause form, or both refuse."""
    base = load_dsl(GAMES[pack])
    d = cond_gen.randomize_dsl(random.Random(seed), base, pack, _acts(base))
    try:
        ot = T.compile_dsl(d)
    except T.DslError:
        with pytest.raises(GeError) as e:
            GameTable(d)
        assert e.value.status == -2
        return
    _assert_same_table(d)


@pytest.mark.parametrize("cond,needle", [
    ("player.is_alive > 1", "non-numeric"),
    ("player.mood == 'angry'", "not in rule pack"),
    ("(player.role == 'Werewolf')", "parentheses"),
    ("player.role == 'Werewolf' and player.selected_target_id not in [1, 3]", "non-contiguous"),
    ("player.role in []", "empty list"),
    ("player.role == 'Werewolf' or player.role == 'Doctor'", "different player actions"),
    ("player.is_alive == true", "cannot classify"),
    ("player.role == 'Werewolf' and player.is_alive in [true, false]", "both values"),
    ("player.role == 'Werewolf' and player.selected_target_id in [1,3,5,7,9]", "too many condition alternatives"),
    ("player.role == 'Werewolf' and len(players) > 3", "unsupported condition"),
])
def test_outside_the_grammar_is_an_error_in_both_compilers(dsl_ww, cond, needle):
    d = copy.deepcopy(dsl_ww)
    d["phases"]["2"]["completion_criteria"]["target_players"]["condition"] = cond
    with pytest.raises(GeError) as e:
        GameTable(d)
    assert e.value.status == -2 and needle in str(e.value), str(e.value)
    with pytest.raises(T.DslError) as oe:
        T.compile_dsl(d)
    assert needle in str(oe.value)


def test_declared_fields_outside_the_pack_are_constants(dsl_ww):
    """A condition may test any declared scalar field; one the rule pack does not model keeps the template's value
    (the fixed policy never writes it), so its literal compiles to 'never' (empty base set) or 'always' (negated)."""
    d = dsl_variants.build("ww_extra_fields", dsl_ww)
    tb, ot = _assert_same_table(d)
    rows = {r["phase_id"]: r for r in tb.rows()}
    assert (1, 0, 0, 0, 0, 0) in rows[3]["clauses"][0]                      # tier != 'gold' : never
    assert rows[7]["clauses"][0].count((1, 1, 0, 0, 0, 0)) == 2             # suspicion >= 2, suspicion not in [4, 5] : always
    assert rows[4]["clauses"][0][1] == (1, 0, 0, 0, 0, 0) and rows[4]["clauses"][1][2] == (1, 1, 0, 0, 0, 0)
    assert tb.extra_fields == {"suspicion": 3, "tier": "gold"}
    for cond, needle in (("player.role == 'Doctor' and player.tier > 2", "non-numeric"),
                         ("player.role == 'Doctor' and player.name == 'Ann'", "not in rule pack"),
                         ("player.role == 'Doctor' and player.investigated_alignments == 1", "unsupported value")):
        bad = copy.deepcopy(d)
        bad["phases"]["3"]["completion_criteria"]["target_players"]["condition"] = cond
        with pytest.raises(GeError) as e:
            GameTable(bad)
        assert needle in str(e.value), str(e.value)
        with pytest.raises(T.DslError) as oe:
            T.compile_dsl(bad)
        assert needle in str(oe.value)


def test_a_field_the_phase_text_writes_is_not_folded(dsl_ww):
    """The generator prompt's own example `player.is_current_turn == true` (dsl_phases_generation_prompt.txt:121) is a
    field the Referee rewrites every turn.  Declared, outside the rule pack and named by a phase's text, it must be a
    compile error in both compilers - folded to the template's value it would target nobody (or everybody) for the
    whole game.  Declared but mentioned nowhere in the phase graph (nobody is told to write it) it still folds."""
    d = copy.deepcopy(dsl_ww)
    d["declaration"]["player_states"]["is_current_turn"] = {"type": "boolean", "description": "whose turn it is", "example": False}
    for tmpl in d["declaration"]["player_states_template"]["player_states"].values():
        tmpl["is_current_turn"] = False
    d["phases"]["7"]["completion_criteria"]["target_players"]["condition"] = "player.can_vote == true and player.is_alive == true and player.is_current_turn == false"
    _assert_same_table(d)                                                     # nobody writes it: a constant, as before
    d["phases"]["6"]["actions"][0]["description"] += " Then set is_current_turn to true for the next speaker."
    with pytest.raises(GeError) as e:
        GameTable(d)
    assert e.value.status == -2 and "is_current_turn" in str(e.value) and "may be written" in str(e.value)
    with pytest.raises(T.DslError) as oe:
        T.compile_dsl(d)
    assert "is_current_turn" in str(oe.value) and "may be written" in str(oe.value)


def test_minimal_schema_keeps_the_other_slots_as_engine_state(dsl_ww):
    """A Werewolf DSL that declares only name / role / team / is_alive / can_vote: both compilers bind five slots and leave
    the rest undeclared; rooms then have exactly those fields, and the rules still run (the reference-run goldens
    traj_variant_ww_minimal_schema_* pin that; there the policy reads targets and the Detective's memory off the action log)."""
    from game_engine_amd.stepper import view_to_agent_state
    from oracle.oracle import Oracle
    from parity_util import oracle_rooms_as_views
    d = dsl_variants.build("ww_minimal_schema", dsl_ww)
    tb, ot = _assert_same_table(d)
    assert [n for n in tb.field_names if n] == ["role", "team", "is_alive", "can_vote"]
    assert [s[0] for s in T.WW_SLOTS if ot.declared(s[0])] == ["role", "team", "is_alive", "can_vote"]
    orc = Oracle(d, 8)
    rooms = orc.init_rooms(1)
    orc.run(rooms, 1, 0, 0, 30)
    st = view_to_agent_state(tb, oracle_rooms_as_views(orc, rooms)[0])
    assert all(set(p) == {"role", "team", "is_alive", "can_vote"} for p in st["player_states"].values())
    assert any(not p["is_alive"] for p in st["player_states"].values())          # the night / day resolutions happened
    for cond in ("player.role == 'Doctor' and player.night_action_submitted == false", "player.selected_target_id > 0"):
        bad = copy.deepcopy(d)
        bad["phases"]["3"]["completion_criteria"]["target_players"]["condition"] = cond
        with pytest.raises(GeError):
            GameTable(bad)
        with pytest.raises(T.DslError):
            T.compile_dsl(bad)


def test_wait_for_kinds(dsl_ww):
    """All three wait_for kinds mean 'every target player' (prompt :138 Completion Logic); others are errors."""
    for wf in ("single_player_choice", "all_players_action", "multiple_players_action"):
        d = copy.deepcopy(dsl_ww)
        d["phases"]["7"]["completion_criteria"]["wait_for"] = wf
        assert [r for r in GameTable(d).rows() if r["phase_id"] == 7][0]["completion"] == 2
        T.compile_dsl(d)
    d["phases"]["7"]["completion_criteria"]["wait_for"] = "majority"
    with pytest.raises(GeError) as e:
        GameTable(d)
    assert "unknown wait_for" in str(e.value)
    with pytest.raises(T.DslError):
        T.compile_dsl(d)


# ------------------------------------------------------------------------------------------------ GPU
def _batch_equals_oracle(dsl, n, rooms, turns, seed, first, rounds=1, restart=True, mask=0):
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views
    orc = Oracle(dsl, n, rounds=rounds)
    want = orc.init_rooms(rooms)
    orc.run(want, seed, first, 0, turns, threads=0, restart=restart, human_mask=mask)
    with RoomBatch([(GameTable(dsl, rounds), n, rooms, mask)], seed=seed, first_room=first, restart=restart) as b:
        b.step(turns)
        assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, want), f"n={n} rooms={rooms}")
        return b.summary()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,rooms", [("ww_generic", 8, 40000), ("ww_generic", 12, 150000), ("ww_generic", 5, 3000),
                                          ("ww_extra_fields", 8, 70000), ("ww_extra_fields", 10, 9000),
                                          ("tt_generic", 4, 200000), ("tt_generic", 9, 30000),
                                          ("ww_minimal_schema", 8, 65536), ("ww_minimal_schema", 11, 9000)])
def test_variant_batches_equal_oracle(name, n, rooms):
    game, builder, rounds = dsl_variants.VARIANTS[name]
    s = _batch_equals_oracle(builder(load_dsl(game)), n, rooms, 120, 0xC0FFEE, 1 << 30, rounds)
    assert s["games_recycled"] >= (rooms // 2 if n <= 8 else 0) and s["turn"] == 120     # (nine players x two rounds outlast 120 turns)


@pytest.mark.gpu
@pytest.mark.parametrize("pack,n", [(1, 8), (1, 11), (2, 4), (2, 10)])
def test_randomly_conditioned_dsls_equal_oracle(pack, n):
    """Twelve random re-conditionings of the game per case: every room of a 4 000-room batch equals the oracle
    after 90 steady-state turns (the GENERIC kernel builds, whatever the batch size)."""
    base = load_dsl(GAMES[pack])
    acts = _acts(base)
    done = generic = 0
    for seed in range(100):
        d = cond_gen.randomize_dsl(random.Random(1000 * pack + 37 * n + seed), base, pack, acts)
        try:
            ot = T.compile_dsl(d)
        except T.DslError:
            continue
        generic += any(p.generic for p in ot.phases)
        _batch_equals_oracle(d, n, 4000, 90, seed, 77 * seed)
        done += 1
        if done == 12:
            break
    assert done == 12 and generic >= 8


@pytest.mark.gpu
def test_host_driven_players_on_a_generic_dsl():
    """ge_batch_inject_actions checks the clause form too: refusals and effects agree with the oracle."""
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views
    for name, n, mask in (("ww_generic", 8, 0b11), ("tt_generic", 4, 0b101)):
        game, builder, rounds = dsl_variants.VARIANTS[name]
        dsl = builder(load_dsl(game))
        orc = Oracle(dsl, n, rounds=rounds)
        R, seed, first = 3000, 5, 999
        rng = np.random.default_rng(n)
        rooms = orc.init_rooms(R)
        humans = [i + 1 for i in range(n) if (mask >> i) & 1]
        ok = bad = 0
        with RoomBatch([(GameTable(dsl, rounds), n, R, mask)], seed=seed, first_room=first, max_fuse=1) as b:
            for t in range(70):
                k = 1500
                rr = rng.integers(0, R, size=k).astype(np.uint64)
                pl = rng.choice(humans, size=k).astype(np.uint32)
                ch = rng.integers(0, n + 2, size=k).astype(np.uint32)
                want = np.array([0 if orc.inject(rooms, int(r), int(p), int(c)) else -1 for r, p, c in zip(rr, pl, ch)], dtype=np.int32)
                got = b.inject_actions(rr, pl, ch)
                assert got.tolist() == want.tolist(), (name, t)
                ok += int((got == 0).sum()); bad += int((got != 0).sum())
                b.step(1)
                orc.run(rooms, seed, first, t, 1, threads=0, human_mask=mask)
                assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"{name} turn {t}")
        assert ok > 200 and bad > 200


@pytest.mark.gpu
@pytest.mark.parametrize("n,rounds", [(4, 15), (7, 9), (12, 11)])
def test_two_truths_numeric_ranges_over_whole_value_ranges(dsl_tt, n, rounds):
    """The numeric literals' SWAR compares (ge_device.h range_*) at the edges of every field: scores up to 255 against bounds
    on both sides of 128 (half-word lanes), rounds up to 14 (byte lanes), subsets of the 2-bit values - from random,
    mostly unreachable states (ge_batch_write_rooms), in both kernel builds (lone-wavefront / large-batch), against the oracle."""
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views, views_as_oracle_rooms
    from test_gpu_fuzz import _random_tt_views
    conds = [("player.is_speaker == true and player.total_score >= 128 or player.is_speaker == true and player.rounds_as_speaker <= 13 and player.total_score < 127",
              "player.is_speaker == true and player.lie_index not in [2, 3] or player.is_speaker == true and player.total_score == 230",
              "player.is_speaker == false and player.total_score <= 200 and player.rounds_as_speaker >= 3 or player.is_speaker == false and player.vote_choice in [0, 2]"),
             ("player.is_speaker == true and player.rounds_as_speaker > 13 or player.is_speaker == true and player.total_score != 0",
              "player.is_speaker == true and player.total_score > 229 or player.is_speaker == true and player.lie_index >= 2",
              "player.is_speaker == false and player.vote_choice != 3 and player.total_score >= 129 and player.total_score <= 131 or player.is_speaker == false and player.rounds_as_speaker == 14")]
    for ci, (c2, c3, c5) in enumerate(conds):
        d = copy.deepcopy(dsl_tt)
        for pid, c in (("2", c2), ("3", c3), ("5", c5)):
            d["phases"][pid]["completion_criteria"]["target_players"]["condition"] = c
        orc = Oracle(d, n, rounds=rounds)
        tb = GameTable(d, rounds)
        assert sum(r["generic"] for r in tb.rows()) == 3
        for R in (5000, 90000):
            rng = np.random.default_rng(100 * n + ci)
            views = _random_tt_views(orc, n, R, rng, rounds)
            p = views["players"]
            # (kept below the counters' ends: a score byte or a rounds nibble that overflows is outside the packed model)
            p[:, :n, 7] = rng.choice([0, 1, 126, 127, 128, 129, 130, 131, 199, 200, 201, 229, 230], size=(R, n))
            p[:, :n, 8] = rng.integers(0, 15, (R, n))
            rooms = views_as_oracle_rooms(orc, views)
            seed, first = 31 + ci, 1 << 33
            with RoomBatch([(tb, n, R)], seed=seed, first_room=first, max_fuse=3) as b:
                b.step(2)
                b.write_rooms(0, views)
                for chunk in (1, 3, 2):
                    b.step(chunk)
                    orc.run(rooms, seed, first, b.turn - chunk, chunk, threads=0)
                    assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"tt x{n} conds {ci} R={R} after turn {b.turn}")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 8, 12])
def test_werewolf_numeric_ranges_over_whole_value_ranges(dsl_ww, n):
    """selected_target_id (a nibble per player, byte lanes; 12 players: three words) against bounds up to 15 and beyond."""
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views, views_as_oracle_rooms
    from test_gpu_fuzz import _random_ww_views
    conds = {"2": "player.role == 'Werewolf' and player.is_alive == true and player.selected_target_id >= 9 or player.role == 'Werewolf' and player.selected_target_id in [0, 1, 2]",
             "3": "player.role == 'Doctor' and player.is_alive == true and player.selected_target_id <= 15",
             "4": "player.role == 'Detective' and player.selected_target_id > 11 or player.role == 'Detective' and player.team not in ['werewolves'] and player.selected_target_id != 7",
             "7": "player.can_vote == true and player.is_alive == true and player.selected_target_id < 12 and player.role in ['Villager', 'Doctor', 'Detective', 'Werewolf']"}
    d = copy.deepcopy(dsl_ww)
    for pid, c in conds.items():
        d["phases"][pid]["completion_criteria"]["target_players"]["condition"] = c
    orc = Oracle(d, n)
    tb = GameTable(d)
    assert sum(r["generic"] for r in tb.rows()) == 4
    for R in (6000, 100000):
        rng = np.random.default_rng(n)
        views = _random_ww_views(orc, n, R, rng, consistent=False)
        views["players"][:, :n, 8] = rng.integers(0, min(n, 15) + 1, (R, n))
        rooms = views_as_oracle_rooms(orc, views)
        seed, first = 77, 12345
        with RoomBatch([(tb, n, R)], seed=seed, first_room=first, max_fuse=4) as b:
            b.step(3)
            b.write_rooms(0, views)
            for chunk in (1, 4, 2):
                b.step(chunk)
                orc.run(rooms, seed, first, b.turn - chunk, chunk, threads=0)
                assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"ww x{n} R={R} after turn {b.turn}")


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,rooms", [("ww_generic", 8, 3000), ("ww_generic", 8, 140001), ("ww_generic", 12, 140001),
                                          ("tt_generic", 4, 3000), ("tt_generic", 4, 140001), ("tt_generic", 9, 70001)])
def test_generic_tables_in_single_turn_launches(name, n, rooms):
    """max_fuse = 1 on a table with generic conditions: the GENERIC single-turn builds (round 5; until then such a batch ran the
    fused-loop kernel for one turn), lone-wavefront and large-batch - 75 launches == the oracle, room by room."""
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views
    game, builder, rounds = dsl_variants.VARIANTS[name]
    dsl = builder(load_dsl(game))
    orc = Oracle(dsl, n, rounds=rounds)
    seed, first, turns = 3, 1 << 31, 75
    want = orc.init_rooms(rooms)
    orc.run(want, seed, first, 0, turns, threads=0, restart=True)
    with RoomBatch([(GameTable(dsl, rounds), n, rooms)], seed=seed, first_room=first, max_fuse=1, restart=True) as b:
        b.step(turns)
        assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, want), f"{name} n={n} rooms={rooms}, single-turn launches")


@pytest.mark.gpu
def test_mixed_batch_with_a_generic_table_in_single_turn_launches(dsl_ww):
    """A mixed batch one of whose tables is generic, stepped one turn per launch (the mixed GENERIC single-turn build)."""
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    from parity_util import assert_views_equal, oracle_rooms_as_views
    game, builder, rounds = dsl_variants.VARIANTS["tt_generic"]
    dsl_g = builder(load_dsl(game))
    segs = [(GameTable(dsl_ww), 12, 50001), (GameTable(dsl_g, rounds), 4, 60000)]
    seed, first, turns = 11, 12345, 70
    with RoomBatch(segs, seed=seed, first_room=first, max_fuse=1, restart=True) as b:
        b.step(turns)
        got = b.read_rooms()
    lo = 0
    for (tb, n, r), (d, rd) in zip(segs, ((dsl_ww, 1), (dsl_g, rounds))):
        orc = Oracle(d, n, rounds=rd)
        want = orc.init_rooms(r)
        orc.run(want, seed, first + lo, 0, turns, threads=0, restart=True)
        assert_views_equal(got[lo:lo + r], oracle_rooms_as_views(orc, want), f"segment x{n}")
        lo += r
