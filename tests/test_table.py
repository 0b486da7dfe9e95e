"""DSL -> phase table: the product's compiler (ge_table_compile_json, C++) against the oracle's
(oracle/dsl_table.py), on the reference's two shipped games.  CPU only (no kernels run)."""
import copy
import json

import pytest

from conftest import load_dsl
from game_engine_amd import GameTable, GeError
from oracle import dsl_table as T

GAMES = ["werewolf-(mafia)", "two-truths-and-a-lie"]


@pytest.mark.parametrize("game", GAMES)
def test_product_table_equals_oracle_table(game):
    dsl = load_dsl(game)
    ot = T.compile_dsl(dsl)
    pt = GameTable(dsl)
    assert pt.pack == ot.pack and pt.n_phases == len(ot.phases)
    for prow, op in zip(pt.rows(), ot.phases):
        assert prow["phase_id"] == op.id and prow["name"] == op.name[:63]
        assert (prow["completion"], prow["act"], prow["effect"]) == (op.completion, op.act, op.effect)
        assert prow["terms"] == [(t.base, int(t.negate)) for t in op.terms]
        assert prow["branches"] == [(b.resolver, b.target_idx) for b in op.branches]


def test_werewolf_phase_graph(dsl_ww):
    """18 phases, ids 0-16 and 99; player_action phases 2,3,4,7,10,11,12,15."""
    rows = GameTable(dsl_ww).rows()
    assert [r["phase_id"] for r in rows] == list(range(17)) + [99]
    assert [r["phase_id"] for r in rows if r["completion"] == 2] == [2, 3, 4, 7, 10, 11, 12, 15]
    assert [r["phase_id"] for r in rows if r["completion"] == 1] == [6, 14]
    nine = rows[9]
    assert [(res, rows[t]["phase_id"]) for res, t in nine["branches"]] == [(1, 99), (2, 99), (3, 10), (4, 14)]
    assert rows[17]["branches"] == []
    tb = GameTable(dsl_ww)
    assert [tb.role_name(i) for i in range(5)] == ["", "Villager", "Werewolf", "Doctor", "Detective"]
    assert list(tb.c.init_fields[:9]) == [0, 0, 1, 0, 1, 0, 0, 0, 0]      # ww:75-86 template


def test_two_truths_phase_graph(dsl_tt):
    rows = GameTable(dsl_tt).rows()
    assert [r["phase_id"] for r in rows] == list(range(9)) + [99]
    assert [r["phase_id"] for r in rows if r["completion"] == 2] == [2, 3, 5]
    assert [(res, rows[t]["phase_id"]) for res, t in rows[8]["branches"]] == [(5, 99), (6, 1)]
    assert list(GameTable(dsl_tt).c.init_fields[:9]) == [0, 0, 0, 0, 1, 0, 0, 0, 0]


def test_int_and_str_phase_keys_compile_alike(dsl_ww):
    """utils.py:29 / v2:1179-1188 accept int and str phase keys; so does the compiler."""
    a = GameTable(dsl_ww).rows()
    d = copy.deepcopy(dsl_ww)
    d["phases"] = {int(k): v for k, v in d["phases"].items()}
    assert GameTable(d).rows() == a


@pytest.mark.parametrize("mutate,needle", [
    (lambda d: d["phases"]["9"]["next_phase"].update({"When the moon is full": {"id": 99, "name": "x"}}), "no branch resolver"),
    (lambda d: d["phases"]["2"]["completion_criteria"]["target_players"].update(condition="player.mood == 'angry'"), "not in rule pack"),
    (lambda d: d["phases"]["2"]["completion_criteria"]["target_players"].update(condition="len(players) > 3"), "unsupported condition"),
    (lambda d: d["phases"]["3"].update(next_phase={"id": 55, "name": "nowhere"}), "not in phases"),
    (lambda d: d["phases"]["3"]["completion_criteria"].update(type="dice_roll"), "unknown completion type"),
    (lambda d: d["declaration"].pop("player_states"), "no rule pack"),
    (lambda d: d.pop("phases"), "needs top-level"),
])
def test_bad_dsl_is_an_error_not_a_guess(dsl_ww, mutate, needle):
    d = copy.deepcopy(dsl_ww)
    mutate(d)
    with pytest.raises(GeError) as e:
        GameTable(d)
    assert e.value.status == -2 and needle in str(e.value)
    with pytest.raises(T.DslError):
        T.compile_dsl(d)


def test_empty_dsl():
    with pytest.raises(GeError):
        GameTable({})


def test_reference_draft_dsl_binds_its_own_field_names():
    """game_draft/werewolf-(mafia).yaml (fixture tests/golden/dsl/draft-werewolf-(mafia).json) is the same game under
    a different player_states schema: has_night_action / known_alignments / wolf_chat_enabled, no selected_target_id,
    no has_secret_role, no night_action_submitted; wolves are addressed by team; two terminal phases; a branch key that
    names a phase ("follows Dawn Reveal").  Both compilers bind the pack's slots to the declared names and agree."""
    from conftest import load_dsl
    d = load_dsl("draft-werewolf-(mafia)")
    tb, ot = GameTable(d), T.compile_dsl(d)
    assert tb.pack == ot.pack == T.PACK_WEREWOLF
    assert tb.field_names[:11] == ["role", "team", "is_alive", "role_revealed", "can_vote", "", "has_night_action", "", "",
                                   "known_alignments", "wolf_chat_enabled"]
    assert [ot.declared(s[0]) or "" for s in T.WW_SLOTS] == tb.field_names[:11]
    rows = {r["phase_id"]: r for r in tb.rows()}
    assert sorted(rows) == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 98, 99]
    for r, p in zip(tb.rows(), ot.phases):
        assert (r["phase_id"], r["completion"], r["act"], r["effect"], r["terms"]) == \
            (p.id, p.completion, p.act, p.effect, [(t.base, int(t.negate)) for t in p.terms])
        assert r["branches"] == [(b.resolver, b.target_idx) for b in p.branches]
    assert (rows[2]["act"], rows[3]["act"], rows[4]["act"], rows[8]["act"]) == (T.ACT_WOLF_TARGET, T.ACT_DOCTOR_PROTECT, T.ACT_DETECTIVE, T.ACT_DAY_VOTE)
    assert (rows[1]["effect"], rows[2]["effect"], rows[6]["effect"], rows[9]["effect"]) == \
        (T.EFF_ASSIGN_ROLES, T.EFF_NIGHT_BEGIN, T.EFF_NIGHT_RESOLVE, T.EFF_DAY_RESOLVE)
    idx = {r["phase_id"]: i for i, r in enumerate(tb.rows())}
    assert rows[10]["branches"] == [(T.RES_WOLVES_ZERO, idx[98]), (T.RES_WOLVES_GE_VILLAGERS, idx[99]),
                                    (T.RES_FOLLOWS_NIGHT, idx[7]), (T.RES_OTHERWISE, idx[2])]
    assert rows[98]["branches"] == rows[99]["branches"] == []
    # a field of the shipped schema that this DSL does not declare is not silently accepted in a condition
    bad = copy.deepcopy(d)
    bad["phases"]["3"]["completion_criteria"]["target_players"]["condition"] = "player.role == 'Doctor' and player.night_action_submitted == false"
    with pytest.raises(GeError) as e:
        GameTable(bad)
    assert "not in rule pack" in str(e.value)
    with pytest.raises(T.DslError):
        T.compile_dsl(bad)
    # a derived slot in a condition: wolf_chat_enabled is the team test
    ok = copy.deepcopy(d)
    ok["phases"]["2"]["completion_criteria"]["target_players"]["condition"] = "player.wolf_chat_enabled == true and player.is_alive == true"
    assert [r for r in GameTable(ok).rows() if r["phase_id"] == 2][0]["terms"] == [(7, 0), (0, 0)]
    assert [(t.base, t.negate) for t in T.compile_dsl(ok).by_id(2).terms] == [(7, False), (0, False)]
    # two names of one slot in the same declaration: which one the rules write would be a guess
    bad = copy.deepcopy(d)
    bad["declaration"]["player_states"]["night_action_eligible"] = {"type": "boolean", "example": False, "description": "x"}
    with pytest.raises(GeError) as e:
        GameTable(bad)
    assert "bind to one state slot" in str(e.value)
    with pytest.raises(T.DslError):
        T.compile_dsl(bad)
    # a branch key that names no resolving phase stays an error
    bad = copy.deepcopy(d)
    nx = bad["phases"]["10"]["next_phase"]
    bad["phases"]["10"]["next_phase"] = {("If this check follows Day Discussion" if "Dawn" in k else k): v for k, v in nx.items()}
    with pytest.raises(GeError) as e:
        GameTable(bad)
    assert "no branch resolver" in str(e.value)
    with pytest.raises(T.DslError):
        T.compile_dsl(bad)


def test_draft_fixture_is_the_reference_file():
    """The fixture is yaml.safe_load of the reference's file (checked where the reference checkout exists)."""
    import os
    path = "/root/reference/game_draft/werewolf-(mafia).yaml"
    if not os.path.exists(path):
        pytest.skip("needs the reference checkout (build container)")
    import json
    import yaml
    from conftest import load_dsl
    with open(path, encoding="utf-8") as f:
        assert json.loads(json.dumps(yaml.safe_load(f))) == load_dsl("draft-werewolf-(mafia)")
