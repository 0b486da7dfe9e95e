"""Host helpers that keep the reference's names and contracts (CPU)."""
import asyncio
import os
import sys

import pytest

from conftest import GOLD, load_dsl
from game_engine_amd import initialize_player_states_from_dsl, load_dsl_by_gamename

REF = "/root/reference/agent/tools/utils.py"


def test_load_dsl_by_gamename_contract(tmp_path):
    assert load_dsl_by_gamename("", str(tmp_path)) == {}                       # utils.py:559-561
    assert load_dsl_by_gamename("nope", str(tmp_path)) == {}                   # utils.py:576-578
    d = load_dsl_by_gamename("werewolf-(mafia)", os.path.join(GOLD, "dsl"))    # the JSON form of the same document
    assert set(d) == {"declaration", "phases"} and len(d["phases"]) == 18
    (tmp_path / "mini.yaml").write_text("declaration: {min_players: 2}\nphases:\n  0: {name: Intro}\n")
    y = load_dsl_by_gamename("mini", str(tmp_path))
    assert y["phases"][0]["name"] == "Intro"                                   # YAML int keys, as in the reference


def test_initialize_player_states(dsl_ww, dsl_tt):
    players = [{"name": "Alice"}, {"name": "Bob"}, {}]
    ps = initialize_player_states_from_dsl(dsl_ww, players)
    assert list(ps) == ["1", "2", "3"] and ps["1"]["name"] == "Alice" and ps["3"]["name"] == "Player 3"
    assert ps["2"]["is_alive"] is True and ps["2"]["selected_target_id"] == 0 and ps["2"]["investigated_alignments"] == {}
    assert initialize_player_states_from_dsl({}, players) == {}
    no_tmpl = {"declaration": {"player_states": {"score": {"type": "num"}, "ok": {"type": "boolean"}, "tag": {"type": "string", "example": "x"}}}}
    assert initialize_player_states_from_dsl(no_tmpl, [{"name": "Z"}]) == {"1": {"score": 0, "ok": True, "tag": "x", "name": "Z"}}


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("game", ["werewolf-(mafia)", "two-truths-and-a-lie"])
def test_initialize_player_states_equals_reference(game):
    from oracle.refharness import walker
    for p in (walker._STANDINS, os.path.join(walker.REFERENCE_ROOT, "agent")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True
    import tools.utils as ut
    import yaml
    with open(os.path.join(walker.REFERENCE_ROOT, "games", f"{game}.yaml"), encoding="utf-8") as f:
        dsl_yaml = yaml.safe_load(f)
    players = [{"name": f"P{i}"} for i in range(5)] + [{}]
    want = asyncio.run(ut.initialize_player_states_from_dsl(dsl_yaml, players))
    assert initialize_player_states_from_dsl(dsl_yaml, players) == want            # YAML (int keys) form
    assert initialize_player_states_from_dsl(load_dsl(game), players) == want      # JSON (str keys) form
