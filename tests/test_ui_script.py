"""Deterministic UI script: audience groups from declaration.audience_groups and the
frontend tool calls of a phase, derived from oracle-stepped rooms (CPU)."""
import pytest

from conftest import load_dsl
from game_engine_amd import GameTable
from game_engine_amd.stepper import view_to_agent_state
from game_engine_amd.ui_script import audience_groups, compile_criteria, ui_tool_calls
from parity_util import oracle_rooms_as_views

FRONTEND_TOOLS = {"createCharacterCard", "createPhaseIndicator", "createTextDisplay", "createVotingPanel",
                  "createAvatarSet", "createResultDisplay", "createTimer", "createDeathMarker", "markPlayerDead",
                  "clearCanvas", "createScoreBoard", "createTurnIndicator", "createStatementBoard",
                  "createTextInputPanel"}          # subset of FRONTEND_TOOL_ALLOWLIST, agent/game_agent_v2.py:144-192


def _room_states(dsl, n, turns, seed=2):
    from oracle.oracle import Oracle
    orc = Oracle(dsl, n)
    tb = GameTable(dsl)
    rooms = orc.init_rooms(1)
    out = []
    for t in range(turns):
        orc.run(rooms, seed, 0, t, 1)
        out.append(view_to_agent_state(tb, oracle_rooms_as_views(orc, rooms)[0]))
    return out


def test_criteria_language():
    p = compile_criteria("player.role in ['Doctor', 'Detective'] and player.is_alive == true")
    assert p({"role": "Doctor", "is_alive": True}) and not p({"role": "Doctor", "is_alive": False})
    assert not p({"role": "Villager", "is_alive": True})
    assert compile_criteria("player.team != 'werewolves'")({"team": "villagers"})
    with pytest.raises(ValueError):
        compile_criteria("len(players) > 3")


def test_audience_groups_follow_the_dsl(dsl_ww):
    for st in _room_states(dsl_ww, 8, 40):
        ps = st["player_states"]
        g = audience_groups(dsl_ww, ps)
        assert set(g) == {"werewolves", "villagers", "alive_players", "dead_players", "special_roles",
                          "night_actors", "voters", "secret_holders"}                     # ww:138-165
        alive = [p for p in ps if ps[p]["is_alive"]]
        assert g["alive_players"] == alive and sorted(g["alive_players"] + g["dead_players"], key=int) == sorted(ps, key=int)
        assert g["werewolves"] == [p for p in alive if ps[p]["team"] == "werewolves"]
        assert g["special_roles"] == [p for p in alive if ps[p]["role"] in ("Doctor", "Detective")]
        assert g["voters"] == [p for p in alive if ps[p]["can_vote"]]


def test_ui_script_of_werewolf_phases(dsl_ww):
    seen = {}
    for st in _room_states(dsl_ww, 8, 60):
        calls = ui_tool_calls(dsl_ww, st)
        assert calls and all(c["name"] in FRONTEND_TOOLS for c in calls)
        want = [t for a in dsl_ww["phases"][str(st["current_phase_id"])]["actions"] for t in a["tools"]]
        got_tools = [c["name"] for c in calls]
        assert [t for t in dict.fromkeys(got_tools)] == [t for t in dict.fromkeys(want)]   # DSL order, tools kept
        for c in calls:
            a = c["args"]
            if c["name"] != "clearCanvas":
                assert a["audience_type"] is True or (a["audience_type"] is False and isinstance(a["audience_ids"], list))
        seen[st["current_phase_id"]] = (st, calls)
    ps, calls = seen[1][0]["player_states"], seen[1][1]                  # Role Assignment: one private card per player
    cards = [c for c in calls if c["name"] == "createCharacterCard"]
    assert [c["args"]["audience_ids"] for c in cards] == [[p] for p in sorted(ps, key=int)]
    assert [c["args"]["role"] for c in cards] == [ps[p]["role"] for p in sorted(ps, key=int)]
    st, calls = seen[10] if 10 in seen else seen[2]                      # werewolves choose: private panel + others wait
    ps = st["player_states"]
    wolves = [p for p in sorted(ps, key=int) if ps[p]["team"] == "werewolves" and ps[p]["is_alive"]]
    panel = next(c for c in calls if c["name"] == "createVotingPanel")
    wait = next(c for c in calls if c["name"] == "createTextDisplay")
    assert panel["args"]["audience_ids"] == wolves
    assert set(wait["args"]["audience_ids"]).isdisjoint(wolves) and all(ps[p]["is_alive"] for p in wait["args"]["audience_ids"])
    st, calls = seen[7] if 7 in seen else seen[15]                       # day vote: panel for the eligible voters
    ps = st["player_states"]
    panel = next(c for c in calls if c["name"] == "createVotingPanel")
    assert panel["args"]["audience_ids"] == [p for p in sorted(ps, key=int) if ps[p]["is_alive"] and ps[p]["can_vote"]]
    assert next(c for c in seen[0][1] if c["name"] == "createAvatarSet")["args"]["audience_type"] is True


def test_ui_script_of_two_truths(dsl_tt):
    for st in _room_states(dsl_tt, 4, 30):
        calls = ui_tool_calls(dsl_tt, st)
        assert all(c["name"] in FRONTEND_TOOLS for c in calls)
        if st["current_phase_id"] == 2:                                   # private statement input for the speaker
            ps = st["player_states"]
            box = next(c for c in calls if c["name"] == "createTextInputPanel")
            assert box["args"]["audience_type"] is True or box["args"]["audience_ids"] == [p for p in ps if ps[p]["is_speaker"]]


def test_js_ui_script_equals_python(dsl_ww, dsl_tt, tmp_path):
    """The TypeScript host renders the same UI script (game_engine_amd/node/ui_script.js)."""
    import json, os, shutil, subprocess
    from conftest import ROOT
    if shutil.which("node") is None:
        pytest.skip("node is not available")
    cases = []
    for dsl, n in ((dsl_ww, 8), (dsl_tt, 4)):
        for st in _room_states(dsl, n, 45):
            cases.append({"dsl": dsl, "state": st, "want": ui_tool_calls(dsl, st)})
    inp = tmp_path / "cases.json"
    inp.write_text(json.dumps(cases))
    js = ("const {uiToolCalls}=require(process.argv[1]);const c=JSON.parse(require('fs').readFileSync(process.argv[2],'utf8'));"
          "let bad=0;c.forEach((x,i)=>{if(JSON.stringify(uiToolCalls(x.dsl,x.state))!==JSON.stringify(x.want)){bad++;console.error('case',i);}});"
          "console.log(JSON.stringify({n:c.length,bad}));")
    out = subprocess.run(["node", "-e", js, os.path.join(ROOT, "game_engine_amd", "node", "ui_script.js"), str(inp)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip())
    assert r["n"] == len(cases) and r["bad"] == 0, out.stderr[:500]
