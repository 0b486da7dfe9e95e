"""The TypeScript/Node host (game_engine_amd/node: N-API addon over the same C ABI).
CPU: the addon loads, compiles the DSL, and refuses to create a batch without a GPU.
GPU (-m gpu): steps golden rooms turn by turn through node and checks the AgentState it returns."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLD, ROOT

NODE_DIR = os.path.join(ROOT, "game_engine_amd", "node")
needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(os.path.join(NODE_DIR, "ge_addon.node")),
                                reason="node or the built addon is not available")


def _run(*args):
    out = subprocess.run(["node", os.path.join(NODE_DIR, "selftest.js"), *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@needs_node
def test_addon_loads_and_compiles_without_gpu():
    r = _run(os.path.join(GOLD, "dsl", "werewolf-(mafia).json"), os.path.join(GOLD, "traj_werewolf_n8.json"))
    assert r["phases"] == 18 and r["pack"] == 1
    if r["devices"] == 0:
        assert r["noDevice"] == "GE-3"          # GE_ERR_NO_DEVICE: no CPU fallback in the JS host either


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("dsl,gold", [("werewolf-(mafia).json", "traj_werewolf_n8.json"),
                                      ("two-truths-and-a-lie.json", "traj_two_truths_and_a_lie_n4.json")])
def test_node_host_matches_golden(dsl, gold):
    r = _run(os.path.join(GOLD, "dsl", dsl), os.path.join(GOLD, gold))
    g = json.load(open(os.path.join(GOLD, gold)))
    assert r["devices"] >= 1 and r["checked"] == sum(len(c["turns"]) for c in g["cases"])
    assert r["turn"] == 64 and r["finished"] > 0 and isinstance(r["sample"], str)
    # the JS host renders the same backend tool calls as the Python host for the same traced room
    from conftest import load_dsl
    from game_engine_amd import GameTable, RoomBatch
    from game_engine_amd.toolcalls import turn_tool_calls
    tb = GameTable(load_dsl(dsl[:-5]))
    with RoomBatch([(tb, g["n_players"], 1)], seed=5, first_room=9, max_fuse=1, trace=True) as b:
        before = b.read_rooms(0, 1)[0].copy()
        for t in range(30):
            b.step(1)
            after = b.read_rooms(0, 1)[0].copy()
            want = turn_tool_calls(tb, before, after, b.read_events(0, 1)[0][0])
            assert r["calls"][t] == json.loads(json.dumps(want)), t
            before = after


@needs_node
@pytest.mark.gpu
def test_single_room_service_over_http():
    """One LangGraph thread served by an N=1 traced batch (RoomService): the trajectory equals the
    Python host's for the same seed / room index, and the log-shaped AgentState parts fill up."""
    from conftest import load_dsl
    from game_engine_amd import GameTable, RoomBatch
    out = subprocess.run(["node", os.path.join(NODE_DIR, "selftest_service.js"), os.path.join(GOLD, "dsl", "werewolf-(mafia).json")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    tb = GameTable(load_dsl("werewolf-(mafia)"))
    with RoomBatch([(tb, 8, 1)], seed=7, first_room=int(r["room"]), max_fuse=1) as b:
        phases = [0]
        for t in range(70):
            b.step(1)
            phases.append(int(b.read_rooms(0, 1)[0]["phase_id"]))
        alive = [int(x) for x in b.read_rooms(0, 1)[0]["players"][:8, 2]]
    assert r["phases"] == phases and r["alive"] == alive and phases[-1] == 99
    assert r["name1"] == "Bot 1" and r["finalPhase"].startswith("Game Over")
    assert r["notes"] > 10 and r["acts"] > 10 and r["ui"] > 100
