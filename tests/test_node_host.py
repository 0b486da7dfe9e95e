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
def test_initialize_players_twin(tmp_path, monkeypatch):
    """initializePlayers = POST /api/games/initialize-players (route.ts:83-166) without HTTP: the
    template copy must agree with the agent-side initialize_player_states_from_dsl (utils.py:584-653)
    on every declared field; name / id / isHost come from the room's players."""
    from game_engine_amd import initialize_player_states_from_dsl
    (tmp_path / "Werewolf-(Mafia).yaml").write_text("x: 1\n")
    (tmp_path / "two-truths-and-a-lie.yaml").write_text("x: 1\n")
    monkeypatch.setenv("GE_TEST_GAMES_DIR", str(tmp_path))
    dsl_path = os.path.join(GOLD, "dsl", "werewolf-(mafia).json")
    r = _run(dsl_path, os.path.join(GOLD, "traj_werewolf_n8.json"))
    dsl = json.load(open(dsl_path))
    agent_side = initialize_player_states_from_dsl(dsl, [{"name": "Ann"}, {"name": "Bob"}, {"name": "Cy"}])
    ps = r["init"]["player_states"]
    assert sorted(ps) == ["1", "2", "3"]                          # gamePlayerId, else 1-based position
    for pid in ps:
        for field, value in agent_side[pid].items():
            if field != "name":
                assert ps[pid][field] == value, (pid, field)
    assert (ps["1"]["name"], ps["1"]["id"], ps["1"]["isHost"]) == ("Ann", "u-7", True)
    assert (ps["2"]["name"], ps["2"]["id"], ps["2"]["isHost"]) == ("Bob", "2", False)
    # no template: defaults from the declared schema (boolean -> example or true, num -> example or 0 ...)
    schema = dsl["declaration"]["player_states"]
    gen = r["initFromSchema"]["player_states"]["1"]
    for field, d in schema.items():
        assert field in gen
        if d.get("type") == "boolean":
            assert gen[field] == (d["example"] if "example" in d else True)
    assert r["initFallback"] == {"player_states": {}, "fallback_mode": True,
                                 "message": "No template found, agent will generate player_states"}
    assert r["found"] == ["Werewolf-(Mafia).yaml", None]
    # both hosts derive the same global room index (= RNG key) from a thread id
    from game_engine_amd.room_service import room_index_of
    assert int(r["roomIndex"]) == room_index_of("thread-42 ✓")


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("dsl,gold", [("werewolf-(mafia).json", "traj_werewolf_n8.json"),
                                      ("two-truths-and-a-lie.json", "traj_two_truths_and_a_lie_n4.json"),
                                      ("draft-werewolf-(mafia).json", "traj_draft_werewolf_n8.json")])
def test_node_host_matches_golden(dsl, gold):
    r = _run(os.path.join(GOLD, "dsl", dsl), os.path.join(GOLD, gold))
    g = json.load(open(os.path.join(GOLD, gold)))
    assert r["devices"] >= 1 and r["checked"] == sum(len(c["turns"]) for c in g["cases"])
    assert r["turn"] == 65 and r["queued"] == 65 and r["finished"] > 0 and isinstance(r["sample"], str)
    # handle safety, checkpoint restore with setTurn, close(), batched injection (ge_addon.cc / index.js)
    assert r["busy"] == "GE_BUSY" and r["restoreEqual"] is True and r["closed"] == "refused"
    assert r["injectBatch"]["same"] and r["injectBatch"]["equal"] and r["injectBatch"]["applied"] > 30
    # ShardedBatch (one Node process, several devices): shard-count invariance per room and in the summary
    assert r["shardRooms"] == 9000 and r["shardSummaryEqual"] and r["shardRoomsEqual"]
    # DeviceGroup: the native ge_group_* path (RCCL all-gather of the summaries inside libge_step.so) from Node
    assert r["groupSummaryEqual"] and r["groupRoomsEqual"] and r["groupDuplicate"] == "GE-1"
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
    # overlapping /continue + /action on one thread are serialised; /close frees the room
    bt = r["burst"]
    assert bt["hist"] == 9 and bt["idsOk"] and not bt["busy"]
    assert bt["closed"] is True and "unknown thread" in bt["afterClose"] and bt["rooms"] == 1
    # POST /message: control -> a turn, chat -> none, a game message -> logged under phase 0's name, then a turn
    m = r["message"]
    assert m["kinds"] == ["control", "chat", "action"] and m["played"] == [True, False, True]
    assert m["hist"] == [10, 10, 11] and m["loggedPhase"] == "Game Introduction"


@needs_node
def test_js_group_room_lookup_follows_the_native_partition():
    """DeviceGroup.readRoom finds a room's shard and local index with locateInShards: the same parts ge_group_partition makes."""
    import random
    from game_engine_amd import _lib
    from game_engine_amd.stepper import partition
    rnd = random.Random(3)
    cases = []
    for rooms in ([16777216], [8388608, 8388608], [1000003, 64, 777, 4099], [65, 66, 67]):
        d = _lib.BatchDesc()
        d.n_segments = len(rooms)
        for k, r in enumerate(rooms):
            d.seg[k].n_rooms, d.seg[k].n_players = r, 8
        for n in (1, 2, 3, 8, 64):
            parts = [partition(d, n, i) for i in range(n)]                     # first[] relative to first_room 0
            seg_base = [sum(rooms[:k]) for k in range(len(rooms))]
            probes = {0, sum(rooms) - 1} | {rnd.randrange(sum(rooms)) for _ in range(40)}
            for i, (sd, first) in enumerate(parts):                            # every part's edges too
                for k in range(len(rooms)):
                    probes |= {first[k], first[k] + int(sd.seg[k].n_rooms) - 1}
            for room in sorted(probes):
                k = max(j for j in range(len(rooms)) if seg_base[j] <= room)
                i = next(i for i, (sd, first) in enumerate(parts) if first[k] <= room < first[k] + int(sd.seg[k].n_rooms))
                local = room - parts[i][1][k] + sum(int(parts[i][0].seg[j].n_rooms) for j in range(k))
                cases.append({"rooms": rooms, "n": n, "room": room, "want": [i, local, k]})
    js = ("const {locateInShards}=require(process.argv[1]);const cs=JSON.parse(require('fs').readFileSync(0,'utf8'));"
          "const bad=cs.filter((c)=>JSON.stringify(locateInShards(c.rooms,c.n,c.room))!==JSON.stringify(c.want));"
          "console.log(JSON.stringify({n:cs.length,bad:bad.slice(0,3)}));")
    out = subprocess.run(["node", "-e", js, os.path.join(NODE_DIR, "index.js")], input=json.dumps(cases), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout)
    assert r["n"] == len(cases) > 1000 and r["bad"] == [], r["bad"]
