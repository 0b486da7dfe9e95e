"""A GPU-stepped turn must be expressible as the reference's backend tool calls (SURVEY §8b,
operator surface): game_engine_amd.toolcalls renders the calls of a turn from (view before, view
after, turn event); applying them with the REFERENCE'S OWN plumbing
(agent/tools/backend_tools.py `_execute_*`) must reproduce the golden projection turn by turn.

CPU part (needs /root/reference, build container): views and events come from the oracle.
GPU part (-m gpu): the kernels' event trace equals the oracle's."""
import asyncio
import os
import sys

import numpy as np
import pytest

from conftest import golden_dsl, load_dsl, load_golden, restart_files
from game_engine_amd import GameTable
from game_engine_amd.toolcalls import turn_tool_calls
from parity_util import oracle_events, oracle_rooms_as_views

REF = "/root/reference/agent/tools/backend_tools.py"


def _reference_plumbing():
    from oracle.refharness import walker
    for p in (walker._STANDINS, os.path.join(walker.REFERENCE_ROOT, "agent")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True
    import tools.backend_tools as bt
    import tools.utils as ut
    return bt, ut


def _replay(name, restart=False):
    from oracle.oracle import Oracle
    from oracle.refharness.walker import project_state
    bt, ut = _reference_plumbing()
    g = load_golden(name)
    dsl = golden_dsl(g)
    # the reference loads YAML with int phase keys; restore them for its helpers
    dsl_ref = dict(dsl, phases={int(k): v for k, v in dsl["phases"].items()})
    n = g["n_players"]
    orc = Oracle(dsl, n, rounds=g["rounds"])
    table = GameTable(dsl, rounds=g["rounds"])
    terminal = {p.id for p in orc.table.phases if not p.branches}
    room_session = {"players": [{"name": f"Bot {i + 1}", "gamePlayerId": i + 1} for i in range(n)]}

    def fresh_state():
        ps = asyncio.run(ut.initialize_player_states_from_dsl(dsl_ref, room_session["players"]))
        return {"current_phase_id": 0, "player_states": ps, "playerActions": {}, "phase_history": [], "game_notes": []}

    for case in g["cases"]:
        rooms = orc.init_rooms(1)
        state, t_enter, prev, end_turn = fresh_state(), -1, 0, -1
        for t, want in enumerate(case["turns"]):
            before = oracle_rooms_as_views(orc, rooms)[0].copy()
            orc.run(rooms, case["seed"], case["room"], t, 1, restart=restart)
            after = oracle_rooms_as_views(orc, rooms)[0]
            ev = oracle_events(orc, rooms, t)[0]
            if ev["restarted"]:                      # a new LangGraph thread on the recycled slot
                before = oracle_rooms_as_views(orc, orc.init_rooms(1))[0]
                state, t_enter, prev, end_turn = fresh_state(), t - 1, 0, -1
            p0 = state["current_phase_id"]
            for call in turn_tool_calls(table, before, after, ev):
                a = call["args"]
                if call["name"] == "update_player_actions":
                    state["playerActions"] = bt._execute_update_player_actions(
                        state["playerActions"], a["player_id"], a["actions"], a["phase"], room_session, state["player_states"])
                elif call["name"] == "update_player_state":
                    state["player_states"] = bt._execute_update_player_state(
                        state["player_states"], a["player_id"], a["state_name"], a["state_value"])
                elif call["name"] == "add_game_note":
                    state["game_notes"] = bt._execute_add_game_note(state["game_notes"], a["note_type"], a["content"])
                elif call["name"] == "set_next_phase":
                    assert a["next_phase_id"] in dsl_ref["phases"]          # v2:1173-1204 validation
                    if a["transition"]:
                        state["current_phase_id"] = a["next_phase_id"]
                    state["phase_history"].append({"phase_id": state["current_phase_id"]})   # v2:1207-1215
            q = state["current_phase_id"]
            if q != p0:
                t_enter, prev = t, p0
                if q in terminal and end_turn < 0:
                    end_turn = t
            got = project_state(orc.table, state, t_enter, prev, end_turn)
            assert got == want, f"{name} seed={case['seed']:#x} room={case['room']} turn={t}"


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("name", ["traj_werewolf_n8.json", "traj_werewolf_n12.json", "traj_werewolf_n5.json",
                                  "traj_two_truths_and_a_lie_n4.json", "traj_two_truths_and_a_lie_n6.json",
                                  "traj_draft_werewolf_n8.json", "traj_draft_werewolf_n12.json",
                                  "traj_variant_ww_minimal_schema_n8.json", "traj_variant_ww_generic_n8.json"])
def test_rendered_calls_replay_through_reference_plumbing(name):
    _replay(name)


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("name", restart_files())
def test_rendered_calls_replay_with_recycled_rooms(name):
    _replay(name, restart=True)


def test_call_shapes(dsl_ww):
    """Argument names and types are the reference's (bt:10-24, 26-40, 144-157)."""
    from oracle.oracle import Oracle
    orc = Oracle(dsl_ww, 8)
    table = GameTable(dsl_ww)
    rooms = orc.init_rooms(1)
    seen = set()
    for t in range(40):
        before = oracle_rooms_as_views(orc, rooms)[0].copy()
        orc.run(rooms, 3, 0, t, 1)
        for c in turn_tool_calls(table, before, oracle_rooms_as_views(orc, rooms)[0], oracle_events(orc, rooms, t)[0]):
            seen.add(c["name"])
            a = c["args"]
            if c["name"] == "set_next_phase":
                assert isinstance(a["transition"], bool) and isinstance(a["next_phase_id"], int) and isinstance(a["transition_reason"], str)
            elif c["name"] == "update_player_state":
                assert isinstance(a["player_id"], str) and isinstance(a["state_name"], str)
            elif c["name"] == "update_player_actions":
                assert isinstance(a["player_id"], str) and isinstance(a["actions"], str) and isinstance(a["phase"], str)
    assert seen == {"set_next_phase", "update_player_state", "update_player_actions", "add_game_note"}


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,restart", [("werewolf-(mafia)", 8, True), ("werewolf-(mafia)", 12, False),
                                            ("two-truths-and-a-lie", 4, True), ("two-truths-and-a-lie", 9, False),
                                            ("draft-werewolf-(mafia)", 8, True)])
def test_kernel_event_trace_equals_oracle(game, n, restart):
    from game_engine_amd import RoomBatch
    from oracle.oracle import Oracle
    dsl = load_dsl(game)
    R, seed, first = 3000, 11, 1 << 20
    orc = Oracle(dsl, n)
    rooms = orc.init_rooms(R)
    with RoomBatch([(GameTable(dsl), n, R)], seed=seed, first_room=first, max_fuse=16, restart=restart, trace=True) as b:
        t = 0
        for chunk in (1, 16, 7, 16, 16, 3, 16, 16, 16):
            b.step(chunk)
            ev = b.read_events()
            assert ev.shape == (R, chunk)
            for k in range(chunk):
                orc.run(rooms, seed, first, t, 1, threads=0, restart=restart)
                want = oracle_events(orc, rooms, t)
                got = np.ascontiguousarray(ev[:, k])
                for f in ("turn", "from_phase_id", "to_phase_id", "acted_now", "restarted", "choice"):
                    assert (got[f] == want[f]).all(), (f, t)
                t += 1
        assert b.read_rooms().tobytes() == oracle_rooms_as_views(orc, rooms).tobytes()
        with pytest.raises(Exception):
            b.step(17)                               # a traced step may not exceed max_fuse turns


@pytest.mark.gpu
@pytest.mark.parametrize("game,n", [("werewolf-(mafia)", 8), ("two-truths-and-a-lie", 4)])
def test_python_room_service_single_thread(game, n):
    """RoomService (Python twin of node/room_service.js): one thread = one N=1 traced batch.  The state it
    returns is the stepped room's; the log-shaped AgentState parts are the fold of the turn's tool calls;
    a human seat only moves through human_action; the same thread id replays identically."""
    from game_engine_amd import GameTable, RoomBatch, RoomService, room_index_of
    from game_engine_amd.stepper import view_to_agent_state
    dsl = load_dsl(game)
    players = [{"name": f"P{i + 1}", "isBot": i != 0} for i in range(n)]          # seat 1 is a person
    pick = 2 if game.startswith("werewolf") else 1             # "Player 2" / "statement 1": what the person always answers
    svc = RoomService(seed=9)
    st = svc.create_room("thread-A", game, players, dsl=dsl)
    assert st["current_phase_id"] == 0 and sorted(st["player_states"], key=int) == [str(i + 1) for i in range(n)]
    assert st["player_states"]["1"]["name"] == "P1"
    tb = GameTable(dsl)
    n_actions = n_notes = 0
    with RoomBatch([(tb, n, 1, 1)], seed=9, first_room=room_index_of("thread-A"), max_fuse=1) as ref:
        from game_engine_amd import GeError
        accepted = 0
        for t in range(60):
            # the person in seat 1 tries to act before every turn; refusals (not a target, already acted,
            # dead target) must agree between the service and a bare batch
            ok_svc = ok_ref = True
            try:
                svc.human_action("thread-A", 1, pick)
            except GeError:
                ok_svc = False
            try:
                ref.inject_action(0, 1, pick)
            except GeError:
                ok_ref = False
            assert ok_svc == ok_ref, t
            accepted += ok_svc
            out = svc.continue_room("thread-A")
            ref.step(1)
            want = view_to_agent_state(tb, ref.read_rooms(0, 1)[0])
            got = out["state"]
            assert got["current_phase_id"] == want["current_phase_id"], t
            for pid, rec in want["player_states"].items():
                assert {k: v for k, v in got["player_states"][pid].items() if k not in ("name", "statements")} == rec, (t, pid)
            n_actions += sum(c["name"] == "update_player_actions" for c in out["toolCalls"])
            n_notes += sum(c["name"] == "add_game_note" for c in out["toolCalls"])
            assert all(c["args"]["player_id"] != "1" for c in out["toolCalls"] if c["name"] == "update_player_actions")
            assert isinstance(out["uiCalls"], list)
        assert len(got["phase_history"]) == 60 and len(got["game_notes"]) == n_notes
        assert sum(len(r["actions"]) for r in got["playerActions"].values()) == n_actions > 0
        assert accepted > 0 or game.startswith("werewolf")      # (seat 1 dies in the first night of this thread)
    # a second service replays the same thread bit for bit
    svc2 = RoomService(seed=9)
    svc2.create_room("thread-A", game, players, dsl=dsl)
    for _ in range(60):
        try:
            svc2.human_action("thread-A", 1, pick)
        except GeError:
            pass
        last = svc2.continue_room("thread-A")
    assert last["state"]["player_states"] == got["player_states"] and last["state"]["game_notes"] == got["game_notes"]
    svc.close(); svc2.close()
