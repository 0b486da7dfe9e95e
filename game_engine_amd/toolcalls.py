"""A stepped turn, expressed as the reference's backend tool calls.

The reference applies a turn to a room as a sequence of tool calls produced by its LLM nodes
(agent/tools/backend_tools.py:10-157):
    update_player_actions(player_id, actions, phase)      BotBehaviorNode   (v2:589-605)
    set_next_phase(transition, next_phase_id, reason)     PhaseNode         (v2:1128-1142)
    update_player_state(player_id, state_name, value)     RefereeNode       (v2:762-779)
    add_game_note(note_type, content)                     RefereeNode       (v2:780-786)
The GPU stepper produces packed state, not calls.  This module renders the calls of one turn of
one room from what the stepper returns — the room view before and after the turn and the turn's
event (GE_FLAG_TRACE) — so that a host can surface them unchanged and so that a room can be
replayed through the reference's own `_execute_*` plumbing (tests/test_toolcalls.py does that).
Applying the returned calls, in order, to the reference's dict state reproduces the GPU state.
"""
from __future__ import annotations

from typing import Any, Dict, List

from .stepper import PACK_WEREWOLF, TT_FIELDS, WW_FIELDS, GameTable

_TEAMS = ["", "villagers", "werewolves"]
ACT_WOLF_TARGET, ACT_DOCTOR_PROTECT, ACT_DETECTIVE, ACT_DAY_VOTE = 1, 2, 3, 4
ACT_TT_STATEMENTS, ACT_TT_LIE, ACT_TT_VOTE = 5, 6, 7


def action_text(act: int, player: int, choice: int) -> str:
    """The logged sentence of a bot action (bot_behavior_system_prompt.txt:98-110 patterns)."""
    if act in (ACT_WOLF_TARGET, ACT_DAY_VOTE):
        return f"voted to eliminate Player {choice}"
    if act == ACT_DOCTOR_PROTECT:
        return f"chose to protect Player {choice}"
    if act == ACT_DETECTIVE:
        return f"investigated Player {choice}"
    if act == ACT_TT_STATEMENTS:
        return "shared three statements: " + ", ".join(f"'Statement {s} of Player {player}'" for s in (1, 2, 3))
    if act == ACT_TT_LIE:
        return f"chose statement {choice} as the lie"
    return f"voted that statement {choice} is the lie"


def _field_values(table: GameTable, view, i: int) -> Dict[str, Any]:
    f = [int(x) for x in view["players"][i]]
    if int(view["pack"]) == PACK_WEREWOLF:
        n = int(view["n_players"])
        det = [int(x) for x in view["det"][:n]]
        return {"role": table.role_name(f[0]), "team": _TEAMS[f[1]], "is_alive": bool(f[2]),
                "role_revealed": bool(f[3]), "can_vote": bool(f[4]), "has_secret_role": bool(f[5]),
                "night_action_eligible": bool(f[6]), "night_action_submitted": bool(f[7]),
                "selected_target_id": f[8],
                "investigated_alignments": ({str(k + 1): _TEAMS[d] for k, d in enumerate(det) if d}
                                            if f[0] == 4 else {})}
    return {"is_speaker": bool(f[0]), "statements_submitted": bool(f[1]), "lie_index": f[2],
            "lie_revealed": bool(f[3]), "can_vote": bool(f[4]), "vote_choice": f[5], "has_voted": bool(f[6]),
            "total_score": f[7], "rounds_as_speaker": f[8]}


def turn_tool_calls(table: GameTable, before, after, event) -> List[Dict[str, Any]]:
    """Tool calls of one turn of one room, in the order the reference's nodes would issue them.

    `before` / `after`: ge_room_view records (RoomBatch.read_rooms) around the turn;
    `event`: the turn's ge_turn_event (RoomBatch.read_events).  If the slot was recycled at the
    start of the turn (`event.restarted`), `before` must be the fresh initial view."""
    calls: List[Dict[str, Any]] = []
    n = int(after["n_players"])
    turn = int(event["turn"])
    p_id, q_id = int(event["from_phase_id"]), int(event["to_phase_id"])
    rows = {r["phase_id"]: r for r in table.rows()}
    act = rows[p_id]["act"]
    p_name = rows[p_id]["name"]

    # BotBehaviorNode: one update_player_actions per bot that acted this turn
    acted = int(event["acted_now"])
    for i in range(n):
        if (acted >> i) & 1:
            c = int(event["choice"][i])
            calls.append({"name": "update_player_actions",
                          "args": {"player_id": str(i + 1), "actions": f"[t={turn}|c={c}] {action_text(act, i + 1, c)}",
                                   "phase": p_name}})

    # PhaseNode
    calls.append({"name": "set_next_phase",
                  "args": {"transition": q_id != p_id, "next_phase_id": q_id,
                           "transition_reason": "phase complete" if q_id != p_id else "waiting"}})

    # RefereeNode: every declared field that changed, player by player
    fields = WW_FIELDS + ["investigated_alignments"] if int(after["pack"]) == PACK_WEREWOLF else TT_FIELDS
    deaths = []
    for i in range(n):
        b, a = _field_values(table, before, i), _field_values(table, after, i)
        for name in fields:
            if b[name] != a[name]:
                calls.append({"name": "update_player_state",
                              "args": {"player_id": str(i + 1), "state_name": name, "state_value": a[name]}})
        if int(after["pack"]) == PACK_WEREWOLF and b["is_alive"] and not a["is_alive"]:
            deaths.append((i + 1, a["role"]))
        if int(after["pack"]) != PACK_WEREWOLF:
            # `statements` is text the packed state does not carry; it follows statements_submitted
            if a["statements_submitted"] and not b["statements_submitted"]:
                calls.append({"name": "update_player_state", "args": {
                    "player_id": str(i + 1), "state_name": "statements",
                    "state_value": {str(s): f"Statement {s} of Player {i + 1}" for s in (1, 2, 3)}}})
            elif b["statements_submitted"] and not a["statements_submitted"]:
                calls.append({"name": "update_player_state",
                              "args": {"player_id": str(i + 1), "state_name": "statements", "state_value": {}}})
    if q_id != p_id:
        calls.append({"name": "add_game_note",
                      "args": {"note_type": "PHASE_STATUS", "content": f"[t={turn}] phase {p_id} -> {q_id}"}})
    for pid, role in deaths:
        calls.append({"name": "add_game_note",
                      "args": {"note_type": "CRITICAL", "content": f"Player {pid} ({role}) eliminated - marked is_alive=false"}})
    return calls
