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

from typing import Any, Dict, List, Optional

from .stepper import PACK_WEREWOLF, SLOT_STATEMENTS, GameTable, slot_values

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


WW_ROLE, WW_IS_ALIVE, WW_SELECTED_TARGET = 0, 2, 8          # slots the notes read (include/ge_step.h GE_WW_*)
TT_IS_SPEAKER, TT_STATEMENTS_SUBMITTED, TT_TOTAL_SCORE = 0, 1, 7


# entry effects (include/ge_step.h GE_EFF_*)
EFF_ASSIGN_ROLES, EFF_NIGHT_BEGIN, EFF_NIGHT_RESOLVE, EFF_DAY_RESOLVE = 1, 2, 3, 4
EFF_TT_ROUND_START, EFF_TT_REVEAL, EFF_TT_SCORE = 5, 6, 7
ROLE_WEREWOLF, ROLE_DOCTOR = 2, 3

# add_game_note's categories and their marks (agent/tools/backend_tools.py:175-187; unknown type -> the EVENT mark)
NOTE_EMOJI = {"CRITICAL": "\U0001F534", "VOTING_STATUS": "\u26A0\uFE0F", "DECISION": "\U0001F3AF", "BOT_REMINDER": "\U0001F916",
              "UI_FILTER": "\U0001F6AB", "PHASE_STATUS": "\u23F3", "NEXT_PHASE": "\U0001F52E", "GAME_STATUS": "\U0001F3C6",
              "PHASE_SUGGESTION": "\U0001F4A1", "BRANCH_RECOMMENDATION": "\U0001F500", "EVENT": "\U0001F4DD"}


def format_note(note_type: str, content: str) -> str:
    """What _execute_add_game_note appends (bt:188-198): '<mark> <TYPE>: <content>', not doubled."""
    prefix = f"{NOTE_EMOJI.get(note_type, NOTE_EMOJI['EVENT'])} {note_type}:"
    return content if content.startswith(prefix) else f"{prefix} {content}"


def _plurality(votes: List[int], n: int) -> int:
    """Most votes, ties -> lowest player id, 0 if nobody voted (POLICY.md §3)."""
    best, best_c = 0, 0
    for k in range(1, n + 1):
        c = sum(1 for v in votes if v == k)
        if c > best_c:
            best, best_c = k, c
    return best


def turn_tool_calls(table: GameTable, before, after, event) -> List[Dict[str, Any]]:
    """Tool calls of one turn of one room, in the order the reference's nodes would issue them.

    `before` / `after`: ge_room_view records (RoomBatch.read_rooms) around the turn;
    `event`: the turn's ge_turn_event (RoomBatch.read_events).  If the slot was recycled at the
    start of the turn (`event.restarted`), `before` must be the fresh initial view.
    The Referee's notes are the fixed policy's (POLICY.md; pinned by tests/golden/strings_*.json):
    a PHASE_STATUS note per transition, then what the entry effect decided."""
    calls: List[Dict[str, Any]] = []
    n = int(after["n_players"])
    turn = int(event["turn"])
    p_id, q_id = int(event["from_phase_id"]), int(event["to_phase_id"])
    rows = {r["phase_id"]: r for r in table.rows()}
    act = rows[p_id]["act"]
    p_name = rows[p_id]["name"]
    ww = int(after["pack"]) == PACK_WEREWOLF

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

    # RefereeNode: every declared field that changed, player by player, under the DSL's own field names
    # (a slot the DSL does not declare is engine state only: no call)
    names = table.field_names
    deaths = []
    bvals = [slot_values(table, before, i) for i in range(n)]
    avals = [slot_values(table, after, i) for i in range(n)]
    for i in range(n):
        b, a = bvals[i], avals[i]
        for s in range(len(a)):
            if names[s] and b[s] != a[s]:
                calls.append({"name": "update_player_state",
                              "args": {"player_id": str(i + 1), "state_name": names[s], "state_value": a[s]}})
        if ww and b[WW_IS_ALIVE] and not a[WW_IS_ALIVE]:
            deaths.append((i + 1, a[WW_ROLE]))
        if not ww and names[SLOT_STATEMENTS]:
            # `statements` is text the packed state does not carry; it follows statements_submitted
            if a[TT_STATEMENTS_SUBMITTED] and not b[TT_STATEMENTS_SUBMITTED]:
                calls.append({"name": "update_player_state", "args": {
                    "player_id": str(i + 1), "state_name": names[SLOT_STATEMENTS],
                    "state_value": {str(s): f"Statement {s} of Player {i + 1}" for s in (1, 2, 3)}}})
            elif b[TT_STATEMENTS_SUBMITTED] and not a[TT_STATEMENTS_SUBMITTED]:
                calls.append({"name": "update_player_state",
                              "args": {"player_id": str(i + 1), "state_name": names[SLOT_STATEMENTS], "state_value": {}}})
    if q_id == p_id:
        return calls

    def note(kind: str, text: str):
        calls.append({"name": "add_game_note", "args": {"note_type": kind, "content": text}})

    note("PHASE_STATUS", f"[t={turn}] phase {p_id} -> {q_id}")
    eff = rows[q_id]["effect"]
    if eff == EFF_ASSIGN_ROLES:
        note("NEXT_PHASE", "Roles assigned: " + ", ".join(f"Player{i + 1}={avals[i][WW_ROLE]}" for i in range(n)))
    elif eff in (EFF_NIGHT_RESOLVE, EFF_DAY_RESOLVE):
        how = "overnight by the werewolves" if eff == EFF_NIGHT_RESOLVE else "by day vote"
        for pid, role in deaths:
            note("CRITICAL", f"Player {pid} ({role}) eliminated {how} - marked is_alive=false")
        if eff == EFF_NIGHT_RESOLVE and not deaths:
            # nobody died: say what was decided (the choices are still in the record's selected-target slot)
            cls = [int(before["players"][i][0]) for i in range(n)]
            alive = [bvals[i][WW_IS_ALIVE] for i in range(n)]
            victim = _plurality([avals[i][WW_SELECTED_TARGET] for i in range(n) if alive[i] and cls[i] == ROLE_WEREWOLF], n)
            protect = 0
            for i in range(n):
                if alive[i] and cls[i] == ROLE_DOCTOR:
                    protect = avals[i][WW_SELECTED_TARGET]
            note("DECISION", f"Werewolves targeted Player {victim}, Doctor protected Player {protect} - no elimination")
    elif eff == EFF_TT_ROUND_START:
        speaker = next((i + 1 for i in range(n) if avals[i][TT_IS_SPEAKER]), 0)
        note("DECISION", f"Selected Player {speaker} as next speaker (turn_order)")
    elif eff == EFF_TT_SCORE:
        if any(bvals[i][TT_IS_SPEAKER] for i in range(n)):
            note("SCORE_UPDATE", "Total scores - " + ", ".join(f"Player {i + 1}: {avals[i][TT_TOTAL_SCORE]}" for i in range(n)))
    return calls


class RoomLog:
    """The log-shaped parts of one room's AgentState that the packed state does not carry -
    playerActions (bt:285-344), game_notes (bt:163-202), phase_history (v2:1207-1215), the Two-Truths
    `statements` texts and the players' names - kept by folding each turn's tool calls exactly as the
    reference's `_execute_*` functions would.  Shared by RoomService and the string-layer tests."""

    def __init__(self, table: GameTable, names: List[str], game_name: str = ""):
        self.table, self.names, self.game_name = table, list(names), game_name
        self.player_actions: Dict[str, Dict[str, Any]] = {}
        self.game_notes: List[str] = []
        self.phase_history: List[Dict[str, Any]] = []
        self.statements: Dict[str, Dict[str, str]] = {}

    def fold(self, calls: List[Dict[str, Any]], after, now_ms: Optional[int] = None) -> None:
        """Apply one turn's calls; `after`: the room view after the turn."""
        import time
        for c in calls:
            a = c["args"]
            if c["name"] == "update_player_actions":
                pid = a["player_id"]
                rec = self.player_actions.setdefault(pid, {"name": self.names[int(pid) - 1], "actions": {}})
                aid = str(max((int(x["id"]) for x in rec["actions"].values()), default=0) + 1)     # per-player sequence, bt:323-332
                rec["name"] = self.names[int(pid) - 1]
                rec["actions"][aid] = {"action": a["actions"], "timestamp": int(time.time() * 1000) if now_ms is None else now_ms,
                                       "phase": a["phase"], "id": aid}
            elif c["name"] == "add_game_note":
                self.game_notes.append(format_note(a["note_type"], a["content"]))
            elif c["name"] == "update_player_state" and a["state_name"] == self.table.field_names[SLOT_STATEMENTS] and \
                    self.table.pack != PACK_WEREWOLF:
                self.statements[a["player_id"]] = dict(a["state_value"])
        pid = int(after["phase_id"])
        entry = {"phase_id": pid, "phase_name": self.table.phase_name(pid)}
        if now_ms is None:
            import datetime
            entry["timestamp"] = datetime.datetime.now().isoformat()
        self.phase_history.append(entry)                      # one entry per turn, transition or not (v2:1207-1215)

    def person_message(self, text: str, now_ms: Optional[int] = None) -> None:
        """File a person's game message as process_human_action_if_needed does (agent/tools/utils.py:343-350): under Player 1
        whoever sent it, the first 200 characters, and under PHASE 0's NAME whatever the current phase is - InitialRouterNode
        passes `state.get("currentPhaseId", 0)` and `state.get("playerStates", {})`, keys the state does not have
        (agent/game_agent_v2.py:324-331), so the phase id is always 0 and the name comes from roomSession.  Mirrored, not
        corrected: the string layer is the reference's (POLICY.md 3b)."""
        import time
        from .messages import logged_text
        rec = self.player_actions.setdefault("1", {"name": self.names[0], "actions": {}})
        aid = str(max((int(x["id"]) for x in rec["actions"].values()), default=0) + 1)       # Player 1's own sequence, bt:323-332
        rec["name"] = self.names[0]
        rec["actions"][aid] = {"action": logged_text(text), "timestamp": int(time.time() * 1000) if now_ms is None else now_ms,
                               "phase": self.table.phase_name(0), "id": aid}

    def agent_state(self, view) -> Dict[str, Any]:
        """AgentState of the room (v2:97-117) with the reference's key order inside player_states."""
        from .stepper import view_to_agent_state
        s = view_to_agent_state(self.table, view)
        tt = int(view["pack"]) != PACK_WEREWOLF
        for i, pid in enumerate(sorted(s["player_states"], key=int)):
            rec = s["player_states"][pid]
            out = {"name": self.names[i]}
            for k, v in rec.items():
                out[k] = v
                if tt and k == self.table.field_names[TT_IS_SPEAKER] and self.table.field_names[SLOT_STATEMENTS]:
                    out[self.table.field_names[SLOT_STATEMENTS]] = dict(self.statements.get(pid, {}))
            s["player_states"][pid] = out
        s.update(gameName=self.game_name, playerActions=self.player_actions, phase_history=self.phase_history,
                 game_notes=self.game_notes)
        return s
