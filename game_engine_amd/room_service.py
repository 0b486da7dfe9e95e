"""RoomService — the single-room drop-in for Python hosts (twin of node/room_service.js).

In the reference one LangGraph thread is one room, and every "Continue" message is one run of the
graph (agent/game_agent_v2.py:1571-1587) that returns the AgentState (v2:97-117) plus an AIMessage
whose tool calls drive the frontend.  `continue_room(thread_id)` is that run without the LLM: an N=1
traced batch advances the room by one turn; the turn comes back as the reference's backend tool
calls (toolcalls.turn_tool_calls, agent/tools/backend_tools.py:10-157) and the phase now showing as
frontend tool calls (ui_script.ui_tool_calls); the log-shaped parts of AgentState the packed state
does not carry — playerActions (bt:285-344), game_notes (bt:163-202), phase_history (v2:1207-1215) —
are folded from those calls exactly as the reference's `_execute_*` functions would.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

from . import messages as M
from .stepper import GE_ERR_ARG, PACK_WEREWOLF, GeError, GameTable, RoomBatch, load_dsl_by_gamename, slot_values, view_to_agent_state
from .toolcalls import WW_IS_ALIVE, RoomLog, turn_tool_calls
from .ui_script import ui_tool_calls


def room_index_of(thread_id: str) -> int:
    """Stable 48-bit global room index of a thread id (FNV-1a 64, low 48 bits; same as the JS host):
    the RNG is keyed by it, so a thread replays identically on any host."""
    h = 0xCBF29CE484222325
    for ch in str(thread_id).encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h & 0xFFFFFFFFFFFF


class RoomService:
    def __init__(self, games_dir: str = "games", seed: int = 0, device: int = 0):
        self.games_dir, self.seed, self.device = games_dir, seed, device
        self._tables: Dict[str, GameTable] = {}
        self._rooms: Dict[str, Dict[str, Any]] = {}

    def table(self, game_name: str, dsl: Optional[dict] = None) -> GameTable:
        if game_name not in self._tables:
            self._tables[game_name] = GameTable(dsl if dsl else load_dsl_by_gamename(game_name, self.games_dir))
        return self._tables[game_name]

    def create_room(self, thread_id: str, game_name: str, players: List[Dict[str, Any]], dsl: Optional[dict] = None,
                    room_index: Optional[int] = None) -> Dict[str, Any]:
        """players: roomSession.players as the lobby builds it; `isBot: False` marks a human seat, which
        the bot policy never acts for (bot_behavior_system_prompt.txt:3) — use human_action for it.
        room_index: the global room index the RNG is keyed by (default: derived from the thread id)."""
        tb = self.table(game_name, dsl)
        human_mask = sum(1 << i for i, p in enumerate(players) if p.get("isBot") is False)
        batch = self._new_batch(tb, len(players), human_mask, room_index_of(thread_id) if room_index is None else room_index)
        if thread_id in self._rooms:
            self.close(thread_id)
        names = [p.get("name") or f"Player {i + 1}" for i, p in enumerate(players)]
        room = {"batch": batch, "table": tb, "gameName": game_name, "names": names, "panel": None,
                "human_seats": [i + 1 for i in range(len(players)) if (human_mask >> i) & 1],
                "view": batch.read_rooms(0, 1)[0], "log": RoomLog(tb, names, game_name)}
        self._rooms[thread_id] = room
        return self._agent_state(room)

    def _new_batch(self, tb: GameTable, n_players: int, human_mask: int, first_room: int) -> RoomBatch:
        """The room's N=1 traced batch on the device (there is no other stepper: without the HIP library this raises)."""
        return RoomBatch([(tb, n_players, 1, human_mask)], seed=self.seed, first_room=first_room,
                         device=self.device, max_fuse=1, trace=True)

    def _agent_state(self, room: Dict[str, Any]) -> Dict[str, Any]:
        return room["log"].agent_state(room["view"])

    def human_action(self, thread_id: str, player_id: int, choice: int) -> Dict[str, Any]:
        """A human's vote / choice (logged by process_human_action_if_needed, agent/tools/utils.py:310-358)."""
        room = self._rooms[thread_id]
        room["batch"].inject_action(0, player_id, choice)
        room["view"] = room["batch"].read_rooms(0, 1)[0]
        return self._agent_state(room)

    def continue_room(self, thread_id: str, items: Optional[List[Dict[str, Any]]] = None) -> Dict[str, Any]:
        """One turn (one graph run): {"state": AgentState, "toolCalls": [...], "uiCalls": [...]}.
        items: the frontend's canvas items (AgentState.items, [{id, type, ...}]) when the caller has them:
        clearCanvas then names the ids to keep (exemptList)."""
        return self._turn(self._rooms[thread_id], items)

    def handle_message(self, thread_id: str, text: str, items: Optional[List[Dict[str, Any]]] = None) -> Dict[str, Any]:
        """The drop-in's message-level entry: what the reference's graph does with ONE message of the browser
        (src/app/page.tsx:183-259 -> agent/game_agent_v2.py:198-349, agent/tools/utils.py:310-358; POLICY.md 3b).
          chat ("... in game chat: ..." / "... to Bot k: ...")  -> ChatBotNode: no turn, no state change;
          control ("Start game.", "Continue")                    -> one turn;
          anything else -> logged verbatim under Player 1 (first 200 characters, phase 0's name - the reference's own quirk),
                           read as a seat's action where it is one (a vote on the newest panel, an input for the
                           statements phase: ge_batch_inject_action), then one turn.
        Returns {"state", "toolCalls", "uiCalls", "played", "kind"}; an action message that is no valid game action is still
        logged and still plays the turn, as in the reference."""
        room = self._rooms[thread_id]
        kind = M.classify(text)
        if kind == M.CHAT:
            return {"state": self._agent_state(room), "toolCalls": [], "uiCalls": [], "played": False, "kind": kind}
        if kind == M.ACTION:
            room["log"].person_message(text)
            view, tb = room["view"], room["table"]
            n = int(view["n_players"])
            pid = int(view["phase_id"])
            act = next((r["act"] for r in tb.rows() if r["phase_id"] == pid), 0)
            alive = [bool(slot_values(tb, view, i)[WW_IS_ALIVE]) for i in range(n)] if tb.pack == PACK_WEREWOLF else [True] * n
            for seat, choice in M.resolve(text, room["panel"], act, tb.pack, room["names"], alive, room["human_seats"]):
                try:
                    room["batch"].inject_action(0, seat, choice)
                    break
                except GeError as e:                     # not a living pending target of this phase: logged, no game effect
                    if e.status != GE_ERR_ARG:
                        raise
        out = self._turn(room, items)
        out.update(played=True, kind=kind)
        return out

    def _turn(self, room: Dict[str, Any], items: Optional[List[Dict[str, Any]]] = None) -> Dict[str, Any]:
        # `before` is the view BEFORE any injected action of this message: the person's record writes (night target, vote
        # choice, ...) then show up among the turn's update_player_state calls, where the reference's Referee issues them
        batch, before = room["batch"], room["view"]
        batch.step(1)
        after = batch.read_rooms(0, 1)[0]
        event = batch.read_events(0, 1)[0][0]
        calls = turn_tool_calls(room["table"], before, after, event)
        room["log"].fold(calls, after)                      # playerActions / game_notes / phase_history, as bt:163-202, 285-344 would
        room["view"] = after
        state = self._agent_state(room)
        deaths = [c["args"]["player_id"] for c in calls if c["name"] == "update_player_state"
                  and c["args"]["state_name"] == "is_alive" and c["args"]["state_value"] is False]
        ui = ui_tool_calls(room["table"].dsl, state, room["table"], turn=int(event["turn"]), deaths=deaths, items=items)
        room["panel"] = M.newest_panel(ui)                # what a person's next vote message can answer
        return {"state": state, "toolCalls": calls, "uiCalls": ui}

    def close(self, thread_id: Optional[str] = None):
        for tid in ([thread_id] if thread_id else list(self._rooms)):
            self._rooms.pop(tid)["batch"].close()
