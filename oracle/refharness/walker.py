"""ORACLE (test infrastructure) — drives the reference's own node coroutines.

Imports /root/reference/agent/game_agent_v2.py (primary; the deployed graph,
agent/langgraph.json:6) or game_agent_v3.py (secondary) with the stand-in
third-party modules of ./standins, replaces the LangGraph runtime by a tiny walker
(follow Command.goto until END, merge Command.update) and every LLM by
policy.FixedPolicy.  Build container only: /root/reference does not exist on the
GPU box, and nothing under tests -m gpu / bench.py imports this file.
"""
from __future__ import annotations

import asyncio
import importlib
import logging
import os
import sys
from typing import Any, Dict, List, Optional

REFERENCE_ROOT = os.environ.get("GE_REFERENCE_ROOT", "/root/reference")
_STANDINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standins")
_loaded: Dict[str, Any] = {}


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "agent", "game_agent_v2.py"))


def load_reference(version: str = "v2"):
    """Import the reference agent module once, neutralising its import side effects
    (v2:76-83 creates /home/lee/game_engine/logs and a FileHandler at import)."""
    if version in _loaded:
        return _loaded[version]
    if not reference_available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True           # the reference mount is read-only
    agent_dir = os.path.join(REFERENCE_ROOT, "agent")
    for p in (_STANDINS, agent_dir):
        if p not in sys.path:
            sys.path.insert(0, p)
    real_makedirs, real_fh = os.makedirs, logging.FileHandler

    def _makedirs(path, *a, **kw):
        if str(path).startswith("/home/lee"):
            return None
        return real_makedirs(path, *a, **kw)

    os.makedirs = _makedirs
    logging.FileHandler = lambda *a, **kw: logging.NullHandler()
    real_sh = logging.StreamHandler
    logging.StreamHandler = lambda *a, **kw: logging.NullHandler()
    try:
        mod = importlib.import_module(f"game_agent_{version}")
    finally:
        os.makedirs, logging.FileHandler, logging.StreamHandler = real_makedirs, real_fh, real_sh
    mod.logger.handlers.clear()
    mod.logger.setLevel(logging.CRITICAL)
    logging.getLogger("tools.utils").setLevel(logging.CRITICAL)
    _loaded[version] = mod
    return mod


class _StubModel:
    """What `init_chat_model(...)` returns: `.bind_tools()` then `.ainvoke()`."""

    def __init__(self, session: "RoomSession", tool_names=()):
        self.session = session
        self.tool_names = frozenset(tool_names)

    def bind_tools(self, tools, **_kw):
        names = []
        for t in tools:
            n = getattr(t, "name", None)
            if n is None and isinstance(t, dict):
                n = (t.get("function") or {}).get("name") or t.get("name")
            names.append(n)
        return _StubModel(self.session, names)

    async def ainvoke(self, messages, config=None):
        from langchain_core.messages import AIMessage
        self.session.llm_calls += 1
        calls = self.session.answer(self.tool_names)
        return AIMessage(content="", tool_calls=[
            {"name": c["name"], "args": c["args"], "id": f"call_{self.session.llm_calls}_{k}",
             "type": "tool_call"} for k, c in enumerate(calls)])


class RoomSession:
    """One room (= one LangGraph thread, src/app/api/copilotkit/route.ts:24-37)."""

    def __init__(self, game: str, n_players: int, seed: int, room: int = 0,
                 version: str = "v2", rounds: int = 1, turn0: int = 0, human_mask: int = 0, human_script=None, game_index: int = 0,
                 dsl_variant=None, names=None):
        """human_mask: host-driven seats (roomSession isBot: false).  human_script(session) -> the message a person sends to
        start the next graph run (None: the "Continue" button), see oracle/human_script.py.  names: roomSession player names."""
        from .. import dsl_table
        from .policy import FixedPolicy
        import yaml
        self.version = version
        self.mod = load_reference(version)
        draft = game.startswith("draft-")
        path = os.path.join(REFERENCE_ROOT, "game_draft", f"{game[6:]}.yaml") if draft else \
            os.path.join(REFERENCE_ROOT, "games", f"{game}.yaml")
        with open(path, encoding="utf-8") as f:
            dsl = yaml.safe_load(f)
        if draft and dsl_variant is None:
            # a DSL from game_draft/ is not where load_dsl_by_gamename looks (utils.py:557-581): it is handed over
            # in the state, as a freshly generated DSL is (v2:254-262 keeps a DSL that is already there)
            dsl_variant = lambda d: d
        if dsl_variant is not None:
            # a variant of the game's DSL (oracle/dsl_variants.py): the reference's InitialRouterNode keeps a DSL
            # that is already in the state instead of loading the game's file (v2:254-262)
            dsl = dsl_variant(dsl)
        self.table = dsl_table.compile_dsl(dsl, rounds=rounds)
        self.human_script, self.human_mask, self.n_players = human_script, human_mask, n_players
        self.last_panel = None         # (votingId, options) of the newest createVotingPanel call, what a person answers
        self.last_message = None       # what started the latest graph run
        self.policy = FixedPolicy(self.table, seed, room, human_mask)
        self.policy.game = min(game_index, 0xFFFF)
        self.turn = turn0              # the clock may start late: a new room on a recycled slot
        self.t_enter, self.prev_phase, self.end_turn = turn0 - 1, 0, -1
        self.llm_calls = 0
        self.node_path: List[str] = []
        self.state: Dict[str, Any] = {
            "messages": [], "gameName": game, "items": [], "tools": [],
            "current_phase_id": 0, "player_states": {}, "playerActions": {},
            "phase_history": [], "game_notes": [], "dsl": (dsl if dsl_variant is not None else {}),
            "roomSession": {"players": [
                {"name": (names[i] if names else f"Bot {i + 1}"), "gamePlayerId": i + 1, "isBot": not (human_mask >> i) & 1}
                for i in range(n_players)]},
        }
        self._p_at_turn_start = 0

    # ---- the stub LLM's dispatch: which node is asking is told by the bound tools
    def answer(self, tool_names) -> List[dict]:
        st, pol = self.state_in_node, self.policy
        self._stamp_person_entries()
        cur = st.get("current_phase_id", 0)
        if tool_names == {"update_player_actions"}:
            return pol.bot_actions(st, cur)
        if tool_names == {"set_next_phase"}:                       # v2 PhaseNode
            tr, q, why = pol.phase_decision(st, cur)
            return [{"name": "set_next_phase",
                     "args": {"transition": tr, "next_phase_id": q, "transition_reason": why}}]
        if tool_names == {"update_player_state", "add_game_note"}:  # v2 RefereeNode
            return pol.referee_updates(st, self._p_at_turn_start, cur)
        if "set_next_phase" in tool_names:                          # v3 ActionExecutor (fused)
            tr, q, why = pol.phase_decision(st, cur)
            calls = pol.referee_updates(st, cur, q if tr else cur)
            calls.append({"name": "set_next_phase",
                          "args": {"transition": tr, "next_phase_id": q if tr else cur,
                                   "transition_reason": why}})
            return calls
        return pol.ui_calls(st, cur)                                # v2 ActionExecutor / v3 UIUpdateNode

    async def _run_graph(self):
        from langgraph.graph import END
        import langchain.chat_models as cm
        cm.set_model_factory(lambda _name: _StubModel(self))
        node = "InitialRouterNode"
        while node != END:
            self.node_path.append(node)
            self.state_in_node = self.state
            cmd = await getattr(self.mod, node)(self.state, {})
            for k, v in (cmd.update or {}).items():
                if k == "messages":
                    if isinstance(v, list):
                        self.state["messages"].extend(v)
                    elif v is not None:
                        self.state["messages"].append(v)
                else:
                    self.state[k] = v
            node = cmd.goto

    def _stamp_person_entries(self):
        """The runtime's clock for what process_human_action_if_needed filed (utils.py:343-350: Player 1's log, wall-clock
        stamp only): the turn of the graph run in which the entry appeared."""
        rec = (self.state.get("playerActions") or {}).get("1") or {}
        for aid, a in (rec.get("actions") or {}).items():
            if str(aid) not in self.policy.human_turns and not str(a.get("action", "")).startswith("[t="):
                self.policy.human_turns[str(aid)] = self.turn

    @staticmethod
    def is_chat(message: str) -> bool:
        """InitialRouterNode's own test (v2:305-311, case-sensitive): such a message goes to ChatBotNode, no turn is played."""
        return "in game chat:" in message or "to Bot" in message

    def step(self, message=None):
        """One message of the browser -> one graph run, then the browser answers every frontend tool call with a ToolMessage.
        Default: what the scripted person sends (human_script), else the "Continue" button (src/app/page.tsx:2962).  A game
        message (vote page.tsx:302-305, button :272-275, input :2843) is logged by the reference's own
        process_human_action_if_needed (utils.py:310-358) inside InitialRouterNode; a chat message (page.tsx:341-349) is routed
        to ChatBotNode and plays no turn (the room's clock stands still).  Returns True when a turn was played."""
        from langchain_core.messages import HumanMessage, ToolMessage, AIMessage
        if message is None and self.human_script is not None:
            message = self.human_script(self)
        if message is None:
            message = "Continue"
        self.last_message = message
        self.policy.turn = self.turn
        self.policy.t_enter, self.policy.prev_phase = self.t_enter, self.prev_phase
        self._p_at_turn_start = p0 = self.state.get("current_phase_id", 0)
        self.state["messages"].append(HumanMessage(content=message))
        self.node_path = []
        if self.is_chat(message):
            try:
                asyncio.run(self._run_graph())
            except FileNotFoundError:
                # v3's ChatBotNode asks for prompt/chat_system_prompt.txt (game_agent_v3.py:407), a file the reference does
                # not ship (v2 reads chatbot_system_prompt.txt): that run fails and leaves the thread as it was
                assert self.version == "v3"
            assert self.node_path == ["InitialRouterNode", "ChatBotNode"], self.node_path
            return False
        asyncio.run(self._run_graph())
        self._stamp_person_entries()
        last = self.state["messages"][-1] if self.state["messages"] else None
        if isinstance(last, AIMessage):
            for tc in last.tool_calls:
                if tc["name"] == "createVotingPanel":
                    self.last_panel = (tc["args"]["votingId"], list(tc["args"]["options"]))
                self.state["messages"].append(ToolMessage(content="ok", tool_call_id=tc["id"]))
        self.state["messages"] = self.state["messages"][-40:]
        q = self.state.get("current_phase_id", 0)
        if q != p0:                       # the runtime's view of the clock
            self.t_enter, self.prev_phase = self.turn, p0
            if not self.table.by_id(q).branches and self.end_turn < 0:
                self.end_turn = self.turn
        self.turn += 1
        return True

    def vote_message(self, player: int, choice: int) -> str:
        """What the browser of seat `player` sends when it picks `choice` (a player id / a statement number) on the newest
        panel: `Player <id> voted "<option>" in voting <votingId>` (page.tsx:302-305); the statements phase has a text panel
        instead: `Input: <text>` (page.tsx:2843)."""
        from .. import dsl_table as T
        ph = self.table.by_id(self.state.get("current_phase_id", 0))
        if ph.act == T.ACT_TT_STATEMENTS:
            return "Input: " + "; ".join(f"Statement {s} of Player {player}" for s in (1, 2, 3))
        voting_id = self.last_panel[0] if self.last_panel else "vote-none"
        if self.table.pack == T.PACK_WEREWOLF:
            ps = self.state["player_states"]
            option = str(ps[str(choice)].get("name") or f"Player {choice}")
        else:
            option = str(choice)
        return f'Player {player} voted "{option}" in voting {voting_id}'

    # ---- canonical projection (ints only; no strings, timestamps, versions)
    def project(self) -> List[int]:
        return project_state(self.table, self.state, self.t_enter, self.prev_phase, self.end_turn,
                             self.human_mask, self.policy.human_turns)


def project_state(table, state: dict, t_enter: int, prev_phase: int, end_turn: int, human_mask: int = 0, human_turns=None) -> List[int]:
    """[phase, prev_phase, phase0_done, end_turn] + per player 11 ints (+ detective memory, ww).

    Layout (also produced by oracle.py and the product's read_rooms):
      ww player: role team alive revealed can_vote secret eligible submitted target acted choice
      tt player: is_speaker submitted lie_index lie_revealed can_vote vote_choice has_voted
                 total_score rounds_as_speaker acted choice
    """
    from .. import dsl_table as T
    from .policy import RoomView
    v = RoomView(table, state, t_enter, prev_phase, human_mask, human_turns)
    cur = int(state.get("current_phase_id", 0))
    ph = table.by_id(cur)
    hist = state.get("phase_history", []) or []
    phase0_done = int(any(e.get("phase_id") == 0 for e in hist))
    out = [cur, v.prev_phase, phase0_done, end_turn]
    acts = v.visit_actions(ph)
    for i in range(v.n):
        acted, choice = (1, acts[i][1]) if i in acts else (0, 0)
        g = lambda f, d=0: v.get(i, f, d)           # canonical slot -> the DSL's own field; an undeclared slot reads as 0
        if table.pack == T.PACK_WEREWOLF:
            team = {"": 0, "villagers": 1, "werewolves": 2}[g("team", "")]
            if table.declared("wolf_chat_enabled"):   # derived slot (POLICY.md §3a): not in the layout, checked here
                assert bool(g("wolf_chat_enabled")) == (team == 2), "wolf_chat_enabled must equal team == werewolves"
            out += [v.role_class(i), team, int(bool(g("is_alive", True))), int(bool(g("role_revealed"))),
                    int(bool(g("can_vote"))), int(bool(g("has_secret_role"))),
                    int(bool(g("night_action_eligible"))), int(bool(g("night_action_submitted"))),
                    int(g("selected_target_id") or 0), acted, choice]
        else:
            out += [int(bool(g("is_speaker"))), int(bool(g("statements_submitted"))), int(g("lie_index") or 0),
                    int(bool(g("lie_revealed"))), int(bool(g("can_vote"))), int(g("vote_choice") or 0),
                    int(bool(g("has_voted"))), int(g("total_score") or 0), int(g("rounds_as_speaker") or 0),
                    acted, choice]
    if table.pack == T.PACK_WEREWOLF:
        holders = [i for i in range(v.n) if (v.get(i, "investigated_alignments") or {})]
        assert len(holders) <= 1, "policy keeps a single detective memory"
        kv, kw = v.known() if table.declared("investigated_alignments") else (0, 0)     # not a field of this DSL's rooms: 0
        out += [(1 if (kv >> i) & 1 else 0) + (2 if (kw >> i) & 1 else 0) for i in range(v.n)]
    return out
