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
                 dsl_variant=None):
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
        human = None
        if human_script is not None:
            def human(turn, view, _s=self):          # the script sees the canonical projection, like the tests do
                return human_script(_s.table, turn, _s.project(), n_players)
        self.policy = FixedPolicy(self.table, seed, room, human_mask, human)
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
                {"name": f"Bot {i + 1}", "gamePlayerId": i + 1, "isBot": True} for i in range(n_players)]},
        }
        self._p_at_turn_start = 0

    # ---- the stub LLM's dispatch: which node is asking is told by the bound tools
    def answer(self, tool_names) -> List[dict]:
        st, pol = self.state_in_node, self.policy
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

    def step(self):
        """One turn: the browser's "Continue" message (src/app/page.tsx:2962) -> one graph run,
        then the browser answers every frontend tool call with a ToolMessage."""
        from langchain_core.messages import HumanMessage, ToolMessage, AIMessage
        self.policy.turn = self.turn
        self.policy.t_enter, self.policy.prev_phase = self.t_enter, self.prev_phase
        self._p_at_turn_start = p0 = self.state.get("current_phase_id", 0)
        self.state["messages"].append(HumanMessage(content="Continue"))
        self.node_path = []
        asyncio.run(self._run_graph())
        last = self.state["messages"][-1] if self.state["messages"] else None
        if isinstance(last, AIMessage):
            for tc in last.tool_calls:
                self.state["messages"].append(ToolMessage(content="ok", tool_call_id=tc["id"]))
        self.state["messages"] = self.state["messages"][-40:]
        q = self.state.get("current_phase_id", 0)
        if q != p0:                       # the runtime's view of the clock
            self.t_enter, self.prev_phase = self.turn, p0
            if not self.table.by_id(q).branches and self.end_turn < 0:
                self.end_turn = self.turn
        self.turn += 1

    # ---- canonical projection (ints only; no strings, timestamps, versions)
    def project(self) -> List[int]:
        return project_state(self.table, self.state, self.t_enter, self.prev_phase, self.end_turn)


def project_state(table, state: dict, t_enter: int, prev_phase: int, end_turn: int) -> List[int]:
    """[phase, prev_phase, phase0_done, end_turn] + per player 11 ints (+ detective memory, ww).

    Layout (also produced by oracle.py and the product's read_rooms):
      ww player: role team alive revealed can_vote secret eligible submitted target acted choice
      tt player: is_speaker submitted lie_index lie_revealed can_vote vote_choice has_voted
                 total_score rounds_as_speaker acted choice
    """
    from .. import dsl_table as T
    from .policy import RoomView
    v = RoomView(table, state, t_enter, prev_phase)
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
