"""Deterministic UI script: the frontend tool calls of a phase, without an LLM.

In the reference the ActionExecutor / UIUpdateNode LLM (agent/game_agent_v2.py:1243-1568,
game_agent_v3.py:879-1095) reads the current phase's `actions[].tools` from the DSL and calls the
frontend tools (`useCopilotAction` handlers, src/app/page.tsx:950-2500) with an audience:
`audience_type=true` for everyone, or `audience_type=false` + `audience_ids=[...]`
(CardRenderer.tsx:58-70 filters on exactly these two fields).  This module emits the same calls
deterministically from the DSL and a room's `player_states`:

  * audience groups = `declaration.audience_groups.<g>.selection_criteria` (ww:138-165), evaluated
    with the DSL's condition mini-language (`==`, `!=`, `in [...]`, `and`);
  * the tier of an action is read from its description ("TIER 1 - PUBLIC", "TIER 2 - GROUP",
    "TIER 3 - INDIVIDUAL", the convention of dsl_phases_generation_prompt.txt); group / individual
    actions are matched to an audience by the role or group the description names.

Every call carries the parameters its frontend handler requires (game_engine_amd/frontend_tools.json:
the parameter lists of page.tsx's useCopilotAction blocks, extracted by
tools/extract_frontend_tools.py - e.g. createVotingPanel{name, votingId, options[], position}
page.tsx:1146-1157, markPlayerDead{playerId, playerName} :1256-1262, clearCanvas{exemptList?} :2418-2426)
and nothing a handler does not declare.  The argument VALUES are this build's deterministic script (the
reference leaves them to the LLM): item names from the phase, a fixed grid plan for positions, options and
contents from the room's state.

Host-side and per room (nothing here is on the batch hot path); no game state is changed.
"""
from __future__ import annotations

import json
import os
import functools
import re
from typing import Any, Callable, Dict, List, Optional

_TERM = re.compile(r"^\s*player\.(\w+)\s*(==|!=|<=|>=|<|>|not\s+in|in)\s*(.+?)\s*$", re.S | re.I)


def _literal(text: str) -> Any:
    t = text.strip()
    if t.lower() == "true":
        return True
    if t.lower() == "false":
        return False
    if len(t) >= 2 and (t[0] == t[-1]) and t[0] in "'\"":
        return t[1:-1]
    return int(t)


def _split_outside(expr: str, word: str) -> List[str]:
    """Split on a blank-delimited keyword outside quotes and brackets."""
    out, cur, quote, depth, i = [], "", "", 0, 0
    pat = f" {word} "
    while i < len(expr):
        c = expr[i]
        if quote:
            quote = "" if c == quote else quote
        elif c in "'\"":
            quote = c
        elif c == "[":
            depth += 1
        elif c == "]":
            depth -= 1
        elif depth == 0 and expr[i:i + len(pat)].lower() == pat:
            out.append(cur)
            cur = ""
            i += len(pat)
            continue
        cur += c
        i += 1
    out.append(cur)
    return out


@functools.lru_cache(maxsize=1024)                  # a DSL's criteria are compiled once, not once per turn
def compile_criteria(expr: str) -> Callable[[Dict[str, Any]], bool]:
    """`player.team == 'werewolves' and player.is_alive == true` -> predicate over one player's state dict.
    The same grammar as phase target conditions (dsl_phases_generation_prompt.txt:120-132): == != < <= > >=,
    `in [a, b]` / `not in [a, b]`, terms joined by `and`, alternatives by `or` (and binds tighter; no parentheses).
    Evaluated on the dict, so any declared field may appear."""
    flat = " ".join(expr.split())
    if re.search(r"[()]", re.sub(r"'[^']*'|\"[^\"]*\"", "", flat)):
        raise ValueError(f"unsupported selection criterion (parentheses): {expr!r}")
    clauses = []
    for alt in _split_outside(flat, "or"):
        terms = []
        for part in _split_outside(alt, "and"):
            m = _TERM.match(part)
            if not m:
                raise ValueError(f"unsupported selection criterion: {part!r}")
            field, op, rhs = m.group(1), " ".join(m.group(2).lower().split()), m.group(3)
            if op in ("in", "not in"):
                inner = rhs.strip()
                if not (inner.startswith("[") and inner.endswith("]")):
                    raise ValueError(f"unsupported list literal: {rhs!r}")
                try:
                    val: Any = [_literal(x) for x in inner[1:-1].split(",") if x.strip()]
                except ValueError:
                    raise ValueError(f"unsupported literal in: {part!r}") from None
            else:
                try:
                    val = _literal(rhs)
                except ValueError:
                    raise ValueError(f"unsupported literal in: {part!r}") from None
                if op in ("<", "<=", ">", ">=") and (isinstance(val, bool) or not isinstance(val, int)):
                    raise ValueError(f"unsupported comparison: {part!r}")
            terms.append((field, op, val))
        clauses.append(terms)

    def holds(player: Dict[str, Any], field: str, op: str, val: Any) -> bool:
        have = player.get(field)
        if op in ("==", "!="):
            return (have == val and type(have) is type(val)) != (op == "!=")
        if op in ("in", "not in"):
            return any(have == v and type(have) is type(v) for v in val) != (op == "not in")
        if isinstance(have, bool) or not isinstance(have, (int, float)):
            return False
        return {"<": have < val, "<=": have <= val, ">": have > val, ">=": have >= val}[op]

    def pred(player: Dict[str, Any]) -> bool:
        return any(all(holds(player, f, o, v) for f, o, v in terms) for terms in clauses)

    return pred


def audience_groups(dsl: dict, player_states: Dict[str, Dict[str, Any]]) -> Dict[str, List[str]]:
    """group name -> player ids (strings, ascending) per declaration.audience_groups."""
    out: Dict[str, List[str]] = {}
    groups = (dsl.get("declaration") or {}).get("audience_groups") or {}
    ids = sorted(player_states, key=int)
    for name, g in groups.items():
        pred = compile_criteria(g.get("selection_criteria", ""))
        out[name] = [pid for pid in ids if pred(player_states[pid])]
    return out


def _audience_for(desc: str, player_states: Dict[str, Dict[str, Any]], groups: Dict[str, List[str]]) -> Optional[List[str]]:
    """Player ids a GROUP / INDIVIDUAL action addresses, from what its description names."""
    d = desc.lower()
    ids = sorted(player_states, key=int)
    alive = [p for p in ids if player_states[p].get("is_alive", True)]

    def role(name: str) -> List[str]:
        return [p for p in alive if str(player_states[p].get("role", "")).lower() == name]

    if "non-werewol" in d:
        return [p for p in alive if player_states[p].get("team") != "werewolves"]
    if "werewol" in d:
        return groups.get("werewolves", [p for p in alive if player_states[p].get("team") == "werewolves"])
    for r in ("doctor", "detective"):
        if f"except the {r}" in d:
            return [p for p in alive if p not in role(r)]
        if r in d:
            return role(r)
    if "eliminated players" in d or "dead players" in d:
        return groups.get("dead_players", [p for p in ids if not player_states[p].get("is_alive", True)])
    if "eligible voters" in d or "voters" in d:
        return groups.get("voters", [p for p in alive if player_states[p].get("can_vote", True)])
    if "non-speaker" in d:
        return [p for p in ids if not player_states[p].get("is_speaker")]
    if "speaker" in d:
        return [p for p in ids if player_states[p].get("is_speaker")]
    return None


_TOOLS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_tools.json")
_tools_cache: Optional[Dict[str, Any]] = None


def frontend_tools() -> Dict[str, List[List[Any]]]:
    """{tool: [[param, type, required], ...]} of the frontend handlers (page.tsx useCopilotAction blocks)."""
    global _tools_cache
    if _tools_cache is None:
        with open(_TOOLS_PATH, encoding="utf-8") as f:
            _tools_cache = json.load(f)
    return _tools_cache["tools"]


def validate_call(call: Dict[str, Any]) -> List[str]:
    """Problems of a frontend tool call against its handler's declared parameters ([] = fine)."""
    spec = frontend_tools().get(call["name"])
    if spec is None:
        return [f"no frontend handler named {call['name']}"]
    known = {p[0] for p in spec}
    out = [f"{call['name']}: missing required parameter {p[0]}" for p in spec if p[2] and call["args"].get(p[0]) in (None, "")]
    out += [f"{call['name']}: unknown parameter {k}" for k in call["args"] if k not in known]
    return out


# the fixed grid plan: where each kind of component goes (the handlers' `position` select lists)
_POSITION = {"createPhaseIndicator": "top-center", "createTextDisplay": "center", "createVotingPanel": "center",
             "createResultDisplay": "center", "createCharacterCard": "bottom-center", "createScoreBoard": "top-right",
             "createTurnIndicator": "top-left", "createStatementBoard": "middle-left"}
_DEATH_POSITIONS = ["bottom-left", "bottom-center", "bottom-right", "middle-left", "middle-right", "top-left", "top-right"]
_LABEL = {"createPhaseIndicator": "phase", "createTextDisplay": "text", "createVotingPanel": "vote", "createResultDisplay": "result",
          "createCharacterCard": "role card", "createScoreBoard": "scores", "createTurnIndicator": "turn", "createStatementBoard": "statements",
          "createAvatarSet": "avatars", "createTimer": "timer", "createDeathMarker": "death"}
# item types a clearCanvas description asks to keep (src/lib/canvas/types.ts item types)
_EXEMPT = (("death marker", "death_marker"), ("elimination indicator", "death_marker"), ("scoreboard", "score_board"),
           ("score board", "score_board"))
DISCUSSION_SECONDS = 60          # the DSL's timer phases give no duration (ww:6, 14; tt:4): a fixed one


def _plain(desc: str) -> str:
    """An action description without its 'TIER n - AUDIENCE:' prefix."""
    return re.sub(r"^\s*TIER\s*\d\s*-\s*\w+\s*:\s*", "", desc).strip()


def _pname(ps: Dict[str, Dict[str, Any]], pid: str) -> str:
    return str(ps.get(pid, {}).get("name") or f"Player {pid}")


def _vote_options(act: int, ps: Dict[str, Dict[str, Any]], voters: Optional[List[str]]) -> List[str]:
    """What a voting panel offers, by the phase's action kind (POLICY.md §3 candidates)."""
    ids = sorted(ps, key=int)
    alive = [p for p in ids if ps[p].get("is_alive", True)]
    if act == 1:                                              # WOLF_TARGET: living non-werewolves
        return [_pname(ps, p) for p in alive if ps[p].get("team") != "werewolves"]
    if act in (2, 4):                                         # DOCTOR_PROTECT (self allowed) / DAY_VOTE: the living
        return [_pname(ps, p) for p in alive]
    if act == 3:                                              # DETECTIVE: the living, except the investigator
        return [_pname(ps, p) for p in alive if not voters or p not in voters]
    return ["1", "2", "3"]                                    # two-truths: statement numbers


def ui_tool_calls(dsl: dict, agent_state: Dict[str, Any], table: Any = None, turn: int = 0,
                  deaths: Optional[List[str]] = None, items: Optional[List[Dict[str, Any]]] = None) -> List[Dict[str, Any]]:
    """Frontend tool calls for the room's current phase, in DSL order, with every required parameter.

    agent_state: what RoomBatch.agent_state() / RoomService return (current_phase_id, player_states, ...);
    table: the room's GameTable (the phase's action kind decides a voting panel's options; compiled from
           `dsl` when omitted);
    turn: the turn just stepped (makes votingId unique per visit);
    deaths: ids of the players eliminated by this turn's transition (markPlayerDead / createDeathMarker);
    items: the frontend's current canvas items ([{id, type, ...}], AgentState.items) - clearCanvas's
           exemptList names the ids of the item types its description asks to keep."""
    phases = dsl.get("phases") or {}
    pid = agent_state["current_phase_id"]
    phase = phases.get(pid) or phases.get(str(pid)) or {}          # int or str keys (utils.py:29)
    pname = phase.get("name", f"Phase {pid}")
    ps = agent_state["player_states"]
    ids = sorted(ps, key=int)
    groups = audience_groups(dsl, ps)
    if table is None:
        from .stepper import GameTable
        table = GameTable(dsl)
    act = next((r["act"] for r in table.rows() if r["phase_id"] == pid), 0)
    deaths = [str(d) for d in (deaths or [])]
    dead_before = len([p for p in ids if not ps[p].get("is_alive", True)]) - len(deaths)
    calls: List[Dict[str, Any]] = []

    def audience(args: Dict[str, Any], aud: Optional[List[str]]) -> Dict[str, Any]:
        if aud is None:
            args["audience_type"] = True
        else:
            args["audience_type"] = False
            args["audience_ids"] = list(aud)
        return args

    for k, action in enumerate(phase.get("actions") or []):
        desc = action.get("description", "")
        text = _plain(desc)
        tier = 1
        m = re.search(r"TIER\s*(\d)", desc)
        if m:
            tier = int(m.group(1))
        per_player = tier >= 3 and "each player" in desc.lower()
        aud = _audience_for(desc, ps, groups) if tier >= 2 else None
        # untiered descriptions (two-truths) name a private audience in words
        if tier == 1 and re.search(r"private|individual audience|eligible voters only", desc, re.I):
            aud = _audience_for(desc, ps, groups)
        for tool in action.get("tools") or []:
            name = f"{pname} - {_LABEL.get(tool, tool)}"
            if tool == "clearCanvas":
                args: Dict[str, Any] = {}
                if items is not None:
                    keep = {t for key, t in _EXEMPT if key in desc.lower() and "no exemption" not in desc.lower()}
                    args["exemptList"] = [str(it["id"]) for it in items if it.get("type") in keep]
                calls.append({"name": tool, "args": args})
            elif tool == "createPhaseIndicator":
                calls.append({"name": tool, "args": audience({"name": name, "currentPhase": pname, "position": _POSITION[tool],
                                                               "description": phase.get("description", text)}, aud)})
            elif tool == "createTextDisplay":
                calls.append({"name": tool, "args": audience({"name": f"{name} {k}", "content": text, "position": _POSITION[tool],
                                                               "title": pname, "type": "info"}, aud)})
            elif tool == "createAvatarSet":
                calls.append({"name": tool, "args": audience({"name": name, "avatarType": "human"}, None)})
            elif tool == "createCharacterCard":
                targets = ids if per_player else (aud if aud is not None else ids)
                for p in targets:                                     # one private card per player (ww:206-210)
                    calls.append({"name": tool, "args": audience({"name": f"{name} {p}", "role": ps[p].get("role", "") or "unassigned",
                                                                   "position": _POSITION[tool], "description": text}, [p])})
            elif tool == "createVotingPanel":
                calls.append({"name": tool, "args": audience({"name": name, "votingId": f"vote-p{pid}-t{turn}",
                                                               "options": _vote_options(act, ps, aud), "position": _POSITION[tool],
                                                               "title": text}, aud)})
            elif tool == "createResultDisplay":
                calls.append({"name": tool, "args": {"name": name, "content": _result_text(dsl, agent_state, text, deaths), "position": _POSITION[tool]}})
            elif tool == "markPlayerDead":
                for p in deaths:
                    calls.append({"name": tool, "args": {"playerId": p, "playerName": _pname(ps, p)}})
            elif tool == "createDeathMarker":
                for j, p in enumerate(deaths):
                    calls.append({"name": tool, "args": audience({"name": f"{_pname(ps, p)} - eliminated", "playerName": _pname(ps, p), "playerId": p,
                                                                   "position": _DEATH_POSITIONS[(dead_before + j) % len(_DEATH_POSITIONS)]}, None)})
            elif tool == "createTimer":
                calls.append({"name": tool, "args": {"name": name, "duration": DISCUSSION_SECONDS, "label": text}})
            elif tool == "createScoreBoard":
                entries = [{"id": p, "name": _pname(ps, p), "score": int(ps[p].get("total_score", 0))} for p in ids]
                calls.append({"name": tool, "args": audience({"name": name, "title": "Scores", "entries": entries, "sort": "desc",
                                                               "position": _POSITION[tool]}, None)})
            elif tool == "createTurnIndicator":
                sp = next((p for p in ids if ps[p].get("is_speaker")), ids[0])
                calls.append({"name": tool, "args": audience({"name": name, "currentPlayerId": sp, "playerName": _pname(ps, sp),
                                                               "label": "Speaker", "position": _POSITION[tool]}, None)})
            elif tool == "createStatementBoard":
                sp = next((p for p in ids if ps[p].get("is_speaker")), None)
                st = (ps[sp].get("statements") or {}) if sp else {}
                statements = [st.get(str(i)) or f"Statement {i} of Player {sp}" for i in (1, 2, 3)] if sp else []
                args = {"name": name, "statements": statements, "locked": True, "position": _POSITION[tool]}
                if sp and ps[sp].get("lie_revealed") and ps[sp].get("lie_index"):
                    args["highlightIndex"] = int(ps[sp]["lie_index"]) - 1
                calls.append({"name": tool, "args": audience(args, None)})
            elif tool == "createTextInputPanel":                   # the handler takes no audience (page.tsx:371-386)
                calls.append({"name": tool, "args": {"title": pname, "placeholder": text}})
            else:                                                  # a tool this script has no builder for: name + audience only
                calls.append({"name": tool, "args": audience({"name": name}, aud)})
    return calls


def _result_text(dsl: dict, agent_state: Dict[str, Any], text: str, deaths: List[str]) -> str:
    """Content of a createResultDisplay: what the phase announces, from the room's state."""
    ps = agent_state["player_states"]
    ids = sorted(ps, key=int)
    if any("is_alive" in ps[p] for p in ids):                     # werewolf
        alive = [p for p in ids if ps[p].get("is_alive", True)]
        wolves = [p for p in alive if ps[p].get("team") == "werewolves"]
        phase = (dsl.get("phases") or {}).get(agent_state["current_phase_id"]) or (dsl.get("phases") or {}).get(str(agent_state["current_phase_id"])) or {}
        if not phase.get("next_phase"):                            # the terminal phase: who won
            side = "Villagers win - every werewolf is eliminated." if not wolves else "Werewolves win - they are no longer outnumbered."
            return f"{side} Survivors: " + (", ".join(_pname(ps, p) for p in alive) or "none") + "."
        if deaths:
            return " ".join(f"{_pname(ps, p)} was eliminated." for p in deaths)
        return "No one was eliminated."
    sp = next((p for p in ids if ps[p].get("is_speaker")), None)   # two-truths
    if sp and ps[sp].get("lie_revealed"):
        return f"The lie was statement {ps[sp].get('lie_index')} of {_pname(ps, sp)}."
    if all(int(ps[p].get("rounds_as_speaker", 0)) > 0 for p in ids):
        best = max(int(ps[p].get("total_score", 0)) for p in ids)
        return "Final scores - " + ", ".join(f"{_pname(ps, p)}: {ps[p].get('total_score', 0)}" for p in ids) + \
            ". Winner: " + ", ".join(_pname(ps, p) for p in ids if int(ps[p].get("total_score", 0)) == best) + "."
    return text
