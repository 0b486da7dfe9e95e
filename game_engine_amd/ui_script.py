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

Host-side and per room (nothing here is on the batch hot path); no game state is changed.
"""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, List, Optional

_TERM = re.compile(r"^\s*player\.(\w+)\s*(==|!=|in)\s*(.+?)\s*$", re.S)


def _literal(text: str) -> Any:
    t = text.strip()
    if t.lower() == "true":
        return True
    if t.lower() == "false":
        return False
    if (t[0] == t[-1]) and t[0] in "'\"":
        return t[1:-1]
    return int(t)


def compile_criteria(expr: str) -> Callable[[Dict[str, Any]], bool]:
    """`player.team == 'werewolves' and player.is_alive == true` -> predicate over one player's
    state dict.  Supports ==, !=, `in [a, b]` and `and` (everything the shipped DSLs and the
    generator prompts use for selection criteria)."""
    terms = []
    for part in re.split(r"\s+and\s+", " ".join(expr.split())):
        m = _TERM.match(part)
        if not m:
            raise ValueError(f"unsupported selection criterion: {part!r}")
        field, op, rhs = m.groups()
        if op == "in":
            inner = rhs.strip()
            if not (inner.startswith("[") and inner.endswith("]")):
                raise ValueError(f"unsupported list literal: {rhs!r}")
            values = [_literal(x) for x in inner[1:-1].split(",") if x.strip()]
            terms.append((field, "in", values))
        else:
            terms.append((field, op, _literal(rhs)))

    def pred(player: Dict[str, Any]) -> bool:
        for field, op, val in terms:
            have = player.get(field)
            ok = (have in val) if op == "in" else (have == val)
            if op == "!=":
                ok = not ok
            if not ok:
                return False
        return True

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
        return groups.get("voters", alive)
    if "non-speaker" in d:
        return [p for p in ids if not player_states[p].get("is_speaker")]
    if "speaker" in d:
        return [p for p in ids if player_states[p].get("is_speaker")]
    return None


def ui_tool_calls(dsl: dict, agent_state: Dict[str, Any]) -> List[Dict[str, Any]]:
    """Frontend tool calls for the room's current phase, in DSL order.

    `agent_state`: what RoomBatch.agent_state() / the JS readRoom() return (current_phase_id,
    current_phase_name, player_states)."""
    phases = dsl.get("phases") or {}
    pid = agent_state["current_phase_id"]
    phase = phases.get(pid) or phases.get(str(pid)) or {}          # int or str keys (utils.py:29)
    ps = agent_state["player_states"]
    groups = audience_groups(dsl, ps)
    calls: List[Dict[str, Any]] = []
    for action in phase.get("actions") or []:
        desc = action.get("description", "")
        tier = 1
        m = re.search(r"TIER\s*(\d)", desc)
        if m:
            tier = int(m.group(1))
        for tool in action.get("tools") or []:
            if tool == "clearCanvas":
                calls.append({"name": tool, "args": {}})
                continue
            base = {"name": phase.get("name", f"Phase {pid}"), "description": desc}
            if tier >= 3 and "each player" in desc.lower():
                for p in sorted(ps, key=int):                        # one private component per player (ww:206-210)
                    calls.append({"name": tool, "args": dict(base, audience_type=False, audience_ids=[p],
                                                             role=ps[p].get("role", ""))})
                continue
            aud = _audience_for(desc, ps, groups) if tier >= 2 else None
            if aud is None:
                calls.append({"name": tool, "args": dict(base, audience_type=True)})
            else:
                calls.append({"name": tool, "args": dict(base, audience_type=False, audience_ids=aud)})
    return calls
