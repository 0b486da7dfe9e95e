"""ORACLE (test infrastructure) - variants of the two shipped game DSLs that use the rest of the condition
grammar the reference's DSL generator is told to write (agent/prompt/dsl_phases_generation_prompt.txt:106-150):
`in [...]`, `!=`, `<`, `<=`, `>`, `>=` over declared `num` fields, `or`, and the three `wait_for` kinds.
The shipped games only use `==` joined by `and`; these variants are what pins the generic path: the
reference's own nodes are run on them (oracle/refharness/make_golden.py::variant_cases -> tests/golden/
traj_variant_*.json), and the oracle and the product are compared on them.

Each variant is a function of the base DSL dict (tests/golden/dsl/<game>.json): data in, data out."""
from __future__ import annotations

import copy
from typing import Callable, Dict, Tuple


def _set_condition(dsl: dict, phase_id, cond: str, wait_for: str = None) -> None:
    ph = dsl["phases"].get(phase_id) or dsl["phases"].get(str(phase_id)) or dsl["phases"][int(phase_id)]
    cc = ph["completion_criteria"]
    cc["target_players"]["condition"] = cond
    if wait_for:
        cc["wait_for"] = wait_for


def ww_generic(base: dict) -> dict:
    """Werewolf: role lists, !=, an `or` of two conjunctions and a numeric comparison that changes who votes by day
    (only players whose selected_target_id is below 5: villagers hold 0, night actors whatever they picked)."""
    d = copy.deepcopy(base)
    for pid in (2, 10):
        _set_condition(d, pid, "player.role in ['Werewolf'] and player.is_alive == true and player.team != 'villagers'",
                       "all_players_action")
    for pid in (3, 11):
        _set_condition(d, pid, "player.role in ['Doctor', 'Medic'] and player.is_alive != false")
    for pid in (4, 12):
        _set_condition(d, pid, "player.role == 'Detective' and player.is_alive == true and player.selected_target_id == 0 "
                               "or player.role == 'Detective' and player.selected_target_id > 0")
    for pid in (7, 15):
        _set_condition(d, pid, "player.can_vote == true and player.is_alive == true and player.selected_target_id < 5",
                       "multiple_players_action")
    return d


def tt_generic(base: dict) -> dict:
    """Two Truths: only players who are behind (score <= 1) or have already spoken vote; boolean list; not in."""
    d = copy.deepcopy(base)
    _set_condition(d, 2, "player.is_speaker == true and player.statements_submitted != true")
    _set_condition(d, 3, "player.is_speaker in [true] and player.lie_index not in [1, 2, 3]", "single_player_choice")
    _set_condition(d, 5, "player.is_speaker == false and player.total_score <= 1 "
                         "or player.is_speaker == false and player.rounds_as_speaker >= 1", "all_players_action")
    return d


def ww_extra_fields(base: dict) -> dict:
    """Werewolf with two declared fields the rule pack does not model (`suspicion: num`, `tier: string`).  Nobody writes
    them under the fixed policy, so conditions on them are constants: the day vote's is always true, the Doctor's never
    (nobody is ever protected), the Detective's first alternative never, its second as in the shipped game."""
    d = copy.deepcopy(base)
    decl = d["declaration"]
    decl["player_states"]["suspicion"] = {"type": "num", "description": "how suspicious the table finds the player", "example": 3}
    decl["player_states"]["tier"] = {"type": "string", "description": "lobby tier", "example": "gold"}
    for tmpl in decl["player_states_template"]["player_states"].values():
        tmpl["suspicion"] = 3
        tmpl["tier"] = "gold"
    for pid in (3, 11):
        _set_condition(d, pid, "player.role == 'Doctor' and player.is_alive == true and player.tier != 'gold'")
    for pid in (4, 12):
        _set_condition(d, pid, "player.role == 'Detective' and player.suspicion > 3 or player.role == 'Detective' and player.is_alive == true "
                               "and player.tier in ['gold', 'silver']")
    for pid in (7, 15):
        _set_condition(d, pid, "player.can_vote == true and player.is_alive == true and player.suspicion >= 2 and player.suspicion not in [4, 5]")
    return d


def ww_minimal_schema(base: dict) -> dict:
    """Werewolf declaring only what its conditions need (name, role, team, is_alive, can_vote): every other slot of the
    rule pack - role_revealed, has_secret_role, night eligibility / submitted, the selected target, the Detective's memory -
    stays engine state and never appears in player_states (POLICY.md 3a).  The night's choices then live in the action log
    only, as in the reference's earlier draft of the game."""
    d = copy.deepcopy(base)
    decl = d["declaration"]
    drop = ("role_revealed", "has_secret_role", "night_action_eligible", "night_action_submitted", "selected_target_id",
            "investigated_alignments")
    for f in drop:
        decl["player_states"].pop(f, None)
        for block in ("player_states_template", "players_example"):
            for rec in ((decl.get(block) or {}).get("player_states") or {}).values():
                rec.pop(f, None)
    for g in ("night_actors", "secret_holders"):
        decl["audience_groups"].pop(g, None)
    return d


# name -> (base game, builder, rounds)
VARIANTS: Dict[str, Tuple[str, Callable[[dict], dict], int]] = {
    "ww_generic": ("werewolf-(mafia)", ww_generic, 1),
    "tt_generic": ("two-truths-and-a-lie", tt_generic, 2),
    "ww_extra_fields": ("werewolf-(mafia)", ww_extra_fields, 1),
    "ww_minimal_schema": ("werewolf-(mafia)", ww_minimal_schema, 1),
}


def build(name: str, base: dict) -> dict:
    return VARIANTS[name][1](base)
