"""ORACLE (test infrastructure) - random target conditions in the DSL generator's grammar
(agent/prompt/dsl_phases_generation_prompt.txt:120-132), for the compiler-agreement and GPU-parity fuzz tests.
A condition is generated FOR a phase's action kind: every alternative keeps the literal that makes the kind
recognisable (role == 'Werewolf', can_vote == true, is_speaker == ...), the rest is random."""
from __future__ import annotations

import random
from typing import List

WW_BOOLS = ["is_alive", "can_vote", "role_revealed", "has_secret_role", "night_action_eligible", "night_action_submitted"]
WW_ROLES = ["Villager", "Werewolf", "Doctor", "Detective"]
TT_BOOLS = ["is_speaker", "statements_submitted", "lie_revealed", "can_vote", "has_voted"]
TT_NUMS = {"lie_index": 3, "vote_choice": 3, "total_score": 6, "rounds_as_speaker": 3}
# the literal that fixes the action kind (ge_step.h GE_ACT_*), per kind
ANCHOR = {1: "player.role == 'Werewolf'", 2: "player.role == 'Doctor'", 3: "player.role == 'Detective'",
          4: "player.can_vote == true", 5: "player.is_speaker == true", 6: "player.is_speaker == true",
          7: "player.is_speaker == false"}


def _bool_lit(rng: random.Random, field: str) -> str:
    form = rng.randrange(5)
    v = rng.choice(["true", "false"])
    if form == 0:
        return f"player.{field} == {v}"
    if form == 1:
        return f"player.{field} != {v}"
    if form == 2:
        return f"player.{field} in [{v}]"
    if form == 3:
        return f"player.{field} not in [{v}]"
    return f"player.{field} == {rng.choice(['0', '1'])}"


def _num_lit(rng: random.Random, field: str, top: int) -> str:
    op = rng.choice(["==", "!=", "<", "<=", ">", ">=", "in", "not in"])
    if op == "in":
        k = rng.randint(0, top)
        vals = sorted({rng.randint(0, top) for _ in range(rng.randint(1, 3))})
        if rng.random() < 0.5:
            vals = list(range(k, min(top, k + rng.randint(0, 2)) + 1))
        return f"player.{field} in [{', '.join(map(str, vals))}]"
    if op == "not in":
        k = rng.randint(0, top)
        return f"player.{field} not in [{', '.join(map(str, range(k, min(top, k + rng.randint(0, 2)) + 1)))}]"
    return f"player.{field} {op} {rng.randint(-1, top + 1)}"


def _ww_extra(rng: random.Random, act: int) -> str:
    r = rng.random()
    if r < 0.45:
        return _bool_lit(rng, rng.choice(WW_BOOLS))
    if r < 0.6:
        return f"player.team {rng.choice(['==', '!='])} '{rng.choice(['villagers', 'werewolves'])}'"
    if r < 0.7:
        return f"player.team {rng.choice(['in', 'not in'])} ['villagers', 'werewolves']"
    if r < 0.8 and act == 4:                      # role lists only where they cannot re-classify the action
        roles = rng.sample(WW_ROLES, rng.randint(2, 3))
        return "player.role %s [%s]" % (rng.choice(["in", "not in"]), ", ".join(f"'{x}'" for x in roles))
    return _num_lit(rng, "selected_target_id", 9)


def _tt_extra(rng: random.Random, act: int) -> str:
    if rng.random() < 0.5:
        return _bool_lit(rng, rng.choice([b for b in TT_BOOLS if b != "is_speaker"]))
    f = rng.choice(sorted(TT_NUMS))
    return _num_lit(rng, f, TT_NUMS[f])


def condition_for(rng: random.Random, pack: int, act: int) -> str:
    """A random condition whose every alternative is an `act` phase's."""
    alts: List[str] = []
    for _ in range(rng.choice([1, 1, 2, 2, 3])):
        lits = [ANCHOR[act]] + [(_ww_extra if pack == 1 else _tt_extra)(rng, act) for _ in range(rng.randint(0, 3))]
        rng.shuffle(lits)
        alts.append(" and ".join(lits))
    return " or ".join(alts)


def randomize_dsl(rng: random.Random, dsl: dict, pack: int, acts: dict) -> dict:
    """A copy of `dsl` in which every player_action phase got a random condition of its own kind.
    acts: {phase id (int): GE_ACT_*} of the base table."""
    import copy
    d = copy.deepcopy(dsl)
    for key, ph in d["phases"].items():
        cc = ph.get("completion_criteria") or {}
        if cc.get("type") == "player_action":
            cc["target_players"]["condition"] = condition_for(rng, pack, acts[int(key)])
            cc["wait_for"] = rng.choice(["single_player_choice", "all_players_action", "multiple_players_action"])
    return d
