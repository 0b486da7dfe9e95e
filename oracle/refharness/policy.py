"""ORACLE (test infrastructure) — FixedPolicy: the deterministic stand-in for every LLM call.

BASELINE.json config 1: "LLM stubbed to fixed policy".  The reference makes four
LLM calls per turn (game_agent_v2.py:571 BotBehaviorNode, :1108 PhaseNode,
:744 RefereeNode, :1461 ActionExecutor); this object answers each with the tool
calls POLICY.md prescribes, reading only what the node itself shows the LLM:
`player_states`, `playerActions`, `game_notes`, `current_phase_id`, the DSL —
plus the turn index the harness supplies (the policy's clock).

Conventions the policy writes into the reference's own state so that it can stay
stateless (all are plain arguments of the reference's backend tools):
  * every bot action text starts with "[t=<turn>|c=<choice>] "   (bt:144-157 `actions`)
  * every transition leaves a note  "[t=<turn>] phase <p> -> <q>"  (bt:42-84, PHASE_STATUS;
    informational: v3 drops game_notes from its update, v3:863-874, so nothing parses it)
The harness (the runtime around the graph) supplies the clock: the turn index, the
turn in which the current phase was entered and the phase it was entered from, all
observable from `current_phase_id` between graph runs - and, for every log entry that
`process_human_action_if_needed` (agent/tools/utils.py:310-358) filed from a person's
message, the turn of the graph run that filed it (`human_turns`; the reference stamps
such an entry with wall-clock milliseconds only).

A person's action is NOT a tool call of this policy: the walker sends the frontend's own
message string (src/app/page.tsx:302-305 vote, :2843 input), the reference's
InitialRouterNode logs it verbatim under Player 1 with phase 0's name (its state keys
`currentPhaseId` / `playerStates` do not exist, v2:324-331), and the policy reads the
person's choice back off that entry (`RoomView.person_entries`, POLICY.md 3b).

This file runs only in the build container next to /root/reference; nothing here
travels to the GPU box except the golden vectors it produces.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Tuple

from .. import dsl_table as T
from .. import rng

_TAG = re.compile(r"^\[t=(\d+)\|c=(\d+)\] ")
# the frontend's vote message (src/app/page.tsx:302-305) with this build's votingId scheme (POLICY.md 3b)
_VOTE = re.compile(r'^Player ([0-9]+) voted "(.*)" in voting vote-p([0-9]+)-t([0-9]+)\Z', re.S)
_INPUT = re.compile(r"^Input: ", re.S)          # src/app/page.tsx:2843


def _pids(player_states: dict) -> List[str]:
    return sorted(player_states.keys(), key=int)


class RoomView:
    """Integer view of the dict state the reference shows its LLM."""

    def __init__(self, table: T.Table, state: dict, t_enter: int = -1, prev_phase: int = 0,
                 human_mask: int = 0, human_turns: Optional[Dict[str, int]] = None):
        self.table = table
        self.human_mask = human_mask              # host-driven seats (bit i: player i+1)
        self.human_turns = human_turns or {}      # id of a person's entry in Player 1's log -> turn of the run that filed it
        self.ps: Dict[str, dict] = state.get("player_states", {}) or {}
        self.ids = _pids(self.ps)
        self.n = len(self.ids)
        self.actions = state.get("playerActions", {}) or {}
        self.t_enter = t_enter          # turn in which the current phase was entered
        self.prev_phase = prev_phase    # phase it was entered from

    # ---- per-player fields (0-based index i <-> player id str(i+1)); `fld` is a canonical slot name, the dict
    # ---- holds it under the name the DSL declares (dsl_table.bind_fields); an undeclared slot reads as `default`
    def get(self, i: int, fld: str, default=None):
        name = self.table.declared(fld) if fld in self.table.names else fld
        return self.ps[self.ids[i]].get(name, default) if name else default

    def alive(self, i: int) -> bool:
        return bool(self.get(i, "is_alive", True))

    def role_class(self, i: int) -> int:
        r = self.get(i, "role", "")
        return self.table.role_names.index(r) if r in self.table.role_names else 0

    def is_wolf_team(self, i: int) -> bool:
        return self.get(i, "team", "") == "werewolves"

    def mask(self, pred) -> int:
        m = 0
        for i in range(self.n):
            if pred(i):
                m |= 1 << i
        return m

    def term_true(self, i: int, term: T.Term) -> bool:
        v = self.get(i, term.field)
        ok = (v == term.value) if not isinstance(term.value, bool) else (bool(v) is True)
        return ok != term.negate

    def base_true(self, i: int, base: int) -> bool:
        """Base predicate `base` of the rule pack (POLICY.md §3 numbering), read off the dict state."""
        if self.table.pack == T.PACK_WEREWOLF:
            for (fld, val), b in T.WW_BASE.items():
                if b == base:
                    if fld == "role":
                        return self.role_class(i) == val
                    return (self.get(i, fld) == val) if isinstance(val, str) else bool(self.get(i, fld))
            raise AssertionError(base)
        for (fld, _), b in T.TT_BASE.items():
            if b == base:
                return bool(self.get(i, fld))
        raise AssertionError(base)

    def literal_true(self, i: int, l: T.Literal) -> bool:
        if l.kind == "base":
            ok = any(self.base_true(i, b) for b in l.bases)
        else:
            ok = l.lo <= int(self.get(i, l.field) or 0) <= l.hi
        return ok != l.negate

    def targets(self, ph: T.Phase) -> int:
        """completion_criteria.target_players.condition AND alive
        (bot_behavior_system_prompt.txt:21-31: dead players never act).  The condition is read in its
        clause form: OR of AND-clauses of literals (dsl_phases_generation_prompt.txt:120-132 grammar)."""
        return self.mask(lambda i: self.alive(i) and (not ph.clauses or any(
            all(self.literal_true(i, l) for l in clause) for clause in ph.clauses)))

    # ---- what a voting panel of `ph` offers (POLICY.md 3b; the product's ui_script builds the same list)
    def name_of(self, i: int) -> str:
        return str(self.ps[self.ids[i]].get("name") or f"Player {self.ids[i]}")

    def panel_options(self, ph: T.Phase) -> List[str]:
        if self.table.pack != T.PACK_WEREWOLF:
            return ["1", "2", "3"]
        if ph.act == T.ACT_WOLF_TARGET:
            keep = lambda i: self.alive(i) and not self.is_wolf_team(i)
        elif ph.act == T.ACT_DETECTIVE:
            keep = lambda i: self.alive(i) and self.role_class(i) != T.ROLE_DETECTIVE
        else:
            keep = self.alive
        return [self.name_of(i) for i in range(self.n) if keep(i)]

    def option_choice(self, option: str) -> int:
        """1-based choice an option string stands for (a player's name -> the lowest id carrying it; a statement number), 0: none."""
        if self.table.pack != T.PACK_WEREWOLF:
            return int(option) if option in ("1", "2", "3") else 0
        for i in range(self.n):
            if self.name_of(i) == option:
                return i + 1
        return 0

    def person_entries(self, ph: T.Phase, ever: bool = False):
        """(seat index, turn, choice) of every message of a person that counts as an action of `ph` (POLICY.md 3b), in log order.
        The reference files every such message under Player 1 with phase 0's name (utils.py:331-349 via v2:324-331), so the
        phase is read from the votingId, the seat from the message, the turn from the runtime's clock (`human_turns`)."""
        rec = self.actions.get("1") or {}
        tgt = self.targets(ph) if not ever else 0
        done = set()
        for aid in sorted(rec.get("actions", {}), key=int):
            text = rec["actions"][aid].get("action", "")
            turn = self.human_turns.get(str(aid))
            if turn is None or _TAG.match(text):
                continue
            if turn <= self.t_enter and not ever:
                continue
            m = _VOTE.match(text)
            if m:
                seat, option, p_id, p_turn = int(m.group(1)) - 1, m.group(2), int(m.group(3)), int(m.group(4))
                if ph.act in (T.ACT_NONE, T.ACT_TT_STATEMENTS) or "createVotingPanel" not in ph.tools or \
                        p_id != ph.id or p_turn + 1 != turn:
                    continue                                  # not a panel of this phase / not the newest panel
                if not (0 <= seat < self.n) or not (self.human_mask >> seat) & 1:
                    continue
                choice = self.option_choice(option)           # what ge_batch_inject_action accepts: a living player / 1..3
                if choice and not ever and self.table.pack == T.PACK_WEREWOLF and not self.alive(choice - 1):
                    continue
            elif _INPUT.match(text) and ph.act == T.ACT_TT_STATEMENTS:
                # a text panel carries no seat: the pending host-driven target with the lowest id
                seats = [i for i in range(self.n) if (self.human_mask >> i) & 1 and (ever or ((tgt >> i) & 1 and i not in done))]
                if not seats:
                    continue
                seat, choice = seats[0], 1
            else:
                continue
            if not choice:
                continue
            if not ever:
                if not (tgt >> seat) & 1 or seat in done:     # only a living target's first action of the visit counts
                    continue
                done.add(seat)
            yield seat, turn, choice

    def entries(self, ph: T.Phase, ever: bool = False):
        """(player index, turn, choice) of every logged action of `ph` - bots' tagged entries and persons' messages."""
        for i, pid in enumerate(self.ids):
            rec = self.actions.get(pid)
            if not rec:
                continue
            for aid in sorted(rec.get("actions", {}), key=int):
                a = rec["actions"][aid]
                m = _TAG.match(a.get("action", ""))
                if not m or a.get("phase") != ph.name:
                    continue
                turn = int(m.group(1))
                if turn <= self.t_enter and not ever:
                    continue
                yield i, turn, int(m.group(2))
        if self.human_mask:
            yield from self.person_entries(ph, ever)

    def visit_actions(self, ph: T.Phase, only_turn: Optional[int] = None, ever: bool = False) -> Dict[int, Tuple[int, int]]:
        """player index -> (turn, choice) of the latest action in this visit of `ph`
        (referee_system_prompt_1.txt:19: latest action matching the current phase name); `ever`: in any visit."""
        out: Dict[int, Tuple[int, int]] = {}
        for i, turn, choice in self.entries(ph, ever):
            if only_turn is not None and turn != only_turn:
                continue
            if i not in out or turn >= out[i][0]:
                out[i] = (turn, choice)
        return out

    def known(self) -> Tuple[int, int]:
        """detective memory as (known_villagers_mask, known_werewolves_mask)."""
        kv = kw = 0
        if self.table.pack == T.PACK_WEREWOLF and not self.table.declared("investigated_alignments"):
            # the DSL keeps no memory field: what the Detective has learnt is what it has investigated so far - every
            # "investigated Player c" of the action log - and a player's team never changes once dealt
            for ph in self.table.phases:
                if ph.act != T.ACT_DETECTIVE:
                    continue
                for _i, _turn, choice in self.entries(ph, ever=True):
                    c = choice - 1
                    if self.get(c, "team", "") == "werewolves":
                        kw |= 1 << c
                    else:
                        kv |= 1 << c
            return kv, kw
        for i in range(self.n):
            mem = self.get(i, "investigated_alignments", {}) or {}
            for k, team in mem.items():
                if team == "werewolves":
                    kw |= 1 << (int(k) - 1)
                else:
                    kv |= 1 << (int(k) - 1)
        return kv, kw


def plurality(votes: List[int], n: int) -> int:
    """1-based id with most votes, ties -> lowest id; 0 if no votes
    (referee_system_prompt_2.txt:19-22 'most voted dies'; tie-break is a build decision)."""
    cnt = [0] * (n + 1)
    for v in votes:
        if 1 <= v <= n:
            cnt[v] += 1
    best, arg = 0, 0
    for k in range(1, n + 1):
        if cnt[k] > best:
            best, arg = cnt[k], k
    return arg


class FixedPolicy:
    def __init__(self, table: T.Table, seed: int, room: int, human_mask: int = 0, human=None):
        self.table = table
        self.rkey = rng.room_key(seed, room)
        self.human_mask = human_mask      # players the bot policy never acts for (player 1 = the human)
        self.human_turns: Dict[str, int] = {}   # runtime clock: id of a person's entry in Player 1's log -> turn that filed it
        self.game = 0            # index of this room's game on its slot (steady-state chains; else 0)
        # clock, set by the walker before each graph run
        self.turn = 0
        self.t_enter = -1
        self.prev_phase = 0

    def _view(self, state: dict) -> RoomView:
        return RoomView(self.table, state, self.t_enter, self.prev_phase, self.human_mask, self.human_turns)

    # ------------------------------------------------------------------ bots
    def bot_actions(self, state: dict, p_id: int) -> List[dict]:
        """BotBehaviorNode's LLM (v2:523-571; v3:462-494)."""
        tb, t = self.table, self.turn
        ph = tb.by_id(p_id)
        if ph.completion != T.COMP_ACTION:
            return []
        v = self._view(state)
        tgt = v.targets(ph)
        acted = v.visit_actions(ph)
        tkey = rng.turn_key(self.rkey, t)
        alive = v.mask(v.alive)
        wolfteam = v.mask(v.is_wolf_team)
        kv, kw = v.known() if tb.pack == T.PACK_WEREWOLF else (0, 0)
        calls = []
        for i in range(v.n):
            if not (tgt >> i) & 1 or i in acted or (self.human_mask >> i) & 1:
                continue
            d = rng.draw(tkey, i)
            if (d & 3) == 0:          # acts this turn with probability 3/4
                continue
            me = 1 << i
            if ph.act == T.ACT_WOLF_TARGET:
                cand, verb = alive & ~wolfteam, "voted to eliminate Player {}"
            elif ph.act == T.ACT_DOCTOR_PROTECT:
                cand, verb = alive, "chose to protect Player {}"
            elif ph.act == T.ACT_DETECTIVE:
                cand, verb = alive & ~me & ~(kv | kw), "investigated Player {}"
                if not cand:
                    cand = alive & ~me
            elif ph.act == T.ACT_DAY_VOTE:
                verb = "voted to eliminate Player {}"
                if (wolfteam >> i) & 1:
                    cand = alive & ~wolfteam
                elif v.role_class(i) == T.ROLE_DETECTIVE and (kw & alive):
                    lo = kw & alive
                    cand = lo & -lo            # lowest known living werewolf, no draw needed
                else:
                    cand = alive & ~me
            elif ph.act == T.ACT_TT_STATEMENTS:
                cand, verb = 0, None
            else:                              # ACT_TT_LIE / ACT_TT_VOTE
                cand, verb = 0, None
            if ph.act in (T.ACT_WOLF_TARGET, T.ACT_DOCTOR_PROTECT, T.ACT_DETECTIVE, T.ACT_DAY_VOTE):
                if not cand:
                    cand = alive
                k = bin(cand).count("1")
                choice = rng.nth_set_bit(cand, rng.pick(d, k)) + 1
                text = verb.format(choice)
            elif ph.act == T.ACT_TT_STATEMENTS:
                choice = 1
                text = ("shared three statements: " + ", ".join(
                    f"'Statement {s} of Player {i + 1}'" for s in (1, 2, 3)))
            elif ph.act == T.ACT_TT_LIE:
                choice = 1 + rng.pick(d, 3)
                text = f"chose statement {choice} as the lie"
            else:
                choice = 1 + rng.pick(d, 3)
                text = f"voted that statement {choice} is the lie"
            calls.append({"name": "update_player_actions",
                          "args": {"player_id": v.ids[i], "actions": f"[t={t}|c={choice}] {text}",
                                   "phase": ph.name}})
        return calls

    # ----------------------------------------------------------------- phase
    def phase_decision(self, state: dict, p_id: int) -> Tuple[bool, int, str]:
        """PhaseNode's LLM (v2:1075-1108): completion check, then first matching branch
        in DSL order (PhaseNode_system_prompt.txt:14-27, 44-56)."""
        tb = self.table
        ph = tb.by_id(p_id)
        if not ph.branches:
            return False, p_id, "terminal phase"
        v = self._view(state)
        if ph.completion == T.COMP_ACTION:
            tgt = v.targets(ph)
            acted = v.visit_actions(ph)
            am = 0
            for i in acted:
                am |= 1 << i
            if (tgt & ~am) != 0:
                return False, p_id, "waiting for target players"
        for b in ph.branches:
            if self._resolve(b.resolver, v):
                return True, b.target_id, b.key or "phase complete"
        return False, p_id, "no branch matched"

    def _resolve(self, res: int, v: RoomView) -> bool:
        tb = self.table
        if res in (T.RES_ALWAYS, T.RES_OTHERWISE):
            return True
        if res in (T.RES_WOLVES_ZERO, T.RES_WOLVES_GE_VILLAGERS):
            w = sum(1 for i in range(v.n) if v.alive(i) and v.get(i, "team") == "werewolves")
            g = sum(1 for i in range(v.n) if v.alive(i) and v.get(i, "team") == "villagers")
            return w == 0 if res == T.RES_WOLVES_ZERO else w >= g
        if res in (T.RES_FOLLOWS_DAY, T.RES_FOLLOWS_NIGHT):
            eff = tb.by_id(v.prev_phase).effect
            return eff == (T.EFF_DAY_RESOLVE if res == T.RES_FOLLOWS_DAY else T.EFF_NIGHT_RESOLVE)
        if res == T.RES_ALL_ROUNDS_DONE:
            return all(int(v.get(i, "rounds_as_speaker", 0)) >= tb.rounds for i in range(v.n))
        raise AssertionError(res)

    # --------------------------------------------------------------- referee
    def referee_updates(self, state: dict, p_id: int, q_id: int) -> List[dict]:
        """RefereeNode's LLM (v2:684-744; in v3 the same calls come from ActionExecutor's
        LLM, v3:599-607): (A) record this turn's actions, (B) entry effect of q."""
        tb, t = self.table, self.turn
        p, q = tb.by_id(p_id), tb.by_id(q_id)
        v = self._view(state)
        calls: List[dict] = []
        # working copy so that (B) sees (A)'s writes, as sequential tool application does
        ps = {pid: dict(v.ps[pid]) for pid in v.ids}

        def declared(name: str) -> Optional[str]:
            return tb.declared(name) if name in tb.names else name

        def put(i: int, name: str, value: Any):
            """`name`: canonical slot; written under the DSL's own name, skipped when the DSL does not declare it."""
            decl = declared(name)
            if decl is None:
                return
            ps[v.ids[i]][decl] = value
            calls.append({"name": "update_player_state",
                          "args": {"player_id": v.ids[i], "state_name": decl, "state_value": value}})

        def rd(i: int, name: str, default=None):
            decl = declared(name)
            return ps[v.ids[i]].get(decl, default) if decl else default

        def note(kind: str, text: str):
            calls.append({"name": "add_game_note", "args": {"note_type": kind, "content": text}})

        # (A) actions emitted this turn, in player-id order
        new = v.visit_actions(p, only_turn=t)
        for i in sorted(new):
            choice = new[i][1]
            if p.act in (T.ACT_WOLF_TARGET, T.ACT_DOCTOR_PROTECT, T.ACT_DETECTIVE):
                put(i, "night_action_submitted", True)
                put(i, "selected_target_id", choice)
                if p.act == T.ACT_DETECTIVE:
                    mem = dict(rd(i, "investigated_alignments") or {})
                    mem[str(choice)] = rd(choice - 1, "team", "")
                    put(i, "investigated_alignments", mem)
            elif p.act == T.ACT_TT_STATEMENTS:
                put(i, "statements", {str(s): f"Statement {s} of Player {i + 1}" for s in (1, 2, 3)})
                put(i, "statements_submitted", True)
            elif p.act == T.ACT_TT_LIE:
                put(i, "lie_index", choice)
            elif p.act == T.ACT_TT_VOTE:
                put(i, "vote_choice", choice)
                put(i, "has_voted", True)
        if q_id == p_id:
            return calls

        # (B) entering q
        note("PHASE_STATUS", f"[t={t}] phase {p_id} -> {q_id}")
        n = v.n
        alive = [bool(rd(i, "is_alive", True)) for i in range(n)]

        def kill(k: int, how: str):
            i = k - 1
            put(i, "is_alive", False)
            put(i, "can_vote", False)
            put(i, "night_action_eligible", False)
            put(i, "role_revealed", True)
            note("CRITICAL", f"Player {k} ({rd(i, 'role', '')}) eliminated {how} - marked is_alive=false")

        if q.effect == T.EFF_ASSIGN_ROLES:
            tkey = rng.deal_key(self.rkey, self.game)       # roles are dealt per game, not per turn
            rem = (1 << n) - 1
            cls = [T.ROLE_VILLAGER] * n
            order = [T.ROLE_WEREWOLF] * T.wolves_for(n) + [T.ROLE_DOCTOR, T.ROLE_DETECTIVE]
            for j, c in enumerate(order):
                k = bin(rem).count("1")
                if k == 0:
                    break
                i = rng.nth_set_bit(rem, rng.pick(rng.draw(tkey, 16 + j), k))
                cls[i] = c
                rem &= ~(1 << i)
            for i in range(n):
                special = cls[i] != T.ROLE_VILLAGER
                put(i, "role", tb.role_names[cls[i]])
                put(i, "team", "werewolves" if cls[i] == T.ROLE_WEREWOLF else "villagers")
                put(i, "has_secret_role", special)
                put(i, "night_action_eligible", special)
                put(i, "wolf_chat_enabled", cls[i] == T.ROLE_WEREWOLF)
            note("NEXT_PHASE", "Roles assigned: " + ", ".join(
                f"Player{i + 1}={tb.role_names[cls[i]]}" for i in range(n)))
        elif q.effect == T.EFF_NIGHT_BEGIN:
            for i in range(n):
                if rd(i, "night_action_submitted"):
                    put(i, "night_action_submitted", False)
                if rd(i, "selected_target_id"):
                    put(i, "selected_target_id", 0)
        elif q.effect == T.EFF_NIGHT_RESOLVE:
            roles = [tb.role_names.index(rd(i, "role", "")) if rd(i, "role", "") in tb.role_names else 0 for i in range(n)]
            if declared("selected_target_id"):
                target = [int(rd(i, "selected_target_id") or 0) for i in range(n)]
            else:
                # the DSL keeps no per-player target: this night's choices are the players' latest night actions in the log
                # (a living wolf / doctor has acted this night, or the night phases would not have completed)
                latest: Dict[int, Tuple[int, int]] = {}
                for ph in tb.phases:
                    if ph.act in (T.ACT_WOLF_TARGET, T.ACT_DOCTOR_PROTECT):
                        for i, (turn, choice) in v.visit_actions(ph, ever=True).items():
                            if i not in latest or turn > latest[i][0]:
                                latest[i] = (turn, choice)
                target = [latest[i][1] if i in latest else 0 for i in range(n)]
            votes = [target[i] for i in range(n) if alive[i] and roles[i] == T.ROLE_WEREWOLF]
            victim = plurality(votes, n)
            protect = 0
            for i in range(n):
                if alive[i] and roles[i] == T.ROLE_DOCTOR:
                    protect = target[i]
            if victim and victim != protect:
                kill(victim, "overnight by the werewolves")
            else:
                note("DECISION", f"Werewolves targeted Player {victim}, Doctor protected Player {protect} - no elimination")
        elif q.effect == T.EFF_DAY_RESOLVE:
            ballots = v.visit_actions(p)
            votes = [ballots[i][1] for i in sorted(ballots) if alive[i]]
            victim = plurality(votes, n)
            if victim:
                kill(victim, "by day vote")
        elif q.effect == T.EFF_TT_ROUND_START:
            speaker = -1
            for i in range(n):
                if int(ps[v.ids[i]].get("rounds_as_speaker", 0)) < tb.rounds:
                    speaker = i
                    break
            for i in range(n):
                put(i, "is_speaker", i == speaker)
                put(i, "can_vote", i != speaker)
                put(i, "statements", {})
                put(i, "statements_submitted", False)
                put(i, "lie_index", 0)
                put(i, "lie_revealed", False)
                put(i, "vote_choice", 0)
                put(i, "has_voted", False)
            note("DECISION", f"Selected Player {speaker + 1} as next speaker (turn_order)")
        elif q.effect == T.EFF_TT_REVEAL:
            for i in range(n):
                if ps[v.ids[i]].get("is_speaker"):
                    put(i, "lie_revealed", True)
        elif q.effect == T.EFF_TT_SCORE:
            sp = [i for i in range(n) if ps[v.ids[i]].get("is_speaker")]
            if sp:
                s = sp[0]
                lie = int(ps[v.ids[s]].get("lie_index") or 0)
                fooled = 0
                for i in range(n):
                    if i == s or not ps[v.ids[i]].get("has_voted"):
                        continue
                    if int(ps[v.ids[i]].get("vote_choice") or 0) == lie:
                        put(i, "total_score", int(ps[v.ids[i]].get("total_score", 0)) + 1)
                    else:
                        fooled += 1
                put(s, "total_score", int(ps[v.ids[s]].get("total_score", 0)) + fooled)
                put(s, "rounds_as_speaker", int(ps[v.ids[s]].get("rounds_as_speaker", 0)) + 1)
                note("SCORE_UPDATE", "Total scores - " + ", ".join(
                    f"Player {i + 1}: {ps[v.ids[i]].get('total_score', 0)}" for i in range(n)))
        return calls

    # -------------------------------------------------------------- executor
    def ui_calls(self, state: dict, q_id: int) -> List[dict]:
        """ActionExecutor / UIUpdateNode's LLM: the phase's DSL tool list, verbatim (ww:171-184 etc.).  Frontend only; no
        game state depends on it - but a person answers a createVotingPanel by its votingId and one of its options
        (src/app/page.tsx:302-305), so those two arguments are scripted (POLICY.md 3b); the others stay with the product's
        ui_script (the reference leaves them to the LLM)."""
        ph = self.table.by_id(q_id)
        v = self._view(state)
        calls = []
        for tname in ph.tools:
            args: Dict[str, Any] = {}
            if tname == "createVotingPanel":
                args = {"votingId": f"vote-p{q_id}-t{self.turn}", "options": v.panel_options(ph)}
            calls.append({"name": tname, "args": args})
        return calls
