"""ORACLE (test infrastructure) — a scripted "human" for player 1, to pin host-driven players.

In the reference player 1 is the human (bot_behavior_system_prompt.txt:3): the bot policy never
acts for it, its action arrives as a chat message and is logged at the start of the next graph run
(agent/tools/utils.py:310-358).  The golden generator and the parity tests need a deterministic
stand-in for that person; this is it.  Rule: if player 1 is a pending target of the current phase
and the turn index is a multiple of 3, it acts — Werewolf: on the lowest living player id other
than itself; Two-Truths: statements -> 1, lie -> 2, vote -> 3.
"""
from . import dsl_table as T

HUMAN_MASK = 1          # player 1


def _base_true(pack, f, base):
    if pack == T.PACK_WEREWOLF:
        return {0: f[2], 1: f[4], 2: f[3], 3: f[5], 4: f[6], 5: f[7]}.get(base, None) if base < 6 else (
            f[1] == 1 if base == 6 else f[1] == 2 if base == 7 else f[0] == base - 7)
    return {0: f[0], 1: f[1], 2: f[3], 3: f[4], 4: f[6]}[base]


def scripted_human(table, turn, projection, n, player=1):
    """(player_id, choice) or None, from the canonical projection before the turn."""
    if turn % 3 != 0:
        return None
    ph = table.by_id(projection[0])
    if ph.completion != T.COMP_ACTION:
        return None
    fields = [projection[4 + 11 * i: 15 + 11 * i] for i in range(n)]
    f = fields[player - 1]
    if table.pack == T.PACK_WEREWOLF and not f[2]:
        return None
    for term in ph.terms:
        if bool(_base_true(table.pack, f, term.base)) == term.negate:
            return None
    if f[9]:
        return None
    if table.pack == T.PACK_WEREWOLF:
        cands = [i + 1 for i in range(n) if fields[i][2] and i + 1 != player]
        return (player, cands[0]) if cands else None
    return player, {T.ACT_TT_STATEMENTS: 1, T.ACT_TT_LIE: 2, T.ACT_TT_VOTE: 3}[ph.act]
