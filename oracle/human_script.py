"""ORACLE (test infrastructure) — a scripted "human" for player 1, to pin host-driven players.

In the reference player 1 is the human (bot_behavior_system_prompt.txt:3): the bot policy never
acts for it, its action arrives as a message of the browser and is logged at the start of the next
graph run (agent/tools/utils.py:310-358).  The golden generator and the parity tests need a
deterministic stand-in for that person; this is it.

* `scripted_human` - WHAT the person decides: if player 1 is a pending target of the current phase and the
  turn index is a multiple of 3, it acts — Werewolf: on the lowest living player id other than itself;
  Two-Truths: statements -> 1, lie -> 2, vote -> 3.
* `person(session)` - the same decisions as the MESSAGE the browser sends for them (the harness's
  RoomSession.vote_message: `Player 1 voted "<option>" in voting <votingId>` / `Input: <text>`), which the
  reference itself logs; tests/golden/human_*.json are generated with it.
* `talkative_person(session)` - the decisions above plus everything else a browser can send: "Start game.",
  chat (no turn is played), a button click, votes on a stale panel / for an option the panel does not offer /
  in a phase that is not the person's, a message over 200 characters, a control-like text, and a second
  host-driven seat; tests/golden/strings_human_*.json record the reference's AgentState after each one.
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


def person(session, seats=(1,)):
    """The message that starts the next graph run of `session` (oracle/refharness/walker.py RoomSession), None = "Continue"."""
    proj = session.project()
    for seat in seats:
        act = scripted_human(session.table, session.turn, proj, session.n_players, player=seat)
        if act:
            return session.vote_message(*act)
    return None


def talkative_person(seats=(1,)):
    """A person who also sends what is NOT a valid action.  Returns script(session) with its own message counter."""
    sent = {"n": 0}

    def script(session):
        k = sent["n"]
        sent["n"] += 1
        t = session.turn
        if k == 0:
            return "Start game."                                              # src/app/page.tsx:2774
        if k % 11 == 4:
            return "Player Alice in game chat: who do you all suspect?"         # page.tsx:345-348: ChatBotNode, no turn
        if k % 17 == 9:
            return "Player Alice to Bot 3: are you the Doctor?"                 # page.tsx:341-344
        if k % 13 == 6:
            return 'Button "Skip" (ID: btn-7) has been clicked. Action: skip'   # page.tsx:272-275: logged, no game effect
        if k % 19 == 12:
            return "I would like to talk to bots about " + "the weather " * 20  # contains "to bot": control-like, not logged
        if k % 23 == 15:
            return "Input: " + "blah " * 60                                     # > 200 characters: the log keeps 200
        msg = person(session, seats)
        if msg and msg.startswith("Player ") and k % 5 == 3 and session.last_panel:
            vid, options = session.last_panel
            p, tt = vid[len("vote-p"):].split("-t")
            stale = f"vote-p{p}-t{int(tt) - 1}"
            return msg.replace(vid, stale)                                      # an older panel's id: logged, no effect
        if msg and msg.startswith("Player ") and k % 7 == 2:
            head, rest = msg.split(' voted "', 1)
            return head + ' voted "Nobody" ' + rest[rest.index('" in voting') + 2:]   # an option the panel does not offer
        if msg is None and session.last_panel and k % 3 == 1:
            vid, options = session.last_panel                                   # a vote when it is not this person's turn to act
            return f'Player {seats[0]} voted "{options[0]}" in voting {vid}' if options else None
        return msg

    return script
