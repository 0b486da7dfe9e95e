"""The browser's messages, as the reference's graph reads them (POLICY.md 3b) — pure host logic, no device code.

Every graph run of the reference is started by ONE message of the frontend (src/app/page.tsx, all through
handleUserInteraction :183-259):
    "Start game."  :2774        "Continue"  :2962                                        control
    `Player ${playerId} voted "${option}" in voting ${votingId}`   :302-305               a vote
    `Button "${item.name}" (ID: ${item.id}) has been clicked. Action: ${action}`  :272-275  a button
    `Input: ${text}`   :2843                                                               a text panel
    `Player ${name} in game chat: ${message}` / `Player ${name} to Bot ${id}: ${message}`  :341-349   chat
InitialRouterNode (agent/game_agent_v2.py:198-349) sends a chat message to ChatBotNode — no turn is played — and hands
every other one to process_human_action_if_needed (agent/tools/utils.py:310-358), which logs it VERBATIM (first 200
characters) as an action of Player 1 unless it is a control message, then plays the turn.  Because InitialRouterNode reads
`currentPhaseId` / `playerStates`, keys the state does not have (v2:324-331), the entry always carries phase 0's name.
`classify` and `logged_text` mirror those tests character by character; `resolve` is the fixed policy's reading of a
logged message (what the reference leaves to its Referee LLM): which seat acted and what it chose.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

CHAT, CONTROL, ACTION = "chat", "control", "action"
_VOTE = re.compile(r'^Player ([0-9]+) voted "(.*)" in voting (\S+)\Z', re.S)
_INPUT = re.compile(r"^Input: ", re.S)
ACT_TT_STATEMENTS = 5                                   # include/ge_step.h GE_ACT_TT_STATEMENTS
PACK_WEREWOLF = 1


def classify(text: str) -> str:
    """chat: routed to ChatBotNode, no turn (v2:305-311 — case-sensitive `to Bot`);
    control: a turn, nothing logged (utils.py:334-339 — lower-cased, so `to bot` anywhere also lands here);
    action: logged under Player 1, then a turn."""
    if "in game chat:" in text or "to Bot" in text:
        return CHAT
    low = text.lower().strip()
    if "in game chat:" in low or "to bot" in low or low in ("continue", "start game", "start game."):
        return CONTROL
    return ACTION


def logged_text(text: str) -> str:
    """What the log keeps of an action message (utils.py:346: `str(content)[:200]`)."""
    return str(text)[:200]


def newest_panel(ui_calls: Sequence[Dict[str, Any]]) -> Optional[Tuple[str, List[str]]]:
    """(votingId, options) of the createVotingPanel among a turn's frontend calls — what a person can answer next."""
    for c in ui_calls:
        if c["name"] == "createVotingPanel":
            return str(c["args"]["votingId"]), [str(o) for o in c["args"]["options"]]
    return None


def resolve(text: str, panel: Optional[Tuple[str, List[str]]], act: int, pack: int, names: Sequence[str], alive: Sequence[bool],
            human_seats: Sequence[int]) -> List[Tuple[int, int]]:
    """Candidate (seat, choice) readings of a logged message, in the order to try them (the stepper refuses a seat that is
    not a living pending target of the phase: ge_batch_inject_action) — [] when the message is no game action.

    * a vote counts when it names the votingId of the NEWEST panel (the one the previous turn's createVotingPanel call
      carried: a person can only click what is on the canvas), a host-driven seat, and an option that stands for a valid
      choice: a living player's name (the lowest id carrying it) or a statement number 1..3;
    * `Input: ...` answers the statements phase's text panel, which names no seat: the host-driven seats in id order;
    * a button click, free text, a vote on an older panel: logged, no game effect."""
    m = _VOTE.match(text)
    if m:
        seat, option, voting_id = int(m.group(1)), m.group(2), m.group(3)
        if panel is None or voting_id != panel[0] or seat not in human_seats or act == ACT_TT_STATEMENTS:
            return []
        if pack == PACK_WEREWOLF:
            choice = next((i + 1 for i, nm in enumerate(names) if nm == option), 0)
            if not choice or not alive[choice - 1]:
                return []
        else:
            choice = int(option) if option in ("1", "2", "3") else 0
            if not choice:
                return []
        return [(seat, choice)]
    if _INPUT.match(text) and act == ACT_TT_STATEMENTS:
        return [(seat, 1) for seat in sorted(human_seats)]
    return []
