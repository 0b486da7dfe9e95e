'use strict';
/**
 * The browser's messages, as the reference's graph reads them (POLICY.md 3b) - twin of game_engine_amd/messages.py.
 *
 * Every graph run of the reference is started by ONE message of the frontend (src/app/page.tsx:183-259):
 *   "Start game." :2774, "Continue" :2962                                                  control
 *   `Player ${playerId} voted "${option}" in voting ${votingId}`  :302-305                  a vote
 *   `Button "${name}" (ID: ${id}) has been clicked. Action: ${action}`  :272-275            a button
 *   `Input: ${text}`  :2843                                                                 a text panel
 *   `Player ${name} in game chat: ${m}` / `Player ${name} to Bot ${id}: ${m}`  :341-349     chat
 * InitialRouterNode (agent/game_agent_v2.py:198-349) sends chat to ChatBotNode - no turn - and hands everything else to
 * process_human_action_if_needed (agent/tools/utils.py:310-358): logged verbatim (200 characters) under Player 1 unless it is
 * a control message, always under phase 0's name (it reads `currentPhaseId` / `playerStates`, keys the state does not have,
 * v2:324-331); then the turn is played.
 */
const CHAT = 'chat', CONTROL = 'control', ACTION = 'action';
const VOTE = /^Player (\d+) voted "([\s\S]*)" in voting (\S+)$/;
const ACT_TT_STATEMENTS = 5, PACK_WEREWOLF = 1;

/** chat: v2:305-311 (case-sensitive `to Bot`); control: utils.py:334-339 (lower-cased); else a logged action. */
function classify(text) {
  if (text.includes('in game chat:') || text.includes('to Bot')) return CHAT;
  const low = text.toLowerCase().trim();
  if (low.includes('in game chat:') || low.includes('to bot') || ['continue', 'start game', 'start game.'].includes(low)) return CONTROL;
  return ACTION;
}
/** utils.py:346 `str(content)[:200]` - Python slices code points, so do we. */
function loggedText(text) { return Array.from(String(text)).slice(0, 200).join(''); }
/** { votingId, options } of the createVotingPanel among a turn's frontend calls - what a person can answer next. */
function newestPanel(uiCalls) {
  const c = uiCalls.find((x) => x.name === 'createVotingPanel');
  return c ? { votingId: String(c.args.votingId), options: c.args.options.map(String) } : null;
}
/**
 * Candidate [seat, choice] readings of a logged message, in the order to try them (the stepper refuses a seat that is not a
 * living pending target: ge_batch_inject_action); [] when the message is no game action.  A vote counts when it names the
 * NEWEST panel's votingId, a host-driven seat and a valid choice (a living player's name - lowest id carrying it - or a
 * statement number 1..3); `Input: ...` answers the statements phase for the host-driven seats in id order.
 */
function resolve(text, panel, act, pack, names, alive, humanSeats) {
  const m = VOTE.exec(text);
  if (m) {
    const seat = Number(m[1]), option = m[2], votingId = m[3];
    if (!panel || votingId !== panel.votingId || !humanSeats.includes(seat) || act === ACT_TT_STATEMENTS) return [];
    let choice;
    if (pack === PACK_WEREWOLF) {
      choice = names.indexOf(option) + 1;
      if (!choice || !alive[choice - 1]) return [];
    } else {
      choice = ['1', '2', '3'].includes(option) ? Number(option) : 0;
      if (!choice) return [];
    }
    return [[seat, choice]];
  }
  if (text.startsWith('Input: ') && act === ACT_TT_STATEMENTS) return humanSeats.slice().sort((a, b) => a - b).map((s) => [s, 1]);
  return [];
}
module.exports = { CHAT, CONTROL, ACTION, classify, loggedText, newestPanel, resolve };
