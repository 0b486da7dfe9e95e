'use strict';
// node selftest_messages.js <dsl.json> <strings_human_golden.json> - GPU: RoomService.handleMessage must leave, after every
// message of the scripted person, the AgentState the reference run left (tests/golden/strings_human_*.json; tests/test_messages.py)
const fs = require('fs');
const { RoomService } = require('./room_service.js');
const dsl = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const gold = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));

const strip = (x) => {
  if (Array.isArray(x)) return x.map(strip);
  if (x && typeof x === 'object') { const o = {}; for (const [k, v] of Object.entries(x)) if (k !== 'timestamp') o[k] = strip(v); return o; }
  return x;
};
const same = (a, b) => JSON.stringify(a) === JSON.stringify(b);      // key order matters: the reference's dict order

(async () => {
  let messages = 0;
  for (const c of gold.cases) {
    const svc = new RoomService({ seed: BigInt(c.seed) });
    const players = c.names.map((name, i) => ({ name, gamePlayerId: i + 1, isBot: !c.human_seats.includes(i + 1) }));
    svc.createRoom({ threadId: 't', gameName: gold.game, players, dsl, roomIndex: c.room });
    let nNotes = 0, nHist = 0, nActs = 0, state = null;
    for (let k = 0; k < c.messages.length; k++) {
      const want = c.messages[k];
      const out = await svc.handleMessage('t', want.message);
      state = out.state;
      const where = `seed ${c.seed} room ${c.room} message ${k} ${JSON.stringify(want.message.slice(0, 60))}`;
      if (out.played !== want.played) throw new Error(`played, ${where}`);
      if (!want.played && (out.toolCalls.length || out.uiCalls.length)) throw new Error(`chat produced calls, ${where}`);
      if (state.current_phase_id !== want.current_phase_id || state.current_phase_name !== want.current_phase_name) throw new Error(`phase, ${where}`);
      const acts = [];
      for (const pid of Object.keys(state.playerActions).sort((a, b) => a - b)) {
        const rec = state.playerActions[pid];
        for (const id of Object.keys(rec.actions).sort((a, b) => a - b)) acts.push({ player_id: pid, name: rec.name, id: rec.actions[id].id, action: rec.actions[id].action, phase: rec.actions[id].phase });
      }
      if (acts.length !== nActs + want.actions_added.length || !want.actions_added.every((a) => acts.some((x) => same(x, a)))) throw new Error(`playerActions, ${where}`);
      if (!same(state.game_notes.slice(nNotes), want.notes_added)) throw new Error(`game_notes, ${where}: ${JSON.stringify(state.game_notes.slice(nNotes))}`);
      if (!same(strip(state.phase_history.slice(nHist)), want.history_added)) throw new Error(`phase_history, ${where}`);
      if (!same(strip(state.player_states), want.player_states)) throw new Error(`player_states, ${where}: ${JSON.stringify(state.player_states['1'])}`);
      nActs = acts.length; nNotes = state.game_notes.length; nHist = state.phase_history.length;
      messages++;
    }
    if (!same(strip(state.playerActions), c.final.playerActions) || !same(state.game_notes, c.final.game_notes)) throw new Error(`final log, seed ${c.seed} room ${c.room}`);
    await svc.close('t');
  }
  console.log(JSON.stringify({ ok: true, messages }));
})().catch((e) => { console.error(e); process.exit(1); });
