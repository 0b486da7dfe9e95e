'use strict';
// node selftest_service.js <dsl.json>  — GPU: drives one room through RoomService over HTTP (tests/test_node_host.py)
const fs = require('fs');
const http = require('http');
const { RoomService, roomIndexOf } = require('./room_service.js');
const dsl = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));

function post(port, path, obj) {
  return new Promise((resolve, reject) => {
    const req = http.request({ host: '127.0.0.1', port, path, method: 'POST', headers: { 'content-type': 'application/json' } }, (res) => {
      let b = ''; res.on('data', (d) => { b += d; }); res.on('end', () => resolve(JSON.parse(b)));
    });
    req.on('error', reject); req.end(JSON.stringify(obj));
  });
}

(async () => {
  const svc = new RoomService({ seed: 7n });
  const server = await svc.serve(0);
  const port = server.address().port;
  const players = Array.from({ length: 8 }, (_, i) => ({ name: `Bot ${i + 1}`, gamePlayerId: i + 1 }));
  const first = await post(port, '/rooms', { threadId: 'room-abc', gameName: 'werewolf-(mafia)', players, dsl });
  const phases = [first.current_phase_id];
  let last = null, notes = 0, acts = 0, ui = 0;
  for (let t = 0; t < 70; t++) {
    last = await post(port, '/continue', { threadId: 'room-abc' });
    phases.push(last.state.current_phase_id);
    ui += last.uiCalls.length;
  }
  notes = last.state.game_notes.length;
  acts = Object.values(last.state.playerActions).reduce((a, r) => a + Object.keys(r.actions).length, 0);
  // overlapping requests on ONE thread (two /continue and an /action in flight together): they must be
  // served one after the other — no GE_BUSY, no duplicated log entries — and /close frees the room
  const seats = players.map((p, i) => Object.assign({}, p, i === 0 ? { isBot: false } : {}));
  await post(port, '/rooms', { threadId: 'room-h', gameName: 'werewolf-(mafia)', players: seats, dsl });
  for (let t = 0; t < 6; t++) await post(port, '/continue', { threadId: 'room-h' });
  const burst = await Promise.all([post(port, '/continue', { threadId: 'room-h' }), post(port, '/action', { threadId: 'room-h', playerId: 1, choice: 2 }),
                                   post(port, '/continue', { threadId: 'room-h' }), post(port, '/continue', { threadId: 'room-h' })]);
  const hist = burst[3].state.phase_history.length;
  const ids = Object.values(burst[3].state.playerActions).map((r) => Object.keys(r.actions));
  const idsOk = ids.every((k) => k.every((id, j) => id === String(j + 1)));
  const busy = burst.some((r) => r && r.error && /busy/i.test(r.error));
  // the message-level entry over HTTP: control plays a turn, chat does not, a game message is logged under Player 1 / phase 0's name
  const m0 = await post(port, '/message', { threadId: 'room-h', text: 'Continue' });
  const m1 = await post(port, '/message', { threadId: 'room-h', text: 'Player Bot 1 in game chat: hello' });
  const m2 = await post(port, '/message', { threadId: 'room-h', text: 'Button "Skip" (ID: b1) has been clicked. Action: skip' });
  const logged = Object.values(m2.state.playerActions['1'].actions).filter((a) => a.action.startsWith('Button "Skip"'));
  const message = { kinds: [m0.kind, m1.kind, m2.kind], played: [m0.played, m1.played, m2.played],
                    hist: [m0.state.phase_history.length, m1.state.phase_history.length, m2.state.phase_history.length],
                    loggedPhase: logged.length === 1 ? logged[0].phase : null };
  const closed = await post(port, '/close', { threadId: 'room-h' });
  const after = await post(port, '/continue', { threadId: 'room-h' });
  server.close();
  console.log(JSON.stringify({ room: roomIndexOf('room-abc').toString(), phases, notes, acts, ui,
                               burst: { hist, idsOk, busy, closed: closed.closed, afterClose: after.error || null, rooms: svc.rooms.size }, message,
                               name1: last.state.player_states['1'].name, finalPhase: last.state.current_phase_name,
                               alive: Object.values(last.state.player_states).map((p) => p.is_alive ? 1 : 0) }));
})().catch((e) => { console.error(e); process.exit(1); });
