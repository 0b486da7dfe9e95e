'use strict';
// node selftest.js <dsl.json> [golden.json]   — used by tests/test_node_host.py
const fs = require('fs');
const { GameTable, RoomBatch, deviceCount, turnToolCalls } = require('./index.js');
const dsl = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const table = new GameTable(dsl);
const out = { phases: table.info.phases.length, pack: table.info.pack, devices: deviceCount() };
{ // room setup helpers (no GPU): initialize-players twin and game-file matching
  const { initializePlayers, findGameFile } = require('./index.js');
  const players = [{ id: 'u-7', name: 'Ann', isHost: true, gamePlayerId: '1' }, { name: 'Bob' }, { id: 'u-9', name: 'Cy', gamePlayerId: '3' }];
  out.init = initializePlayers(dsl, players);
  const bare = JSON.parse(JSON.stringify(dsl)); delete bare.declaration.player_states_template;
  out.initFromSchema = initializePlayers(bare, players.slice(0, 1));
  out.initFallback = initializePlayers({ declaration: {} }, players);
  out.roomIndex = require('./room_service.js').roomIndexOf('thread-42 ✓').toString();
  if (process.env.GE_TEST_GAMES_DIR) out.found = [findGameFile('Werewolf (Mafia)', process.env.GE_TEST_GAMES_DIR), findGameFile('chess', process.env.GE_TEST_GAMES_DIR)];
}
if (deviceCount() === 0) {
  try { new RoomBatch({ segments: [{ table, nPlayers: 8, nRooms: 4 }] }); out.noDevice = 'created?!'; } catch (e) { out.noDevice = e.code; }
  console.log(JSON.stringify(out));
  process.exit(0);
}
(async () => {
  const golden = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
  let checked = 0;
  for (const c of golden.cases) {
    const b = new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 1 }], seed: BigInt(c.seed), firstRoom: BigInt(c.room), maxFuse: 1 });
    for (let t = 0; t < c.turns.length; t++) {
      await b.step(1);
      const r = b.readRoom(0);
      const want = c.turns[t];
      if (r.current_phase_id !== want[0] || r.previous_phase_id !== want[1] || r.end_turn !== want[3]) throw new Error(`phase mismatch turn ${t}`);
      const n = golden.n_players;
      for (let i = 0; i < n; i++) {
        const ps = r.player_states[String(i + 1)], w = want.slice(4 + 11 * i, 15 + 11 * i);
        if (table.info.pack === 1) {
          // a slot the DSL does not declare (the draft has no selected_target_id) reads 0 in the golden and is absent from player_states
          const target = table.info.fieldNames[8] ? ps[table.info.fieldNames[8]] : 0;
          if (ps.is_alive !== !!w[2] || target !== w[8] || ps.role !== table.info.roleNames[w[0]]) throw new Error(`player ${i + 1} turn ${t}`);
          const elig = table.info.fieldNames[6];
          if (elig && ps[elig] !== !!w[6]) throw new Error(`player ${i + 1} turn ${t}: ${elig}`);
        } else if (ps.total_score !== w[7] || ps.vote_choice !== w[5] || ps.is_speaker !== !!w[0]) throw new Error(`player ${i + 1} turn ${t}`);
        if (r.acted[i] !== w[9] || r.choice[i] !== w[10]) throw new Error(`log ${i + 1} turn ${t}`);
      }
      checked++;
    }
  }
  // tool calls of a traced single room (the drop-in case: one LangGraph thread), for comparison with the Python host
  const one = new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 1 }], seed: 5n, firstRoom: 9n, maxFuse: 1, trace: true });
  out.calls = [];
  let before = one.readRoom(0);
  for (let t = 0; t < 30; t++) {
    one.stepSync(1);
    const after = one.readRoom(0);
    out.calls.push(turnToolCalls(table, before, after, one.readEvents(0, 1)[0][0]));
    before = after;
  }
  const big = new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 10000 }], seed: 7n });
  { // the handle is not thread-safe: a synchronous call while the async step owns it throws GE_BUSY
    const { addon } = require('./index.js');
    const p = addon.step(big.handle, 64);                      // the raw binding: RoomBatch.step() would queue instead
    try { addon.summary(big.handle); out.busy = 'not refused'; } catch (e) { out.busy = e.code; }
    await p;
    // through the host class, calls queue up behind the step instead of being refused
    const q = big.step(1);
    out.queued = Number((await big.whenIdle(() => big.summary())).turn);
    await q;
  }
  const s = big.summary();
  { // checkpoint (raw room views + turn) restored into a FRESH batch continues bit-identically; close() frees
    const mk = () => new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 500 }], seed: 3n, firstRoom: 40n, restart: true });
    const a = mk(), b = mk();
    await a.step(29);
    b.writeRoomsRaw(0, a.readRoomsRaw(0, 500));
    b.setTurn(29);
    await Promise.all([a.step(35), b.step(35)]);
    out.restoreEqual = Buffer.from(a.readRoomsRaw(0, 500)).equals(Buffer.from(b.readRoomsRaw(0, 500)));
    a.close(); b.close();
    try { a.readRoom(0); out.closed = 'still readable'; } catch (e) { out.closed = 'refused'; }
  }
  { // batched injection of host-driven players' actions: per-action status, same effect as one by one
    const mk = () => new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 64, humanMask: 1 }], seed: 13n, maxFuse: 1 });
    const a = mk(), b = mk();
    let applied = 0, same = true;
    for (let t = 0; t < 40; t++) {
      const rooms = Array.from({ length: 64 }, (_, r) => r), pl = rooms.map(() => 1), ch = rooms.map((r) => 1 + ((r + t) % 3));
      const st = a.injectActions(rooms, pl, ch);
      rooms.forEach((r, k) => {
        let ok = true;
        try { b.injectAction(r, 1, ch[k]); } catch (e) { ok = false; }
        if (ok !== (st[k] === 0)) same = false;
        applied += ok ? 1 : 0;
      });
      a.stepSync(1); b.stepSync(1);
    }
    out.injectBatch = { same, applied, equal: Buffer.from(a.readRoomsRaw(0, 64)).equals(Buffer.from(b.readRoomsRaw(0, 64))) };
  }
  { // three shards (all on device 0 here) == one batch of the same rooms: per room and in the summary
    const { ShardedBatch } = require('./index.js');
    const seg = [{ table, nPlayers: golden.n_players, nRooms: 3000 }];
    const sh = new ShardedBatch({ segments: seg, devices: [0, 0, 0], seed: 11n, firstRoom: 1000n, restart: true });
    const one = new RoomBatch({ segments: [{ table, nPlayers: golden.n_players, nRooms: 9000 }], seed: 11n, firstRoom: 1000n, restart: true });
    await Promise.all([sh.step(100), one.step(100)]);
    const a = sh.summary(), b = one.summary();
    out.shardSummaryEqual = ['rooms', 'finished', 'village_wins', 'wolf_wins', 'alive_players', 'sum_end_turn', 'checksum', 'turn', 'games_recycled']
      .every((f) => a[f] === b[f]) && a.end_turn_hist.every((x, i) => x === b.end_turn_hist[i]);
    out.shardRoomsEqual = [0, 2999, 3000, 4567, 8999].every((r) => JSON.stringify(sh.readRoom(r)) === JSON.stringify(one.readRoom(r)));
    out.shardRooms = Number(a.rooms);
  }
  { // the native device group (one process, RCCL all-gather inside the library) on the one device of this box == one batch,
    // per room and in the summary; duplicate devices are refused before RCCL is asked
    const { DeviceGroup } = require('./index.js');
    const segs = [{ table, nPlayers: golden.n_players, nRooms: 7001 }];
    const grp = new DeviceGroup({ segments: segs, devices: [0], seed: 11n, firstRoom: 1000n, restart: true });
    const one = new RoomBatch({ segments: segs, seed: 11n, firstRoom: 1000n, restart: true });
    await Promise.all([grp.step(60), one.step(60)]);
    await grp.step(40); await one.step(40);
    const a = grp.summary(), b = one.summary();
    out.groupSummaryEqual = ['rooms', 'finished', 'village_wins', 'wolf_wins', 'alive_players', 'sum_end_turn', 'checksum', 'turn', 'games_recycled']
      .every((f) => a[f] === b[f]) && a.end_turn_hist.every((x, i) => x === b.end_turn_hist[i]);
    out.groupRoomsEqual = [0, 1, 3500, 7000].every((r) => JSON.stringify(grp.readRoom(r)) === JSON.stringify(one.readRoom(r)));
    try { new DeviceGroup({ segments: segs, devices: [0, 0] }); out.groupDuplicate = 'accepted'; } catch (e) { out.groupDuplicate = e.code; }
    grp.close(); one.close();
  }
  out.checked = checked; out.finished = Number(s.finished); out.turn = Number(s.turn);
  out.sample = big.readRoom(123).current_phase_name;
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
