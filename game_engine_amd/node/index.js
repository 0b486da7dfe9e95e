'use strict';
/**
 * TypeScript/Node host of the MI355X batch stepper (types: index.d.ts).
 *
 * Drop-in point in the reference: src/app/api/copilotkit/route.ts:22-47 hands a room's
 * AgentState to the Python LangGraph server (graph "sample_agent" = agent/game_agent_v2.py).
 * This module offers the same step — "advance this room by one turn" — for whole batches of
 * rooms, on the GPU, and returns state in the AgentState shape the frontend already syncs
 * (src/lib/canvas/types.ts:338-360): current_phase_id, current_phase_name,
 * player_states {"1": {<declared fields>}}.  YAML is parsed here with js-yaml, exactly as the
 * reference's TS side does (src/app/api/games/initialize-players/route.ts); everything else is
 * the C ABI behind ge_addon.node.
 */
const fs = require('fs');
const path = require('path');
const addon = require('./ge_addon.node');

function loadYaml() {
  try { return require('js-yaml'); } catch (e) { /* fall through */ }
  try { return require('/usr/share/nodejs/js-yaml'); } catch (e) { /* fall through */ }
  throw new Error('js-yaml is required to read games/*.yaml (the reference ships it: package.json)');
}

/** Same contract as the reference's load_dsl_by_gamename (agent/tools/utils.py:557-581). */
function loadDslByGamename(gamename, gamesDir) {
  if (!gamename) return {};
  const dir = gamesDir || process.env.GE_GAMES_DIR || 'games';
  const y = path.join(dir, `${gamename}.yaml`);
  const j = path.join(dir, `${gamename}.json`);
  if (fs.existsSync(y)) return loadYaml().load(fs.readFileSync(y, 'utf8')) || {};
  if (fs.existsSync(j)) return JSON.parse(fs.readFileSync(j, 'utf8'));
  return {};
}

/** Game name -> file in the games directory, as the reference's TS routes match it
 * (src/app/api/games/initialize-players/route.ts:62-70): case-insensitive, every character outside
 * [a-z0-9] equals '-'.  Returns the file name or null. */
function findGameFile(gameName, gamesDir) {
  const dir = gamesDir || process.env.GE_GAMES_DIR || 'games';
  const norm = (x) => String(x).toLowerCase().replace(/[^a-z0-9]/g, '-');
  const want = norm(gameName);
  for (const f of fs.readdirSync(dir)) {
    if (norm(f.toLowerCase().replace('.yaml', '')) === want) return f;
  }
  return null;
}

/** player_states for the real players of a room, as POST /api/games/initialize-players builds them
 * (route.ts:83-166): template = declaration.player_states_template.player_states['1'], else its first
 * entry, else defaults generated from the declaration.player_states schema (string '' / num 0 /
 * boolean true / list [] / dict {}, or the field's `example`), else {player_states: {}, fallback_mode: true}.
 * Every player gets a copy of the template plus name, id, isHost; keys are gamePlayerId or the
 * 1-based position.  (The Python host's initialize_player_states_from_dsl is the agent-side twin,
 * agent/tools/utils.py:584-653.) */
function initializePlayers(dsl, roomPlayers) {
  const decl = (dsl && dsl.declaration) || {};
  let template;
  const tpl = decl.player_states_template && decl.player_states_template.player_states;
  if (tpl) {
    template = tpl['1'];
    if (!template) {
      const ids = Object.keys(tpl);
      if (ids.length) template = tpl[ids[0]];
    }
  }
  if (!template && decl.player_states) {
    template = {};
    for (const [field, def] of Object.entries(decl.player_states)) {
      const type = (def && def.type) || 'string';
      const ex = def ? def.example : undefined;
      if (type === 'string') template[field] = ex || '';
      else if (type === 'num' || type === 'number') template[field] = ex || 0;
      else if (type === 'boolean') template[field] = ex !== undefined ? ex : true;
      else if (type === 'array' || type === 'list') template[field] = ex || [];
      else if (type === 'object' || type === 'dict') template[field] = ex || {};
      else template[field] = ex || null;
    }
  }
  if (!template || !Object.keys(template).length) {
    return { player_states: {}, fallback_mode: true, message: 'No template found, agent will generate player_states' };
  }
  const out = {};
  roomPlayers.forEach((pl, k) => {
    const pid = pl.gamePlayerId || String(k + 1);
    out[pid] = Object.assign(JSON.parse(JSON.stringify(template)), { name: pl.name, id: pl.id || pid, isHost: pl.isHost || false });
  });
  return { player_states: out };
}

const TEAMS = ['', 'villagers', 'werewolves'];
const VIEW = { size: addon.roomViewSize(), players: 20, det: 20 + 16 * 12 };

class GameTable {
  constructor(dsl, rounds = 1) {
    if (!dsl || typeof dsl !== 'object' || !Object.keys(dsl).length) throw new Error('empty DSL');
    this.dsl = dsl;
    this.handle = addon.compileTable(JSON.stringify(dsl), rounds);
    this.info = addon.tableInfo(this.handle);
    // declared fields the rule packs do not model: constants from the template (nobody writes them under the fixed policy)
    const tps = ((dsl.declaration || {}).player_states_template || {}).player_states || {};
    const tmpl = tps['1'] || tps[Object.keys(tps)[0]] || {};
    // info.fieldNames: slot (include/ge_step.h GE_WW_* / GE_TT_*) -> the DSL's own field name, '' = not declared
    const modelled = new Set(['name', ...this.info.fieldNames.filter((x) => x)]);
    this.extraFields = {};
    for (const [k, v] of Object.entries(tmpl)) if (!modelled.has(k) && ['boolean', 'number', 'string'].includes(typeof v)) this.extraFields[k] = v;
  }
  static fromGamename(gamename, gamesDir, rounds = 1) {
    return new GameTable(loadDslByGamename(gamename, gamesDir), rounds);
  }
  phaseName(id) {
    const p = this.info.phases.find((x) => x.id === id);
    return p ? p.name : `Phase ${id}`;            // utils.py:30 fallback
  }
}

function decodeRoom(table, buf, off) {
  const dv = new DataView(buf, off, VIEW.size);
  const u8 = new Uint8Array(buf, off, VIEW.size);
  const n = u8[17];
  const pack = u8[18];
  const playerStates = {};
  const det = Array.from(u8.slice(VIEW.det, VIEW.det + n));
  const names = table.info.fieldNames;
  const slots = [];              // per player: one value per slot of the pack, declared by the DSL or not
  for (let i = 0; i < n; i++) {
    const f = u8.slice(VIEW.players + 12 * i, VIEW.players + 12 * i + 12);
    let vals;
    if (pack === 1) {
      const mem = {};
      if (f[0] === 4) det.forEach((d, k) => { if (d) mem[String(k + 1)] = TEAMS[d]; });
      vals = [table.info.roleNames[f[0]], TEAMS[f[1]], !!f[2], !!f[3], !!f[4], !!f[5], !!f[6], !!f[7], f[8], mem, f[1] === 2];
    } else {
      vals = [!!f[0], !!f[1], f[2], !!f[3], !!f[4], f[5], !!f[6], f[7], f[8]];
    }
    slots.push(vals);
    const rec = {};                // player_states hold exactly what the DSL declares, under its own names
    vals.forEach((v, s) => { if (names[s]) rec[names[s]] = v; });
    playerStates[String(i + 1)] = Object.assign(rec, table.extraFields || {});
  }
  const phaseId = dv.getInt32(0, true);
  return {
    current_phase_id: phaseId, current_phase_name: table.phaseName(phaseId),
    previous_phase_id: dv.getInt32(4, true), end_turn: dv.getInt32(8, true), games: dv.getInt32(12, true),
    player_states: playerStates, pack, slots,
    acted: Array.from({ length: n }, (_, i) => u8[VIEW.players + 12 * i + 9]),
    choice: Array.from({ length: n }, (_, i) => u8[VIEW.players + 12 * i + 10]),
  };
}

const EVENT_SIZE = 32;

function actionText(act, player, c) {
  if (act === 1 || act === 4) return `voted to eliminate Player ${c}`;
  if (act === 2) return `chose to protect Player ${c}`;
  if (act === 3) return `investigated Player ${c}`;
  if (act === 5) return 'shared three statements: ' + [1, 2, 3].map((s) => `'Statement ${s} of Player ${player}'`).join(', ');
  if (act === 6) return `chose statement ${c} as the lie`;
  return `voted that statement ${c} is the lie`;
}

// add_game_note's categories and their marks (agent/tools/backend_tools.py:175-187; unknown type -> the EVENT mark)
const NOTE_EMOJI = { CRITICAL: '\u{1F534}', VOTING_STATUS: '⚠️', DECISION: '\u{1F3AF}', BOT_REMINDER: '\u{1F916}', UI_FILTER: '\u{1F6AB}',
                     PHASE_STATUS: '⏳', NEXT_PHASE: '\u{1F52E}', GAME_STATUS: '\u{1F3C6}', PHASE_SUGGESTION: '\u{1F4A1}',
                     BRANCH_RECOMMENDATION: '\u{1F500}', EVENT: '\u{1F4DD}' };
/** What _execute_add_game_note appends (bt:188-198). */
function formatNote(noteType, content) {
  const prefix = `${NOTE_EMOJI[noteType] || NOTE_EMOJI.EVENT} ${noteType}:`;
  return content.startsWith(prefix) ? content : `${prefix} ${content}`;
}

function plurality(votes, n) {
  let best = 0, bestC = 0;
  for (let k = 1; k <= n; k++) {
    const c = votes.filter((v) => v === k).length;
    if (c > bestC) { best = k; bestC = c; }
  }
  return best;
}

/**
 * One stepped turn of one room as the reference's backend tool calls
 * (agent/tools/backend_tools.py:10-157), in node order: update_player_actions* ->
 * set_next_phase -> update_player_state* -> add_game_note*.  Same rendering as the Python host
 * (game_engine_amd/toolcalls.py); applying the calls to the reference's dict state reproduces
 * the GPU state, and the notes are the fixed policy's (pinned by tests/golden/strings_*.json).
 * `before`/`after`: RoomState; `event`: from RoomBatch.readEvents().
 */
function turnToolCalls(table, before, after, event) {
  const calls = [];
  const ids = Object.keys(after.player_states).sort((a, b) => Number(a) - Number(b));
  const n = ids.length;
  const from = table.info.phases.find((x) => x.id === event.from_phase_id);
  const to = table.info.phases.find((x) => x.id === event.to_phase_id);
  ids.forEach((pid, i) => {
    if ((event.acted_now >> i) & 1) {
      const c = event.choice[i];
      calls.push({ name: 'update_player_actions', args: { player_id: pid, actions: `[t=${event.turn}|c=${c}] ${actionText(from.act, i + 1, c)}`, phase: from.name } });
    }
  });
  const moved = event.to_phase_id !== event.from_phase_id;
  calls.push({ name: 'set_next_phase', args: { transition: moved, next_phase_id: event.to_phase_id, transition_reason: moved ? 'phase complete' : 'waiting' } });
  const deaths = [];
  const names = table.info.fieldNames;
  const SA = after.slots, SB = before.slots;                      // slot values (WW: 0 role, 2 is_alive, 8 target; TT: 0 speaker, 1 submitted, 7 score)
  const ww = after.pack === 1;
  ids.forEach((pid, i) => {
    const b = before.player_states[pid], a = after.player_states[pid];
    for (const name of Object.keys(a)) {
      if (JSON.stringify(b[name]) !== JSON.stringify(a[name])) calls.push({ name: 'update_player_state', args: { player_id: pid, state_name: name, state_value: a[name] } });
    }
    if (ww && SB[i][2] && !SA[i][2]) deaths.push([pid, SA[i][0]]);
    if (!ww && names[9]) {                                         // `statements`: text the record does not carry
      if (SA[i][1] && !SB[i][1]) {
        const st = {}; [1, 2, 3].forEach((s) => { st[String(s)] = `Statement ${s} of Player ${i + 1}`; });
        calls.push({ name: 'update_player_state', args: { player_id: pid, state_name: names[9], state_value: st } });
      } else if (SB[i][1] && !SA[i][1]) {
        calls.push({ name: 'update_player_state', args: { player_id: pid, state_name: names[9], state_value: {} } });
      }
    }
  });
  if (!moved) return calls;
  const note = (kind, text) => calls.push({ name: 'add_game_note', args: { note_type: kind, content: text } });
  note('PHASE_STATUS', `[t=${event.turn}] phase ${event.from_phase_id} -> ${event.to_phase_id}`);
  if (to.effect === 1) {                                          // GE_EFF_ASSIGN_ROLES
    note('NEXT_PHASE', 'Roles assigned: ' + SA.map((p, i) => `Player${i + 1}=${p[0]}`).join(', '));
  } else if (to.effect === 3 || to.effect === 4) {               // NIGHT_RESOLVE / DAY_RESOLVE
    const how = to.effect === 3 ? 'overnight by the werewolves' : 'by day vote';
    deaths.forEach(([pid, role]) => note('CRITICAL', `Player ${pid} (${role}) eliminated ${how} - marked is_alive=false`));
    if (to.effect === 3 && !deaths.length) {
      const roles = table.info.roleNames;                        // class 2 = Werewolf, 3 = Doctor
      const victim = plurality(SA.filter((p, i) => SB[i][2] && SB[i][0] === roles[2]).map((p) => p[8]), n);
      let protect = 0;
      SA.forEach((p, i) => { if (SB[i][2] && SB[i][0] === roles[3]) protect = p[8]; });
      note('DECISION', `Werewolves targeted Player ${victim}, Doctor protected Player ${protect} - no elimination`);
    }
  } else if (to.effect === 5) {                                   // TT_ROUND_START
    const sp = SA.findIndex((p) => p[0]);
    note('DECISION', `Selected Player ${sp + 1} as next speaker (turn_order)`);
  } else if (to.effect === 7) {                                   // TT_SCORE
    if (SB.some((p) => p[0])) note('SCORE_UPDATE', 'Total scores - ' + SA.map((p, i) => `Player ${i + 1}: ${p[7]}`).join(', '));
  }
  return calls;
}

/**
 * The log-shaped parts of one room's AgentState that the packed state does not carry - playerActions
 * (bt:285-344), game_notes (bt:163-202), phase_history (v2:1207-1215), the Two-Truths `statements` texts
 * and the players' names - kept by folding each turn's tool calls as the reference's `_execute_*` would.
 */
class RoomLog {
  constructor(table, names, gameName = '') {
    this.table = table; this.names = names.slice(); this.gameName = gameName;
    this.playerActions = {}; this.gameNotes = []; this.phaseHistory = []; this.statements = {};
  }
  /** Apply one turn's calls; `after`: the RoomState after the turn. */
  fold(calls, after, now = Date.now()) {
    for (const c of calls) {
      const a = c.args;
      if (c.name === 'update_player_actions') {
        const pid = a.player_id;
        const rec = this.playerActions[pid] || (this.playerActions[pid] = { name: this.names[Number(pid) - 1], actions: {} });
        const id = String(Object.values(rec.actions).reduce((m, x) => Math.max(m, Number(x.id)), 0) + 1);   // per-player sequence, bt:323-332
        rec.name = this.names[Number(pid) - 1];
        rec.actions[id] = { action: a.actions, timestamp: now, phase: a.phase, id };
      } else if (c.name === 'add_game_note') {
        this.gameNotes.push(formatNote(a.note_type, a.content));
      } else if (c.name === 'update_player_state' && this.table.info.pack === 2 && a.state_name === this.table.info.fieldNames[9]) {
        this.statements[a.player_id] = Object.assign({}, a.state_value);
      }
    }
    this.phaseHistory.push({ phase_id: after.current_phase_id, phase_name: after.current_phase_name, timestamp: new Date(now).toISOString() });   // every turn, v2:1207-1215
  }
  /** File a person's game message as process_human_action_if_needed does (agent/tools/utils.py:343-350): under Player 1,
   * the first 200 characters, and under PHASE 0's NAME whatever the current phase is (InitialRouterNode passes state keys that
   * do not exist, agent/game_agent_v2.py:324-331) - mirrored, not corrected (POLICY.md 3b). */
  personMessage(text, now = Date.now()) {
    const rec = this.playerActions['1'] || (this.playerActions['1'] = { name: this.names[0], actions: {} });
    const id = String(Object.values(rec.actions).reduce((m, x) => Math.max(m, Number(x.id)), 0) + 1);
    rec.name = this.names[0];
    rec.actions[id] = { action: Array.from(String(text)).slice(0, 200).join(''), timestamp: now, phase: this.table.phaseName(0), id };
  }
  /** AgentState of the room (v2:97-117), player_states in the reference's key order. */
  agentState(room) {
    const ps = {};
    Object.keys(room.player_states).forEach((pid, i) => {
      const out = { name: this.names[i] };
      for (const [k, v] of Object.entries(room.player_states[pid])) {
        out[k] = v;
        const fn = this.table.info.fieldNames;
        if (this.table.info.pack === 2 && k === fn[0] && fn[9]) out[fn[9]] = Object.assign({}, this.statements[pid] || {});
      }
      ps[pid] = out;
    });
    return { gameName: this.gameName, current_phase_id: room.current_phase_id, current_phase_name: room.current_phase_name,
             player_states: ps, playerActions: this.playerActions, phase_history: this.phaseHistory, game_notes: this.gameNotes };
  }
}

class RoomBatch {
  /** segments: [{table: GameTable, nPlayers, nRooms}] */
  constructor({ segments, seed = 0n, firstRoom = 0n, device = 0, maxFuse = 0, restart = false, trace = false }) {
    this.segments = segments;
    this.handle = addon.createBatch({
      seed, firstRoom, device, maxFuse, restart, trace,
      segments: segments.map((s) => ({ table: s.table.handle, nPlayers: s.nPlayers, nRooms: s.nRooms, humanMask: s.humanMask || 0 })),
    });
    this.nRooms = segments.reduce((a, s) => a + s.nRooms, 0);
    this._tail = Promise.resolve();     // async steps of one handle run strictly one after the other
  }
  /** Advance every room by nTurns turns; resolves with the batch's turn counter.  The C handle is not
   * thread-safe: async steps of one batch are chained, and any synchronous call (readRoom, injectAction,
   * summary ...) made while one is in flight throws GE_BUSY instead of racing with the worker thread —
   * `await` the step, or queue the call with `whenIdle`. */
  step(nTurns = 1) {
    const p = this._tail.then(() => addon.step(this.handle, nTurns));
    this._tail = p.catch(() => {});
    return p;
  }
  /** Runs fn() once every step queued so far has finished (and before any queued later). */
  whenIdle(fn) {
    const p = this._tail.then(fn);
    this._tail = p.catch(() => {});
    return p;
  }
  stepSync(nTurns = 1) { return addon.stepSync(this.handle, nTurns); }
  reset() { addon.reset(this.handle); }
  /** Restore a checkpoint: room records (readRoomsRaw's ArrayBuffer) and the turn they were taken at. */
  writeRoomsRaw(first, buffer) { addon.writeRooms(this.handle, first, buffer); }
  readRoomsRaw(first, count) { return addon.readRooms(this.handle, first, count); }
  setTurn(turn) { addon.setTurn(this.handle, turn); }
  /** Releases the device memory now (otherwise at garbage collection). */
  close() { if (this.handle) { addon.destroyBatch(this.handle); this.handle = null; } }
  /** Log an action of a host-driven (human) player in the room's current phase (segment.humanMask). */
  injectAction(room, playerId, choice) { addon.injectAction(this.handle, room, playerId, choice); }
  /** Many at once (one kernel): rooms[], playerIds[], choices[] -> Int32Array of per-action status (0 = applied). */
  injectActions(rooms, playerIds, choices) {
    return addon.injectActions(this.handle, BigUint64Array.from(rooms, (r) => BigInt(r)), Uint32Array.from(playerIds), Uint32Array.from(choices));
  }
  tableOf(room) {
    let base = 0;
    for (const s of this.segments) { if (room < base + s.nRooms) return s.table; base += s.nRooms; }
    throw new RangeError(`room ${room}`);
  }
  /** AgentState-shaped view of one room (agent/game_agent_v2.py:97-117). */
  readRoom(room) { return decodeRoom(this.tableOf(room), addon.readRooms(this.handle, room, 1), 0); }
  readRooms(first, count) {
    const buf = addon.readRooms(this.handle, first, count);
    const out = [];
    for (let i = 0; i < count; i++) out.push(decodeRoom(this.tableOf(first + i), buf, i * VIEW.size));
    return out;
  }
  /** [room][turn] events of the most recent step() (batch created with trace: true). */
  readEvents(first, count) {
    const { nTurns, buffer } = addon.readEvents(this.handle, first, count);
    const out = [];
    for (let r = 0; r < count; r++) {
      const row = [];
      for (let t = 0; t < nTurns; t++) {
        const off = (r * nTurns + t) * EVENT_SIZE;
        const dv = new DataView(buffer, off, EVENT_SIZE);
        row.push({ turn: dv.getUint32(0, true), from_phase_id: dv.getInt32(4, true), to_phase_id: dv.getInt32(8, true),
                   acted_now: dv.getUint16(12, true), restarted: dv.getUint8(14), choice: Array.from(new Uint8Array(buffer, off + 16, 16)) });
      }
      out.push(row);
    }
    return out;
  }
  summary() { return decodeSummary(addon.summary(this.handle)); }
}

/**
 * The same rooms spread over several GPUs of one node, from ONE Node process (SURVEY §8e): device d
 * owns the global rooms [firstRoom + d*R, firstRoom + (d+1)*R) with R = rooms per device, so per-room
 * results equal those of a single batch or of any other device count (the RNG is keyed by the global
 * index).  Rooms never interact: step() just runs every shard's launches concurrently (one async
 * work item per shard, each on its own device); the only cross-GPU step is summary(), a host-side
 * sum of the fixed-size per-device summaries (the multi-process Python host does the same sum after
 * one RCCL all-gather, game_engine_amd/dist.py).
 */
class ShardedBatch {
  /** segments: per-device segments (every device gets the same mix); devices: HIP device indices */
  constructor({ segments, devices, seed = 0n, firstRoom = 0n, maxFuse = 0, restart = false, trace = false }) {
    if (!devices || !devices.length) throw new RangeError('devices');
    this.roomsPerDevice = segments.reduce((a, s) => a + s.nRooms, 0);
    this.shards = devices.map((device, d) => new RoomBatch({
      segments, seed, device, maxFuse, restart, trace,
      firstRoom: BigInt(firstRoom) + BigInt(d) * BigInt(this.roomsPerDevice),
    }));
    this.nRooms = this.roomsPerDevice * devices.length;
  }
  async step(nTurns = 1) { return (await Promise.all(this.shards.map((b) => b.step(nTurns))))[0]; }
  reset() { this.shards.forEach((b) => b.reset()); }
  close() { this.shards.forEach((b) => b.close()); }
  shardOf(room) {
    if (!(room >= 0 && room < this.nRooms)) throw new RangeError(`room ${room}`);
    return [this.shards[Math.floor(room / this.roomsPerDevice)], room % this.roomsPerDevice];
  }
  readRoom(room) { const [b, r] = this.shardOf(room); return b.readRoom(r); }
  injectAction(room, playerId, choice) { const [b, r] = this.shardOf(room); b.injectAction(r, playerId, choice); }
  /** whole-job summary: every field is a sum over rooms (checksum: mod 2^64), `turn` is common */
  summary() {
    const parts = this.shards.map((b) => b.summary());
    const add = (f) => BigInt.asUintN(64, parts.reduce((a, p) => a + p[f], 0n));
    const addHist = (f) => parts[0][f].map((_, i) => BigInt.asUintN(64, parts.reduce((a, p) => a + p[f][i], 0n)));
    return {
      rooms: add('rooms'), finished: add('finished'), village_wins: add('village_wins'), wolf_wins: add('wolf_wins'),
      alive_players: add('alive_players'), sum_end_turn: add('sum_end_turn'), end_turn_hist: addHist('end_turn_hist'),
      score_hist: addHist('score_hist'), checksum: add('checksum'), turn: parts[0].turn, games_recycled: add('games_recycled'),
    };
  }
}

function decodeSummary(buffer) {
  const w = new BigUint64Array(buffer);
  return {
    rooms: w[0], finished: w[1], village_wins: w[2], wolf_wins: w[3], alive_players: w[4],
    sum_end_turn: w[5], end_turn_hist: Array.from(w.slice(6, 22)), score_hist: Array.from(w.slice(22, 38)),
    checksum: w[38], turn: w[39], games_recycled: w[40],
  };
}

/**
 * The native device group (ge_group_*, include/ge_step.h; SURVEY §8e process model): ONE Node process drives N
 * distinct GPUs of a node.  `segments` describe the WHOLE job; device i of n owns the i-th of n contiguous parts of
 * every segment, and every room keeps the global index it has in one RoomBatch of the same arguments - results are
 * identical to that batch's for any device count.  step() runs all devices concurrently on the libuv pool; summary()
 * is the per-device reductions + ONE RCCL all-gather over xGMI + the sum, all inside libge_step.so (RCCL is loaded at
 * run time; there is none in the host).  ShardedBatch above is the host-side form of the same thing (its summary is a
 * host sum): it also runs with several shards on ONE device, which RCCL refuses, and serves as the cross-check.
 */
/**
 * Where a room of the WHOLE job lives after the device group's sharding (ge_group_partition, csrc/ge_host.h group_partition:
 * part i of n holds rooms [floor(R i / n), floor(R (i + 1) / n)) of every segment of R rooms, segment by segment):
 * [part, index inside that part's batch, segment] for `room` counted segment-major as in one RoomBatch.
 */
function locateInShards(segmentRooms, n, room) {
  let base = 0;
  for (let k = 0; k < segmentRooms.length; k++) {
    const R = segmentRooms[k];
    if (room < base + R) {
      const r = room - base;
      const part = (j, Rj) => Number(BigInt(Rj) * BigInt(j) / BigInt(n));     // exact floor, whatever the room count
      let i = Math.min(n - 1, Math.floor((r + 1) * n / R));
      while (i > 0 && part(i, R) > r) i--;
      while (part(i + 1, R) <= r) i++;
      let local = r - part(i, R);                          // rooms of the earlier segments on this part come first
      for (let j = 0; j < k; j++) local += part(i + 1, segmentRooms[j]) - part(i, segmentRooms[j]);
      return [i, local, k];
    }
    base += R;
  }
  throw new RangeError(`room ${room}`);
}

class DeviceGroup {
  constructor({ segments, devices, seed = 0n, firstRoom = 0n, maxFuse = 0, restart = false, trace = false }) {
    if (!devices || !devices.length) throw new RangeError('devices');
    this.segments = segments;
    this.devices = devices.slice();
    this.handle = addon.createGroup({
      seed, firstRoom, maxFuse, restart, trace, devices,
      segments: segments.map((s) => ({ table: s.table.handle, nPlayers: s.nPlayers, nRooms: s.nRooms, humanMask: s.humanMask || 0 })),
    });
    this.nRooms = segments.reduce((a, s) => a + s.nRooms, 0);
    this._tail = Promise.resolve();
  }
  /** Advance every room of every device by nTurns turns (async steps of one group are chained, as for RoomBatch). */
  step(nTurns = 1) {
    const p = this._tail.then(() => addon.groupStep(this.handle, nTurns));
    this._tail = p.catch(() => {});
    return p;
  }
  whenIdle(fn) {
    const p = this._tail.then(fn);
    this._tail = p.catch(() => {});
    return p;
  }
  /** whole-job summary: per-device reductions, one ncclAllGather of the ge_summary records, the sum */
  summary() { return decodeSummary(addon.groupSummary(this.handle)); }
  /** [shard, local index, table] of a room given by its index in segment-major order (the order of one RoomBatch) */
  locate(room) {
    const [i, local, k] = locateInShards(this.segments.map((s) => s.nRooms), this.devices.length, room);
    return [i, local, this.segments[k].table];
  }
  readRoom(room) {
    const [shard, local, table] = this.locate(room);
    return decodeRoom(table, addon.groupReadRooms(this.handle, shard, local, 1), 0);
  }
  close() { if (this.handle) { addon.destroyGroup(this.handle); this.handle = null; } }
}

const { compileCriteria, audienceGroups, uiToolCalls } = require('./ui_script.js');

module.exports = { GameTable, RoomBatch, ShardedBatch, DeviceGroup, locateInShards, RoomLog, formatNote, loadDslByGamename, findGameFile, initializePlayers, turnToolCalls, compileCriteria, audienceGroups, uiToolCalls,
                   deviceCount: addon.deviceCount, addon };
