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

const TEAMS = ['', 'villagers', 'werewolves'];
const VIEW = { size: addon.roomViewSize(), players: 20, det: 20 + 16 * 12 };

class GameTable {
  constructor(dsl, rounds = 1) {
    if (!dsl || typeof dsl !== 'object' || !Object.keys(dsl).length) throw new Error('empty DSL');
    this.dsl = dsl;
    this.handle = addon.compileTable(JSON.stringify(dsl), rounds);
    this.info = addon.tableInfo(this.handle);
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
  for (let i = 0; i < n; i++) {
    const f = u8.slice(VIEW.players + 12 * i, VIEW.players + 12 * i + 12);
    if (pack === 1) {
      const mem = {};
      if (f[0] === 4) det.forEach((d, k) => { if (d) mem[String(k + 1)] = TEAMS[d]; });
      playerStates[String(i + 1)] = {
        role: table.info.roleNames[f[0]], team: TEAMS[f[1]], is_alive: !!f[2], role_revealed: !!f[3],
        can_vote: !!f[4], has_secret_role: !!f[5], night_action_eligible: !!f[6],
        night_action_submitted: !!f[7], selected_target_id: f[8], investigated_alignments: mem,
      };
    } else {
      playerStates[String(i + 1)] = {
        is_speaker: !!f[0], statements_submitted: !!f[1], lie_index: f[2], lie_revealed: !!f[3],
        can_vote: !!f[4], vote_choice: f[5], has_voted: !!f[6], total_score: f[7], rounds_as_speaker: f[8],
      };
    }
  }
  const phaseId = dv.getInt32(0, true);
  return {
    current_phase_id: phaseId, current_phase_name: table.phaseName(phaseId),
    previous_phase_id: dv.getInt32(4, true), end_turn: dv.getInt32(8, true), games: dv.getInt32(12, true),
    player_states: playerStates,
    acted: Array.from({ length: n }, (_, i) => u8[VIEW.players + 12 * i + 9]),
    choice: Array.from({ length: n }, (_, i) => u8[VIEW.players + 12 * i + 10]),
  };
}

class RoomBatch {
  /** segments: [{table: GameTable, nPlayers, nRooms}] */
  constructor({ segments, seed = 0n, firstRoom = 0n, device = 0, maxFuse = 0, restart = false }) {
    this.segments = segments;
    this.handle = addon.createBatch({
      seed, firstRoom, device, maxFuse, restart,
      segments: segments.map((s) => ({ table: s.table.handle, nPlayers: s.nPlayers, nRooms: s.nRooms })),
    });
    this.nRooms = segments.reduce((a, s) => a + s.nRooms, 0);
  }
  /** Advance every room by nTurns turns; resolves with the batch's turn counter. */
  step(nTurns = 1) { return addon.step(this.handle, nTurns); }
  stepSync(nTurns = 1) { return addon.stepSync(this.handle, nTurns); }
  reset() { addon.reset(this.handle); }
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
  summary() {
    const w = new BigUint64Array(addon.summary(this.handle));
    return {
      rooms: w[0], finished: w[1], village_wins: w[2], wolf_wins: w[3], alive_players: w[4],
      sum_end_turn: w[5], end_turn_hist: Array.from(w.slice(6, 22)), score_hist: Array.from(w.slice(22, 38)),
      checksum: w[38], turn: w[39], games_recycled: w[40],
    };
  }
}

module.exports = { GameTable, RoomBatch, loadDslByGamename, deviceCount: addon.deviceCount, addon };
