'use strict';
/**
 * RoomService — the single-room drop-in: one LangGraph thread (= one room) served by an N=1
 * batch through the same C ABI as the big batches.
 *
 * In the reference, src/app/api/copilotkit/route.ts:22-47 binds each request's X-Thread-ID to a
 * LangGraphAgent and every "Continue" / "Start game." message (src/app/page.tsx:2774, 2962) is one
 * run of the Python graph.  `continueRoom(threadId)` is that run: it advances the room by one turn
 * and returns what the graph returns to CopilotKit — the AgentState (agent/game_agent_v2.py:97-117)
 * plus the AIMessage's tool calls: here the backend calls that describe the turn
 * (turnToolCalls) and the frontend calls of the phase now showing (uiToolCalls).
 * A handler that wants this instead of the LLM graph calls it in place of constructing the
 * LangGraphAgent (route.ts:30-39); `serve()` exposes the same over plain HTTP for a quick try.
 */
const http = require('http');
const { GameTable, RoomBatch, RoomLog, loadDslByGamename, turnToolCalls, uiToolCalls } = require('./index.js');
const M = require('./messages.js');

/** stable 48-bit room index from a thread id (the RNG is keyed by it) */
function roomIndexOf(threadId) {
  let h = 0xcbf29ce484222325n;
  for (const ch of Buffer.from(String(threadId), 'utf8')) { h ^= BigInt(ch); h = (h * 0x100000001b3n) & 0xffffffffffffffffn; }
  return h & 0xffffffffffffn;
}

class RoomService {
  constructor({ gamesDir = 'games', seed = 0n, device = 0 } = {}) {
    this.gamesDir = gamesDir; this.seed = BigInt(seed); this.device = device;
    this.tables = new Map();       // gameName -> GameTable
    this.rooms = new Map();        // threadId -> { batch, table, state, phaseHistory, playerActions, gameNotes, names }
  }
  table(gameName, dsl) {
    if (!this.tables.has(gameName)) this.tables.set(gameName, dsl ? new GameTable(dsl) : GameTable.fromGamename(gameName, this.gamesDir));
    return this.tables.get(gameName);
  }
  /** roomSession.players as the lobby builds it (src/app/game-library/[game]/room/page.tsx:261-369). */
  /** players[i].isBot === false marks a human seat: the bot policy never acts for it (humanAction does). */
  createRoom({ threadId, gameName, players, dsl, roomIndex }) {
    const table = this.table(gameName, dsl);
    const humanMask = players.reduce((m, p, i) => (p.isBot === false ? m | (1 << i) : m), 0);
    const batch = new RoomBatch({ segments: [{ table, nPlayers: players.length, nRooms: 1, humanMask }], seed: this.seed,
                                  firstRoom: roomIndex === undefined ? roomIndexOf(threadId) : BigInt(roomIndex),   // the RNG is keyed by it
                                  device: this.device, maxFuse: 1, trace: true });
    if (this.rooms.has(threadId)) this.close(threadId);
    const names = players.map((p, i) => p.name || `Player ${i + 1}`);
    const humanSeats = players.map((p, i) => (p.isBot === false ? i + 1 : 0)).filter((x) => x);
    const room = { batch, table, gameName, names, humanSeats, panel: null, state: batch.readRoom(0), log: new RoomLog(table, names, gameName), queue: Promise.resolve() };
    this.rooms.set(threadId, room);
    return this.agentState(room);
  }
  agentState(room) { return room.log.agentState(room.state); }
  /** Requests of one thread run strictly one after the other (the reference's LangGraph server queues
   * runs per thread the same way): overlapping /continue and /action calls neither race on the batch
   * handle nor see a half-updated log. */
  _serial(room, fn) {
    const p = room.queue.then(fn);
    room.queue = p.catch(() => {});
    return p;
  }
  /** Forget a thread and free its device memory (after queued requests have finished). */
  close(threadId) {
    const room = this.rooms.get(threadId);
    if (!room) return Promise.resolve(false);
    this.rooms.delete(threadId);
    return this._serial(room, () => { room.batch.close(); return true; });
  }
  /** A human's vote / choice (the frontend's "Player X voted ..." message, src/app/page.tsx:302-305). */
  humanAction(threadId, playerId, choice) {
    const room = this.rooms.get(threadId);
    if (!room) return Promise.reject(new Error(`unknown thread ${threadId}`));
    return this._serial(room, () => {
      room.batch.injectAction(0, playerId, choice);
      room.state = room.batch.readRoom(0);
      return this.agentState(room);
    });
  }
  /** One turn (one graph run).  Returns { state, toolCalls, uiCalls }. */
  /** items: the frontend's canvas items (AgentState.items) when the caller has them (clearCanvas exemptList). */
  continueRoom(threadId, items) {
    const room = this.rooms.get(threadId);
    if (!room) return Promise.reject(new Error(`unknown thread ${threadId}`));
    return this._serial(room, () => this._continue(room, items));
  }
  /**
   * The drop-in's message-level entry: what the reference's graph does with ONE message of the browser
   * (src/app/page.tsx:183-259 -> agent/game_agent_v2.py:198-349, agent/tools/utils.py:310-358; POLICY.md 3b).
   *   chat ("... in game chat: ..." / "... to Bot k: ...")  -> ChatBotNode: no turn, no state change;
   *   control ("Start game.", "Continue")                    -> one turn;
   *   anything else -> logged verbatim under Player 1 (200 characters, phase 0's name - the reference's own quirk), read as a
   *                    seat's action where it is one (a vote on the newest panel, an input for the statements phase), then one turn.
   * Resolves { state, toolCalls, uiCalls, played, kind }; an action message that is no valid game action is still logged and
   * still plays the turn, as in the reference.
   */
  handleMessage(threadId, text, items) {
    const room = this.rooms.get(threadId);
    if (!room) return Promise.reject(new Error(`unknown thread ${threadId}`));
    return this._serial(room, async () => {
      const kind = M.classify(text);
      if (kind === M.CHAT) return { state: this.agentState(room), toolCalls: [], uiCalls: [], played: false, kind };
      if (kind === M.ACTION) {
        room.log.personMessage(text);
        const st = room.state, info = room.table.info;
        const phase = info.phases.find((x) => x.id === st.current_phase_id);
        const alive = st.slots.map((v) => (st.pack === 1 ? !!v[2] : true));
        for (const [seat, choice] of M.resolve(text, room.panel, phase ? phase.act : 0, st.pack, room.names, alive, room.humanSeats)) {
          try { room.batch.injectAction(0, seat, choice); break; } catch (e) {
            if (e.code !== 'GE-1') throw e;            // GE_ERR_ARG: not a living pending target of this phase - logged, no game effect
          }
        }
      }
      const out = await this._continue(room, items);
      return Object.assign(out, { played: true, kind });
    });
  }
  async _continue(room, items) {
    // `before` is the state BEFORE any injected action of this message: the person's record writes then show up among the
    // turn's update_player_state calls, where the reference's Referee issues them
    const before = room.state;
    await room.batch.step(1);
    const after = room.batch.readRoom(0);
    const event = room.batch.readEvents(0, 1)[0][0];
    const toolCalls = turnToolCalls(room.table, before, after, event);
    // fold the calls into the log-shaped parts of AgentState the packed state does not carry
    // (playerActions / game_notes / phase_history, as backend_tools.py:163-202, 285-344 would)
    room.log.fold(toolCalls, after);
    room.state = after;
    const state = this.agentState(room);
    const deaths = toolCalls.filter((c) => c.name === 'update_player_state' && c.args.state_name === 'is_alive' && c.args.state_value === false).map((c) => c.args.player_id);
    const uiCalls = uiToolCalls(room.table.dsl, state, { table: room.table, turn: event.turn, deaths, items });
    room.panel = M.newestPanel(uiCalls);               // what a person's next vote message can answer
    return { state, toolCalls, uiCalls };
  }
  serve(port = 8124) {
    const server = http.createServer((req, res) => {
      let body = '';
      req.on('data', (d) => { body += d; });
      req.on('end', async () => {
        try {
          const msg = body ? JSON.parse(body) : {};
          let out;
          if (req.method === 'POST' && req.url === '/rooms') out = this.createRoom(msg);
          else if (req.method === 'POST' && req.url === '/continue') out = await this.continueRoom(msg.threadId, msg.items);
          else if (req.method === 'POST' && req.url === '/message') out = await this.handleMessage(msg.threadId, msg.text, msg.items);
          else if (req.method === 'POST' && req.url === '/action') out = await this.humanAction(msg.threadId, msg.playerId, msg.choice);
          else if (req.method === 'POST' && req.url === '/close') out = { closed: await this.close(msg.threadId) };
          else { res.writeHead(404); res.end(); return; }
          res.writeHead(200, { 'content-type': 'application/json' });
          res.end(JSON.stringify(out));
        } catch (e) { res.writeHead(400); res.end(JSON.stringify({ error: String(e.message || e) })); }
      });
    });
    return new Promise((resolve) => server.listen(port, '127.0.0.1', () => resolve(server)));
  }
}

module.exports = { RoomService, roomIndexOf };
