'use strict';
/**
 * Deterministic UI script (JS twin of game_engine_amd/ui_script.py): the frontend tool calls of a
 * room's current phase, from the DSL's `actions[].tools`, with the audience each call is for.
 * Replaces what the reference's ActionExecutor / UIUpdateNode LLM produces
 * (agent/game_agent_v2.py:1243-1568); the calls are executed by the existing useCopilotAction
 * handlers (src/app/page.tsx:950-2500), which filter on audience_type / audience_ids
 * (src/components/canvas/CardRenderer.tsx:58-70).
 */

function literal(text) {
  const t = text.trim();
  if (t.toLowerCase() === 'true') return true;
  if (t.toLowerCase() === 'false') return false;
  if (t.length >= 2 && t[0] === t[t.length - 1] && (t[0] === "'" || t[0] === '"')) return t.slice(1, -1);
  const n = Number(t);
  if (Number.isNaN(n)) throw new Error(`unsupported literal: ${text}`);
  return n;
}

function splitOutside(expr, word) {
  const out = [];
  const pat = ` ${word} `;
  let cur = '', quote = '', depth = 0;
  for (let i = 0; i < expr.length;) {
    const c = expr[i];
    if (quote) { if (c === quote) quote = ''; } else if (c === "'" || c === '"') quote = c;
    else if (c === '[') depth++;
    else if (c === ']') depth--;
    else if (depth === 0 && expr.substr(i, pat.length).toLowerCase() === pat) { out.push(cur); cur = ''; i += pat.length; continue; }
    cur += c; i++;
  }
  out.push(cur);
  return out;
}

/**
 * `player.team == 'werewolves' and player.is_alive == true` -> predicate over one player's state.  The same grammar as
 * phase target conditions (dsl_phases_generation_prompt.txt:120-132): == != < <= > >=, in [..] / not in [..], terms joined
 * by `and`, alternatives by `or` (and binds tighter; no parentheses).  Same semantics as game_engine_amd/ui_script.py.
 */
const criteriaCache = new Map();                 // a DSL's criteria are compiled once, not once per turn
function compileCriteria(expr) {
  let f = criteriaCache.get(expr);
  if (!f) {
    f = compileCriteriaUncached(expr);
    if (criteriaCache.size < 1024) criteriaCache.set(expr, f);
  }
  return f;
}
function compileCriteriaUncached(expr) {
  const flat = expr.split(/\s+/).filter((x) => x).join(' ');
  if (/[()]/.test(flat.replace(/'[^']*'|"[^"]*"/g, ''))) throw new Error(`unsupported selection criterion (parentheses): ${expr}`);
  const clauses = splitOutside(flat, 'or').map((alt) => splitOutside(alt, 'and').map((part) => {
    const m = /^\s*player\.(\w+)\s*(==|!=|<=|>=|<|>|not\s+in|in)\s*(.+?)\s*$/i.exec(part);
    if (!m) throw new Error(`unsupported selection criterion: ${part}`);
    const field = m[1], op = m[2].toLowerCase().split(/\s+/).join(' '), rhs = m[3];
    if (op === 'in' || op === 'not in') {
      const inner = rhs.trim();
      if (!(inner.startsWith('[') && inner.endsWith(']'))) throw new Error(`unsupported list literal: ${rhs}`);
      return [field, op, inner.slice(1, -1).split(',').filter((x) => x.trim()).map(literal)];
    }
    const val = literal(rhs);
    if (['<', '<=', '>', '>='].includes(op) && typeof val !== 'number') throw new Error(`unsupported comparison: ${part}`);
    return [field, op, val];
  }));
  const holds = (player, field, op, val) => {
    const have = player[field];
    if (op === '==' || op === '!=') return (have === val) !== (op === '!=');
    if (op === 'in' || op === 'not in') return val.includes(have) !== (op === 'not in');
    if (typeof have !== 'number') return false;
    return op === '<' ? have < val : op === '<=' ? have <= val : op === '>' ? have > val : have >= val;
  };
  return (player) => clauses.some((terms) => terms.every(([f, o, v]) => holds(player, f, o, v)));
}

const byId = (a, b) => Number(a) - Number(b);

function audienceGroups(dsl, playerStates) {
  const out = {};
  const groups = (dsl.declaration || {}).audience_groups || {};
  const ids = Object.keys(playerStates).sort(byId);
  for (const [name, g] of Object.entries(groups)) {
    const pred = compileCriteria(g.selection_criteria || '');
    out[name] = ids.filter((p) => pred(playerStates[p]));
  }
  return out;
}

function audienceFor(desc, ps, groups) {
  const d = desc.toLowerCase();
  const ids = Object.keys(ps).sort(byId);
  const isAlive = (p) => (ps[p].is_alive === undefined ? true : ps[p].is_alive);
  const alive = ids.filter(isAlive);
  const role = (name) => alive.filter((p) => String(ps[p].role || '').toLowerCase() === name);
  if (d.includes('non-werewol')) return alive.filter((p) => ps[p].team !== 'werewolves');
  if (d.includes('werewol')) return groups.werewolves || alive.filter((p) => ps[p].team === 'werewolves');
  for (const r of ['doctor', 'detective']) {
    if (d.includes(`except the ${r}`)) { const rr = role(r); return alive.filter((p) => !rr.includes(p)); }
    if (d.includes(r)) return role(r);
  }
  if (d.includes('eliminated players') || d.includes('dead players')) return groups.dead_players || ids.filter((p) => !isAlive(p));
  if (d.includes('eligible voters') || d.includes('voters')) return groups.voters || alive.filter((p) => (ps[p].can_vote === undefined ? true : ps[p].can_vote));
  if (d.includes('non-speaker')) return ids.filter((p) => !ps[p].is_speaker);
  if (d.includes('speaker')) return ids.filter((p) => ps[p].is_speaker);
  return null;
}

const path = require('path');
let toolsCache = null;
/** {tool: [[param, type, required], ...]} of the frontend handlers (page.tsx useCopilotAction blocks). */
function frontendTools() {
  if (!toolsCache) toolsCache = JSON.parse(require('fs').readFileSync(path.join(__dirname, '..', 'frontend_tools.json'), 'utf8'));
  return toolsCache.tools;
}
/** Problems of a frontend tool call against its handler's declared parameters ([] = fine). */
function validateCall(call) {
  const spec = frontendTools()[call.name];
  if (!spec) return [`no frontend handler named ${call.name}`];
  const known = new Set(spec.map((p) => p[0]));
  const out = spec.filter((p) => p[2] && (call.args[p[0]] === undefined || call.args[p[0]] === null || call.args[p[0]] === ''))
    .map((p) => `${call.name}: missing required parameter ${p[0]}`);
  for (const k of Object.keys(call.args)) if (!known.has(k)) out.push(`${call.name}: unknown parameter ${k}`);
  return out;
}

// the fixed grid plan: where each kind of component goes (the handlers' `position` select lists)
const POSITION = { createPhaseIndicator: 'top-center', createTextDisplay: 'center', createVotingPanel: 'center', createResultDisplay: 'center',
                   createCharacterCard: 'bottom-center', createScoreBoard: 'top-right', createTurnIndicator: 'top-left', createStatementBoard: 'middle-left' };
const DEATH_POSITIONS = ['bottom-left', 'bottom-center', 'bottom-right', 'middle-left', 'middle-right', 'top-left', 'top-right'];
const LABEL = { createPhaseIndicator: 'phase', createTextDisplay: 'text', createVotingPanel: 'vote', createResultDisplay: 'result',
                createCharacterCard: 'role card', createScoreBoard: 'scores', createTurnIndicator: 'turn', createStatementBoard: 'statements',
                createAvatarSet: 'avatars', createTimer: 'timer', createDeathMarker: 'death' };
// item types a clearCanvas description asks to keep (src/lib/canvas/types.ts item types)
const EXEMPT = [['death marker', 'death_marker'], ['elimination indicator', 'death_marker'], ['scoreboard', 'score_board'], ['score board', 'score_board']];
const DISCUSSION_SECONDS = 60;   // the DSL's timer phases give no duration (ww:6, 14; tt:4): a fixed one

const plain = (desc) => desc.replace(/^\s*TIER\s*\d\s*-\s*\w+\s*:\s*/, '').trim();
const pname = (ps, pid) => String((ps[pid] && ps[pid].name) || `Player ${pid}`);
const aliveOf = (ps, p) => (ps[p].is_alive === undefined ? true : ps[p].is_alive);

function voteOptions(act, ps, voters) {
  const ids = Object.keys(ps).sort(byId);
  const alive = ids.filter((p) => aliveOf(ps, p));
  if (act === 1) return alive.filter((p) => ps[p].team !== 'werewolves').map((p) => pname(ps, p));
  if (act === 2 || act === 4) return alive.map((p) => pname(ps, p));
  if (act === 3) return alive.filter((p) => !voters || !voters.length || !voters.includes(p)).map((p) => pname(ps, p));
  return ['1', '2', '3'];
}

function resultText(dsl, roomState, text, deaths) {
  const ps = roomState.player_states;
  const ids = Object.keys(ps).sort(byId);
  if (ids.some((p) => 'is_alive' in ps[p])) {
    const alive = ids.filter((p) => aliveOf(ps, p));
    const wolves = alive.filter((p) => ps[p].team === 'werewolves');
    const phases = dsl.phases || {};
    const phase = phases[roomState.current_phase_id] || phases[String(roomState.current_phase_id)] || {};
    if (!phase.next_phase) {
      const side = !wolves.length ? 'Villagers win - every werewolf is eliminated.' : 'Werewolves win - they are no longer outnumbered.';
      return `${side} Survivors: ${alive.map((p) => pname(ps, p)).join(', ') || 'none'}.`;
    }
    if (deaths.length) return deaths.map((p) => `${pname(ps, p)} was eliminated.`).join(' ');
    return 'No one was eliminated.';
  }
  const sp = ids.find((p) => ps[p].is_speaker);
  if (sp && ps[sp].lie_revealed) return `The lie was statement ${ps[sp].lie_index} of ${pname(ps, sp)}.`;
  if (ids.every((p) => Number(ps[p].rounds_as_speaker || 0) > 0)) {
    const best = Math.max(...ids.map((p) => Number(ps[p].total_score || 0)));
    return 'Final scores - ' + ids.map((p) => `${pname(ps, p)}: ${ps[p].total_score || 0}`).join(', ') +
      '. Winner: ' + ids.filter((p) => Number(ps[p].total_score || 0) === best).map((p) => pname(ps, p)).join(', ') + '.';
  }
  return text;
}

/**
 * Frontend tool calls for the room's current phase, in DSL order, with every parameter the handler requires
 * (frontend_tools.json) and none it does not declare.  roomState: what RoomBatch.readRoom() / RoomService return;
 * opts.act: the phase's action kind (GameTable.info.phases[].act) or opts.table: the GameTable; opts.turn: the turn
 * just stepped (votingId); opts.deaths: ids eliminated by this turn's transition; opts.items: the frontend's canvas
 * items (clearCanvas's exemptList).  Same script as game_engine_amd/ui_script.py.
 */
function uiToolCalls(dsl, roomState, opts = {}) {
  const phases = dsl.phases || {};
  const pid = roomState.current_phase_id;
  const phase = phases[pid] || phases[String(pid)] || {};
  const phaseName = phase.name || `Phase ${pid}`;
  const ps = roomState.player_states;
  const ids = Object.keys(ps).sort(byId);
  const groups = audienceGroups(dsl, ps);
  let act = opts.act;
  if (act === undefined && opts.table) { const row = opts.table.info.phases.find((x) => x.id === pid); act = row ? row.act : 0; }
  if (act === undefined) act = 0;
  const turn = opts.turn || 0;
  const deaths = (opts.deaths || []).map(String);
  const items = opts.items;
  const deadBefore = ids.filter((p) => !aliveOf(ps, p)).length - deaths.length;
  const calls = [];
  const audience = (args, aud) => {
    if (aud === null || aud === undefined) args.audience_type = true;
    else { args.audience_type = false; args.audience_ids = aud.slice(); }
    return args;
  };
  (phase.actions || []).forEach((action, k) => {
    const desc = action.description || '';
    const text = plain(desc);
    const m = /TIER\s*(\d)/.exec(desc);
    const tier = m ? Number(m[1]) : 1;
    const perPlayer = tier >= 3 && desc.toLowerCase().includes('each player');
    let aud = tier >= 2 ? audienceFor(desc, ps, groups) : null;
    if (tier === 1 && /private|individual audience|eligible voters only/i.test(desc)) aud = audienceFor(desc, ps, groups);
    for (const tool of action.tools || []) {
      const name = `${phaseName} - ${LABEL[tool] || tool}`;
      if (tool === 'clearCanvas') {
        const args = {};
        if (items !== undefined && items !== null) {
          const d = desc.toLowerCase();
          const keep = new Set(EXEMPT.filter(([key]) => d.includes(key) && !d.includes('no exemption')).map(([, t]) => t));
          args.exemptList = items.filter((it) => keep.has(it.type)).map((it) => String(it.id));
        }
        calls.push({ name: tool, args });
      } else if (tool === 'createPhaseIndicator') {
        calls.push({ name: tool, args: audience({ name, currentPhase: phaseName, position: POSITION[tool], description: phase.description === undefined ? text : phase.description }, aud) });
      } else if (tool === 'createTextDisplay') {
        calls.push({ name: tool, args: audience({ name: `${name} ${k}`, content: text, position: POSITION[tool], title: phaseName, type: 'info' }, aud) });
      } else if (tool === 'createAvatarSet') {
        calls.push({ name: tool, args: audience({ name, avatarType: 'human' }, null) });
      } else if (tool === 'createCharacterCard') {
        const targets = perPlayer ? ids : (aud !== null && aud !== undefined ? aud : ids);
        for (const p of targets) {
          calls.push({ name: tool, args: audience({ name: `${name} ${p}`, role: ps[p].role || 'unassigned', position: POSITION[tool], description: text }, [p]) });
        }
      } else if (tool === 'createVotingPanel') {
        calls.push({ name: tool, args: audience({ name, votingId: `vote-p${pid}-t${turn}`, options: voteOptions(act, ps, aud), position: POSITION[tool], title: text }, aud) });
      } else if (tool === 'createResultDisplay') {
        calls.push({ name: tool, args: { name, content: resultText(dsl, roomState, text, deaths), position: POSITION[tool] } });
      } else if (tool === 'markPlayerDead') {
        for (const p of deaths) calls.push({ name: tool, args: { playerId: p, playerName: pname(ps, p) } });
      } else if (tool === 'createDeathMarker') {
        deaths.forEach((p, j) => calls.push({ name: tool, args: audience({ name: `${pname(ps, p)} - eliminated`, playerName: pname(ps, p), playerId: p,
                                                                          position: DEATH_POSITIONS[(deadBefore + j) % DEATH_POSITIONS.length] }, null) }));
      } else if (tool === 'createTimer') {
        calls.push({ name: tool, args: { name, duration: DISCUSSION_SECONDS, label: text } });
      } else if (tool === 'createScoreBoard') {
        const entries = ids.map((p) => ({ id: p, name: pname(ps, p), score: Number(ps[p].total_score || 0) }));
        calls.push({ name: tool, args: audience({ name, title: 'Scores', entries, sort: 'desc', position: POSITION[tool] }, null) });
      } else if (tool === 'createTurnIndicator') {
        const sp = ids.find((p) => ps[p].is_speaker) || ids[0];
        calls.push({ name: tool, args: audience({ name, currentPlayerId: sp, playerName: pname(ps, sp), label: 'Speaker', position: POSITION[tool] }, null) });
      } else if (tool === 'createStatementBoard') {
        const sp = ids.find((p) => ps[p].is_speaker);
        const st = sp ? (ps[sp].statements || {}) : {};
        const statements = sp ? [1, 2, 3].map((i) => st[String(i)] || `Statement ${i} of Player ${sp}`) : [];
        const args = { name, statements, locked: true, position: POSITION[tool] };
        if (sp && ps[sp].lie_revealed && ps[sp].lie_index) args.highlightIndex = Number(ps[sp].lie_index) - 1;
        calls.push({ name: tool, args: audience(args, null) });
      } else if (tool === 'createTextInputPanel') {                  // the handler takes no audience (page.tsx:371-386)
        calls.push({ name: tool, args: { title: phaseName, placeholder: text } });
      } else {
        calls.push({ name: tool, args: audience({ name }, aud) });
      }
    }
  });
  return calls;
}

module.exports = { compileCriteria, audienceGroups, uiToolCalls, validateCall, frontendTools };
