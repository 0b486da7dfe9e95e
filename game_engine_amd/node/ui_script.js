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

/** `player.team == 'werewolves' and player.is_alive == true` -> predicate (==, !=, in [...], and). */
function compileCriteria(expr) {
  const terms = expr.split(/\s+/).join(' ').split(/\s+and\s+/).map((part) => {
    const m = /^\s*player\.(\w+)\s*(==|!=|in)\s*(.+?)\s*$/.exec(part);
    if (!m) throw new Error(`unsupported selection criterion: ${part}`);
    const [, field, op, rhs] = m;
    if (op === 'in') {
      const inner = rhs.trim();
      if (!(inner.startsWith('[') && inner.endsWith(']'))) throw new Error(`unsupported list literal: ${rhs}`);
      return [field, op, inner.slice(1, -1).split(',').filter((x) => x.trim()).map(literal)];
    }
    return [field, op, literal(rhs)];
  });
  return (player) => terms.every(([field, op, val]) => {
    const have = player[field];
    const ok = op === 'in' ? val.includes(have) : have === val;
    return op === '!=' ? !ok : ok;
  });
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
  if (d.includes('eligible voters') || d.includes('voters')) return groups.voters || alive;
  if (d.includes('non-speaker')) return ids.filter((p) => !ps[p].is_speaker);
  if (d.includes('speaker')) return ids.filter((p) => ps[p].is_speaker);
  return null;
}

/** roomState: what RoomBatch.readRoom() returns. */
function uiToolCalls(dsl, roomState) {
  const phases = dsl.phases || {};
  const pid = roomState.current_phase_id;
  const phase = phases[pid] || phases[String(pid)] || {};
  const ps = roomState.player_states;
  const groups = audienceGroups(dsl, ps);
  const calls = [];
  for (const action of phase.actions || []) {
    const desc = action.description || '';
    const m = /TIER\s*(\d)/.exec(desc);
    const tier = m ? Number(m[1]) : 1;
    for (const tool of action.tools || []) {
      if (tool === 'clearCanvas') { calls.push({ name: tool, args: {} }); continue; }
      const base = { name: phase.name || `Phase ${pid}`, description: desc };
      if (tier >= 3 && desc.toLowerCase().includes('each player')) {
        for (const p of Object.keys(ps).sort(byId)) {
          calls.push({ name: tool, args: { ...base, audience_type: false, audience_ids: [p], role: ps[p].role || '' } });
        }
        continue;
      }
      const aud = tier >= 2 ? audienceFor(desc, ps, groups) : null;
      if (aud === null) calls.push({ name: tool, args: { ...base, audience_type: true } });
      else calls.push({ name: tool, args: { ...base, audience_type: false, audience_ids: aud } });
    }
  }
  return calls;
}

module.exports = { compileCriteria, audienceGroups, uiToolCalls };
