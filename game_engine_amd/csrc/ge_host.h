// ge_host.h - the host-only half of libge_step.so that needs no HIP: compiled DSL rows -> the kernels' table rows and literal
// image, the restart template, and the canonical room view <-> packed record conversion of ge_batch_read_rooms / write_rooms.
// Plain C++17 (ge_layout.h is host / device neutral), so this file and ge_table.cpp also build with g++ under
// AddressSanitizer + UBSan (tests/test_host_sanitizers.py) - sanitizers are CPU-only on this pool.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/ge_step.h"
#include "ge_layout.h"

namespace ge {

// v_perm_b32 selectors that gather up to 4 (Werewolf N <= 8: a byte each) or 2 (every other layout: a half-word each)
// single base predicates out of the packed predicate word pairs (W1:W0), (W3:W2) / (W2:W2), (W5:W4), 0xFF where a term lives
// elsewhere, and the XOR mask of the negated ones - the form of a row's own terms (DevRow r4..r7) and of a `conj` literal
inline void term_selectors(uint32_t kind, int n, const uint8_t *bases, const uint8_t *negs, uint32_t sel[3], uint32_t &negmask) {
    const bool bytes4 = kind == K_WW8;
    const int fpw = bytes4 ? 4 : 2, fbytes = bytes4 ? 1 : 2;
    sel[0] = sel[1] = sel[2] = 0x0D0D0D0Du;                      // selector 0x0D = the constant 0xFF
    negmask = 0;
    for (int j = 0; j < n; j++) {
        const uint32_t base = bases[j];
        const uint32_t word = base / fpw, pair = bytes4 ? (word < 2 ? 0u : 1u) : word / 2;
        const uint32_t in_pair = bytes4 ? (word < 2 ? word : 0u) : word % 2;     // N <= 8: W2 is the low word of pair 1
        const uint32_t byte0 = in_pair * 4u + (base % fpw) * fbytes;
        for (int k = 0; k < fbytes; k++) {
            const int ob = j * fbytes + k;                        // output byte
            sel[pair] = (sel[pair] & ~(0xFFu << (8 * ob))) | ((byte0 + k) << (8 * ob));
            if (negs[j]) negmask |= 0xFFu << (8 * ob);
        }
    }
}

inline DevRow to_dev_row(const ge_game_table &tb, const ge_phase_row &r, uint32_t kind) {
    // predicate masks per 32-bit word (ge_device.h): 4 bytes (werewolf N<=8) or 2 half-words (all others)
    const bool bytes4 = kind == K_WW8;
    const int fpw = bytes4 ? 4 : 2;
    const int stride = bytes4 ? 8 : 16;
    DevRow d = {0, 0, 0, 0, 0, 0, 0, 0};
    // a generic row's condition is its literal image (build_cond_image): it carries no terms, so the kernels' term path has
    // nothing to do for it (a leftover n_terms > 2 would run the slow per-term branch of ww_targets / tt_turn for nothing)
    const int n_terms = r.generic ? 0 : r.n_terms;
    d.r0 = (r.completion & 3u) | ((r.act & 7u) << 2) | ((r.effect & 7u) << 5) | (((uint32_t)n_terms & 7u) << 8) |
           ((r.n_branches & 7u) << 11) | (r.generic ? ROW_GENERIC : 0u);
    for (int j = 0; j < GE_MAX_TERMS; j++) {
        if (j < n_terms) d.r0 |= (uint32_t)(r.term_neg[j] & 1u) << (16 + j);
        const uint32_t base = r.term_base[j];
        const uint32_t enc = j < n_terms ? ((((base / fpw) & 7u) << 5) | (((base % fpw) * stride) & 31u)) : (7u << 5);
        d.r1 |= enc << (8 * j);
    }
    for (int b = 0; b < GE_MAX_BRANCHES; b++) {
        if (b < r.n_branches) d.r2 |= (1u << (r.br_res[b] & 7u)) << (8 * b);
        // target row index, and the target's entry effect (so the effect can start before the row load returns)
        const uint32_t tgt = r.br_target[b] & 31u;
        d.r3 |= (tgt | ((uint32_t)(tb.rows[tgt].effect & 7u) << 5)) << (8 * b);
        if (b < r.n_branches && r.br_res[b] == GE_RES_ALL_ROUNDS_DONE) d.r0 |= 1u << 20;
    }
    {
        // the first 4 (Werewolf N <= 8) / 2 terms through the permute path
        uint32_t sel[3], nm;
        term_selectors(kind, std::min(n_terms, bytes4 ? 4 : 2), r.term_base, r.term_neg, sel, nm);
        d.r4 = sel[0]; d.r5 = sel[1]; d.r6 = sel[2]; d.r7 = nm;
    }
    return d;
}

inline DevCond to_dev_cond(const ge_phase_row &r) {                       // the clause form as ge_inject_kernel reads it
    DevCond c;
    memset(&c, 0, sizeof c);
    const uint32_t ncl = r.n_clauses <= GE_MAX_CLAUSES ? r.n_clauses : GE_MAX_CLAUSES;
    c.meta = ncl;
    for (uint32_t k = 0; k < ncl; k++) {
        const uint32_t len = r.clause_len[k] <= GE_MAX_TERMS ? r.clause_len[k] : GE_MAX_TERMS;
        c.meta |= len << (4 + 4 * k);
        for (uint32_t l = 0; l < len; l++) {
            const ge_literal &x = r.clause[k][l];
            const uint32_t payload = x.kind == GE_LIT_NUM ? ((uint32_t)x.lo | ((uint32_t)x.hi << 8)) : x.bases;
            c.lit[k][l] = payload | ((uint32_t)(x.num_field & 7u) << 16) | ((uint32_t)(x.kind & 3u) << 28) | (x.neg ? 1u << 30 : 0u);
        }
    }
    return c;
}

// The generic rows of a table as the step kernels evaluate them: the literal image a block copies into its LDS
// (ge_layout.h CondLit / CondLit12, ge_device.h eval_cond_image).  Every generic row gets a slot and is padded to the
// table's common shape with neutral literals; dt.rows[] must already hold the rows (the slot number goes into r0).
// Inside a clause the literals that test ONE base predicate each ("role == 'Doctor'", "is_alive != false": most of what a
// generated condition consists of) are merged into `conj` literals - up to 4 (2) terms answered by one permute-and-fold,
// like a shipped row's own terms - so a typical clause is one conj literal, or one and a numeric range.
inline void build_cond_image(const ge_game_table &tb, uint32_t kind, DevTable &dt) {
    struct Lit { int what; ge_literal x; int n; uint8_t bases[4], negs[4]; };          // what: 0 base set, 1 numeric range, 2 conj, 3 constant FALSE (x.neg: TRUE)
    const bool ww8 = kind == K_WW8, ww12 = kind == K_WW12;
    const int cap = ww8 ? 4 : 2;                                                        // terms per conj literal
    std::vector<std::vector<std::vector<Lit>>> rows;                                    // generic row -> clause -> literal
    std::vector<int> row_of;
    uint32_t ncl = 0, len = 0;
    for (int r = 0; r < tb.n_phases; r++) {
        const ge_phase_row &pr = tb.rows[r];
        if (!pr.generic) continue;
        const uint32_t rc = pr.n_clauses <= GE_MAX_CLAUSES ? pr.n_clauses : GE_MAX_CLAUSES;
        std::vector<std::vector<Lit>> clauses;
        for (uint32_t k = 0; k < rc; k++) {
            const uint32_t rl = pr.clause_len[k] <= GE_MAX_TERMS ? pr.clause_len[k] : GE_MAX_TERMS;
            std::vector<Lit> singles, others;
            for (uint32_t l = 0; l < rl; l++) {
                Lit t;
                memset(&t, 0, sizeof t);
                t.x = pr.clause[k][l];
                const bool nibbles = t.x.num_field == GE_NUM_SELECTED_TARGET || t.x.num_field == GE_NUM_ROUNDS_AS_SPEAKER;
                if (t.x.kind == GE_LIT_NUM) {
                    t.what = (t.x.lo > t.x.hi || (nibbles && t.x.lo > 15)) ? 3 : 1;                // an empty range: the constant FALSE (negated: TRUE)
                    others.push_back(t);
                } else if (__builtin_popcount(t.x.bases) == 1) {
                    t.what = 2; t.n = 1; t.bases[0] = (uint8_t)__builtin_ctz(t.x.bases); t.negs[0] = t.x.neg ? 1 : 0;
                    singles.push_back(t);
                } else {
                    t.what = t.x.bases ? 0 : 3;
                    others.push_back(t);
                }
            }
            // every single-predicate literal goes into a conj, `cap` terms per literal - a lone one too: a slot that holds one
            // kind of literal in every row evaluates one form, and a conj costs what a base set does
            std::vector<Lit> out;
            for (size_t i = 0; i < singles.size(); i += cap) {
                Lit c;
                memset(&c, 0, sizeof c);
                c.what = 2;
                for (size_t j = i; j < singles.size() && j < i + cap; j++) { c.bases[c.n] = singles[j].bases[0]; c.negs[c.n] = singles[j].negs[0]; c.n++; }
                out.push_back(c);
            }
            for (const Lit &t : others) out.push_back(t);
            len = std::max<uint32_t>(len, (uint32_t)out.size());
            clauses.push_back(out);
        }
        ncl = std::max<uint32_t>(ncl, std::max<uint32_t>((uint32_t)clauses.size(), 1u));          // no clause at all = everybody: one clause of TRUE literals
        rows.push_back(clauses);
        row_of.push_back(r);
    }
    dt.cond_shape = 0; dt.cond_g[0] = dt.cond_g[1] = 0; dt.cond_fields[0] = dt.cond_fields[1] = 0; dt.cond_n16 = 0;
    if (rows.empty()) return;
    len = std::max<uint32_t>(len, 1u);
    dt.cond_shape = ncl | (len << 4);
    // per slot, over all rows: which kinds (g: 1 base set, 2 numeric, 4 clause end, 8 conj) and which numeric fields (f)
    uint32_t g_of[16] = {0}, f_of[16] = {0};
    for (const auto &clauses : rows)
        for (size_t k = 0; k < clauses.size(); k++)
            for (size_t l = 0; l < clauses[k].size(); l++) {
                const Lit &t = clauses[k][l];
                const uint32_t i = (uint32_t)(k * len + l);
                if (t.what == 0) g_of[i] |= 1u;
                if (t.what == 2) g_of[i] |= 8u;
                if (t.what == 1) { g_of[i] |= 2u; if (t.x.num_field >= 1 && t.x.num_field <= 4) f_of[i] |= 1u << (t.x.num_field - 1u); }
            }
    {
        uint64_t G = 0, F = 0;                                                                     // a nibble per slot, in the evaluator's order
        for (uint32_t i = 0; i < ncl * len; i++) {
            G |= (uint64_t)(g_of[i] | ((i + 1u) % len == 0u ? 4u : 0u)) << (4u * i);
            F |= (uint64_t)f_of[i] << (4u * i);
        }
        dt.cond_g[0] = (uint32_t)G; dt.cond_g[1] = (uint32_t)(G >> 32);
        dt.cond_fields[0] = (uint32_t)F; dt.cond_fields[1] = (uint32_t)(F >> 32);
    }
    const uint32_t stride = ww12 ? sizeof(CondLit12) : sizeof(CondLit);                            // bytes per literal
    uint32_t *img = reinterpret_cast<uint32_t *>(dt.cond_img);
    for (size_t slot_no = 0; slot_no < rows.size(); slot_no++) {
        const auto &clauses = rows[slot_no];
        dt.rows[row_of[slot_no]].r0 |= (uint32_t)slot_no << ROW_COND_SLOT_SHIFT;
        for (uint32_t k = 0; k < ncl; k++)
            for (uint32_t l = 0; l < len; l++) {
                const uint32_t i = k * len + l;
                uint32_t *d = img + ((size_t)slot_no * ncl * len + i) * (stride / 4u);
                memset(d, 0, stride);
                // a neutral literal in the cheapest kind the slot evaluates anyway: a base set with no field (FALSE; negated:
                // TRUE), or - in a slot that holds conj literals but no base sets - a conj of no terms (TRUE; one negated
                // constant byte: FALSE)
                auto constant = [&](bool value) {
                    if ((g_of[i] & 8u) && !(g_of[i] & 1u)) {
                        d[0] = 0x100u; d[1] = d[2] = 0x0D0D0D0Du;
                        if (ww12) { d[3] = 0x0D0D0D0Du; d[4] = value ? 0u : 0xFFFFFFFFu; } else d[3] = value ? 0u : 0xFFFFFFFFu;
                    } else {
                        d[0] = value ? 0xFFFF0000u : 0u;
                    }
                };
                const bool clause_used = clauses.empty() ? k == 0 : k < clauses.size();
                if (!clause_used) { constant(false); continue; }
                if (clauses.empty() || l >= clauses[k].size()) { constant(true); continue; }
                const Lit &t = clauses[k][l];
                const ge_literal &x = t.x;
                const uint32_t neg = x.neg ? 0xFFFF0000u : 0u;
                if (t.what == 3) { constant(x.neg != 0); continue; }
                if (t.what == 2) {                                                                 // conj: selectors + XOR mask, like a row's own terms
                    uint32_t sel[3], nm;
                    term_selectors(kind, t.n, t.bases, t.negs, sel, nm);
                    d[0] = 0x100u;
                    if (ww12) { d[1] = sel[0]; d[2] = sel[1]; d[3] = sel[2]; d[4] = nm; }
                    else { d[1] = sel[0]; d[2] = sel[1]; d[3] = nm; }
                    continue;
                }
                if (t.what == 1) {
                    const uint32_t f = x.num_field & 7u;
                    d[0] = 1u | (f << 1) | (f ? 1u << (3u + f) : 0u) | neg;
                    if (f == GE_NUM_LIE_INDEX || f == GE_NUM_VOTE_CHOICE) {              // 2-bit fields: the allowed values as masks
                        auto allowed = [&](uint32_t v) { return v >= x.lo && v <= x.hi ? 0x00555555u : 0u; };
                        d[1] = allowed(0) | (allowed(1) << 1);
                        d[2] = allowed(2) | (allowed(3) << 1);
                    } else if (f == GE_NUM_TOTAL_SCORE) {                                // bytes compared in half-word lanes
                        d[1] = (uint32_t)x.lo * 0x00010001u;
                        d[2] = ((uint32_t)x.hi * 0x00010001u) | 0x80008000u;
                    } else {                                                             // nibble arrays (selected_target_id, rounds_as_speaker) in byte lanes
                        d[1] = (uint32_t)x.lo * 0x01010101u;
                        d[2] = (std::min<uint32_t>(x.hi, 15u) * 0x01010101u) | 0x80808080u;
                    }
                    continue;
                }
                d[0] = neg;                                                              // a base set: AND-masks over the packed predicate words
                for (uint32_t b = 0; b < 16u; b++) {
                    if (!((x.bases >> b) & 1u)) continue;
                    if (ww8) { if (b < 12u) d[1 + b / 4u] |= 0xFFu << (8u * (b % 4u)); }
                    else if (ww12) { if (b < 12u) d[1 + b / 2u] |= 0xFFFFu << (16u * (b % 2u)); }
                    else if (b < 5u) d[1 + b / 2u] |= 0xFFFFu << (16u * (b % 2u));       // speaker | submitted << 16, revealed | can_vote << 16, has_voted
                }
            }
    }
    dt.cond_n16 = (uint32_t)rows.size() * ncl * len * stride / 16u;
}

// the initial record in the kernels' register form (SegDev::init_regs)
inline void init_regs_of(uint32_t kind, const uint32_t *w, uint32_t *regs) {
    switch (kind) {
    case K_WW8: { WWR<8> s; uint32_t c; ww_load_regs<8>(w, s, c); s.to_regs(regs); break; }
    case K_WW12: { WWR<12> s; uint32_t c; ww_load_regs<12>(w, s, c); s.to_regs(regs); break; }
    case K_TT4: { TT<4> s; TTLayout<4>::unpack(w, s); s.to_regs(regs); break; }
    case K_TT8: { TT<8> s; TTLayout<8>::unpack(w, s); s.to_regs(regs); break; }
    default: { TT<12> s; TTLayout<12>::unpack(w, s); s.to_regs(regs); break; }
    }
}

inline int words_of(uint32_t kind) {
    switch (kind) {
    case K_WW8: return 8; case K_WW12: return 10; case K_TT4: return 6; case K_TT8: return 8; default: return 12;
    }
}

// canonical view <-> packed words (host side of ge_batch_read_rooms / write_rooms)
template <int NB> inline void view_to_ww(const ge_room_view &v, const ge_game_table &tb, uint32_t *w) {
    WW<NB> s;
    memset(&s, 0, sizeof s);
    for (int i = 0; i < v.n_players; i++) {
        const uint8_t *f = v.players[i];
        const uint32_t b = 1u << i;
        auto put = [b](bool on, uint32_t &mask) { mask |= on ? b : 0u; };
        put(f[0] & 1, s.rb0); put(f[0] & 2, s.rb1); put(f[0] & 4, s.rb2);
        put(f[1] == 1, s.team_v); put(f[1] == 2, s.team_w);
        put(f[2], s.alive); put(f[3], s.revealed); put(f[4], s.can_vote); put(f[5], s.secret);
        put(f[6], s.elig); put(f[7], s.sub);
        s.sel |= (typename WW<NB>::nib_t)(f[8] & 15) << (4 * i);
        put(f[9], s.acted);
        s.choice |= (typename WW<NB>::nib_t)(f[10] & 15) << (4 * i);
        put(v.det[i] == 1, s.det_v); put(v.det[i] == 2, s.det_w);
    }
    int pi = 0, qi = 0;
    for (int k = 0; k < tb.n_phases; k++) {
        if (tb.rows[k].phase_id == v.phase_id) pi = k;
        if (tb.rows[k].phase_id == v.prev_phase_id) qi = k;
    }
    s.phase = pi; s.prev = qi;
    s.flags = (v.phase0_done ? FLAG_PHASE0_DONE : 0) | ((uint32_t)tb.rows[qi].effect << 1);
    s.end_turn = v.end_turn < 0 ? END_NONE : (uint32_t)v.end_turn;
    s.games = (uint32_t)v.games & 0xFFFFu;
    WWLayout<NB>::pack(s, w);
}

template <int NB> inline void ww_to_view(const uint32_t *w, const ge_game_table &tb, int n, ge_room_view &v) {
    WW<NB> s;
    WWLayout<NB>::unpack(w, s);
    memset(&v, 0, sizeof v);
    v.pack = GE_PACK_WEREWOLF; v.n_players = (uint8_t)n;
    v.phase_id = tb.rows[s.phase].phase_id; v.prev_phase_id = tb.rows[s.prev].phase_id;
    v.phase0_done = s.flags & FLAG_PHASE0_DONE;
    v.end_turn = s.end_turn == END_NONE ? -1 : (int32_t)s.end_turn;
    v.games = (int32_t)s.games;
    for (int i = 0; i < n; i++) {
        uint8_t *f = v.players[i];
        f[0] = (uint8_t)(((s.rb0 >> i) & 1) | (((s.rb1 >> i) & 1) << 1) | (((s.rb2 >> i) & 1) << 2));
        f[1] = (uint8_t)(((s.team_v >> i) & 1) ? 1 : (((s.team_w >> i) & 1) ? 2 : 0));
        f[2] = (s.alive >> i) & 1; f[3] = (s.revealed >> i) & 1; f[4] = (s.can_vote >> i) & 1;
        f[5] = (s.secret >> i) & 1; f[6] = (s.elig >> i) & 1; f[7] = (s.sub >> i) & 1;
        f[8] = (uint8_t)((s.sel >> (4 * i)) & 15); f[9] = (s.acted >> i) & 1;
        f[10] = (uint8_t)((s.choice >> (4 * i)) & 15);
        v.det[i] = (uint8_t)(((s.det_v >> i) & 1) ? 1 : (((s.det_w >> i) & 1) ? 2 : 0));
    }
}

template <int NB> inline void view_to_tt(const ge_room_view &v, const ge_game_table &tb, uint32_t *w) {
    TT<NB> s;
    memset(&s, 0, sizeof s);
    for (int i = 0; i < v.n_players; i++) {
        const uint8_t *f = v.players[i];
        const uint32_t b = 1u << i;
        auto put = [b](bool on, uint32_t &mask) { mask |= on ? b : 0u; };
        put(f[0], s.speaker); put(f[1], s.submitted); put(f[3], s.revealed);
        put(f[4], s.can_vote); put(f[6], s.has_voted); put(f[9], s.acted);
        s.lie |= (uint32_t)(f[2] & 3) << (2 * i); s.vote |= (uint32_t)(f[5] & 3) << (2 * i);
        s.choice |= (uint32_t)(f[10] & 3) << (2 * i);
        s.score[i / 4] |= (uint32_t)f[7] << (8 * (i % 4));
        s.rounds |= (uint64_t)(f[8] & 15) << (4 * i);
    }
    int pi = 0, qi = 0;
    for (int k = 0; k < tb.n_phases; k++) {
        if (tb.rows[k].phase_id == v.phase_id) pi = k;
        if (tb.rows[k].phase_id == v.prev_phase_id) qi = k;
    }
    s.phase = pi; s.prev = qi;
    s.flags = (v.phase0_done ? FLAG_PHASE0_DONE : 0) | ((uint32_t)tb.rows[qi].effect << 1);
    s.end_turn = v.end_turn < 0 ? END_NONE : (uint32_t)v.end_turn;
    s.games = (uint32_t)v.games & 0xFFFFu;
    TTLayout<NB>::pack(s, w);
}

template <int NB> inline void tt_to_view(const uint32_t *w, const ge_game_table &tb, int n, ge_room_view &v) {
    TT<NB> s;
    TTLayout<NB>::unpack(w, s);
    memset(&v, 0, sizeof v);
    v.pack = GE_PACK_TWO_TRUTHS; v.n_players = (uint8_t)n;
    v.phase_id = tb.rows[s.phase].phase_id; v.prev_phase_id = tb.rows[s.prev].phase_id;
    v.phase0_done = s.flags & FLAG_PHASE0_DONE;
    v.end_turn = s.end_turn == END_NONE ? -1 : (int32_t)s.end_turn;
    v.games = (int32_t)s.games;
    for (int i = 0; i < n; i++) {
        uint8_t *f = v.players[i];
        f[0] = (s.speaker >> i) & 1; f[1] = (s.submitted >> i) & 1; f[2] = (s.lie >> (2 * i)) & 3;
        f[3] = (s.revealed >> i) & 1; f[4] = (s.can_vote >> i) & 1; f[5] = (s.vote >> (2 * i)) & 3;
        f[6] = (s.has_voted >> i) & 1; f[7] = (uint8_t)((s.score[i / 4] >> (8 * (i % 4))) & 255);
        f[8] = (uint8_t)((s.rounds >> (4 * i)) & 15); f[9] = (s.acted >> i) & 1; f[10] = (s.choice >> (2 * i)) & 3;
    }
}

inline void view_to_words(uint32_t kind, const ge_room_view &v, const ge_game_table &tb, uint32_t *w) {
    switch (kind) {
    case K_WW8: view_to_ww<8>(v, tb, w); break;
    case K_WW12: view_to_ww<12>(v, tb, w); break;
    case K_TT4: view_to_tt<4>(v, tb, w); break;
    case K_TT8: view_to_tt<8>(v, tb, w); break;
    default: view_to_tt<12>(v, tb, w); break;
    }
}

inline void words_to_view(uint32_t kind, const uint32_t *w, const ge_game_table &tb, int n, ge_room_view &v) {
    switch (kind) {
    case K_WW8: ww_to_view<8>(w, tb, n, v); break;
    case K_WW12: ww_to_view<12>(w, tb, n, v); break;
    case K_TT4: tt_to_view<4>(w, tb, n, v); break;
    case K_TT8: tt_to_view<8>(w, tb, n, v); break;
    default: tt_to_view<12>(w, tb, n, v); break;
    }
}


// what ge_batch_write_rooms checks of every view before anything is written: the view belongs to the segment (player count,
// rule pack), both phase ids name rows of the segment's table (the reference never stores an id outside dsl['phases']
// either: agent/game_agent_v2.py:1173-1191), werewolf role classes are 0 (unassigned) .. 4
inline bool view_fits(const ge_room_view &v, const ge_game_table &tb, uint32_t n_players) {
    auto known_phase = [&tb](int32_t id) {
        for (int k = 0; k < tb.n_phases; k++) if (tb.rows[k].phase_id == id) return true;
        return false;
    };
    bool ok = v.n_players == n_players && v.pack == (uint8_t)tb.pack && known_phase(v.phase_id) && known_phase(v.prev_phase_id);
    if (ok && v.pack == GE_PACK_WEREWOLF)
        for (uint32_t i = 0; i < n_players && i < 16u; i++) ok &= v.players[i][0] <= 4;
    return ok;
}


// The sharding of a device group (ge_group_create) and of any host that splits a job by hand: part i of n takes the i-th of n
// contiguous parts of EVERY segment of `desc` (the whole job), and seg_first[k] is the GLOBAL index of that part's first room of
// segment k - the index the room has in one batch of `desc`, which is what the RNG is keyed by (POLICY.md 2), so results do not
// depend on n.  Rooms never interact in the reference (one LangGraph thread each, src/app/api/copilotkit/route.ts:24-37), so
// this arithmetic is all there is to multi-GPU stepping.  HIP-free: also built with g++ under ASan + UBSan (tests/native).
inline int group_partition(const ge_batch_desc &desc, int n, int i, ge_batch_desc *shard, uint64_t *seg_first) {
    if (n < 1 || i < 0 || i >= n || !shard || !seg_first || desc.n_segments == 0 || desc.n_segments > GE_MAX_SEGMENTS) return GE_ERR_ARG;
    uint64_t acc = desc.first_room;
    ge_batch_desc d = desc;
    for (uint32_t k = 0; k < desc.n_segments; k++) {
        const uint64_t R = desc.seg[k].n_rooms;
        if (R < (uint64_t)n) return GE_ERR_ARG;                       // every part holds rooms of every segment
        // R * i can pass 2^64 for absurd R only (R < 2^57 at n <= 64 is safe); split the product to stay exact anyway
        const uint64_t q = R / (uint64_t)n, r = R % (uint64_t)n;
        const uint64_t lo = q * (uint64_t)i + r * (uint64_t)i / (uint64_t)n;
        const uint64_t hi = q * (uint64_t)(i + 1) + r * (uint64_t)(i + 1) / (uint64_t)n;
        d.seg[k].n_rooms = hi - lo;
        seg_first[k] = acc + lo;
        acc += R;
    }
    for (uint32_t k = desc.n_segments; k < GE_MAX_SEGMENTS; k++) seg_first[k] = 0;
    *shard = d;
    return GE_OK;
}

}  // namespace ge
