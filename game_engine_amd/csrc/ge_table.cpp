// ge_table.cpp — game DSL (as JSON text) -> ge_game_table.
//
// The reference never compiles its DSL: each turn it pastes dsl['phases'][id] into an LLM
// prompt (agent/game_agent_v2.py:1057, 1087-1103) and lets the model read completion
// criteria, target conditions and the natural-language next_phase keys.  This file is the
// deterministic replacement: it reads the same document (games/*.yaml after the host's YAML
// loader, handed over as JSON so that Python and Node hosts share one implementation) and
// emits the table the kernels interpret.  Classification rules are POLICY.md §Table.
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ge_step.h"

namespace {

// ------------------------------------------------------------------ minimal ordered JSON
struct JVal;
using JPtr = std::unique_ptr<JVal>;
struct JVal {
    enum Type { NUL, BOOL, NUM, STR, ARR, OBJ } type = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;   // insertion order = DSL order

    const JVal *get(const char *key) const {
        if (type != OBJ) return nullptr;
        for (auto &kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    bool is_null() const { return type == NUL; }
};

struct Parser {
    const char *p, *end;
    std::string err;
    int depth = 0;

    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }

    static void utf8(std::string &s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t &v) {
        if (end - p < 4) return fail("bad \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                char c = *p++;
                switch (c) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': {
                    uint32_t v, lo;
                    if (!hex4(v)) return false;
                    if (v >= 0xD800 && v < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2;
                        if (!hex4(lo)) return false;
                        v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, v);
                    break;
                }
                default: out += c; break;
                }
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(JPtr &out) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        out.reset(new JVal());
        if (p >= end) return fail("unexpected end");
        bool ok = true;
        if (*p == '{') {
            out->type = JVal::OBJ;
            p++; ws();
            if (p < end && *p == '}') p++;
            else for (;;) {
                ws();
                std::string k;
                if (!string(k)) { ok = false; break; }
                ws();
                if (p >= end || *p++ != ':') { ok = fail("expected ':'"); break; }
                JPtr v;
                if (!value(v)) { ok = false; break; }
                out->obj.emplace_back(std::move(k), std::move(v));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                ok = fail("expected ',' or '}'"); break;
            }
        } else if (*p == '[') {
            out->type = JVal::ARR;
            p++; ws();
            if (p < end && *p == ']') p++;
            else for (;;) {
                JPtr v;
                if (!value(v)) { ok = false; break; }
                out->arr.push_back(std::move(v));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                ok = fail("expected ',' or ']'"); break;
            }
        } else if (*p == '"') {
            out->type = JVal::STR;
            ok = string(out->str);
        } else if (end - p >= 4 && !strncmp(p, "true", 4)) { out->type = JVal::BOOL; out->b = true; p += 4; }
        else if (end - p >= 5 && !strncmp(p, "false", 5)) { out->type = JVal::BOOL; out->b = false; p += 5; }
        else if (end - p >= 4 && !strncmp(p, "null", 4)) { out->type = JVal::NUL; p += 4; }
        else {
            char *e = nullptr;
            std::string tmp(p, (size_t)((end - p) < 40 ? (end - p) : 40));
            out->type = JVal::NUM;
            out->num = strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) ok = fail("unexpected character");
            else p += e - tmp.c_str();
        }
        depth--;
        return ok;
    }
};

std::string lower(const std::string &s) {
    std::string o = s;
    for (auto &c : o) c = (char)tolower((unsigned char)c);
    return o;
}
bool has(const std::string &hay, const char *needle) { return hay.find(needle) != std::string::npos; }

void copy_name(char *dst, const std::string &s) {
    size_t n = s.size() < GE_NAME_LEN - 1 ? s.size() : GE_NAME_LEN - 1;
    while (n > 0 && ((unsigned char)s[n] & 0xC0) == 0x80) n--;      // do not cut a UTF-8 sequence
    memcpy(dst, s.data(), n);
    dst[n] = 0;
}

enum { ROLE_NONE, ROLE_VILLAGER, ROLE_WEREWOLF, ROLE_DOCTOR, ROLE_DETECTIVE };

int role_class(const std::string &name) {
    std::string n = lower(name);
    if (has(n, "wolf") || has(n, "mafia")) return ROLE_WEREWOLF;
    if (has(n, "doctor") || has(n, "medic")) return ROLE_DOCTOR;
    if (has(n, "detective") || has(n, "seer")) return ROLE_DETECTIVE;
    return ROLE_VILLAGER;
}

struct Err {
    char *buf; size_t cap;
    int set(const std::string &m) const {
        if (buf && cap) { snprintf(buf, cap, "%s", m.c_str()); }
        return GE_ERR_DSL;
    }
};

bool truthy(const JVal *v) {
    if (!v) return false;
    switch (v->type) {
    case JVal::BOOL: return v->b;
    case JVal::NUM: return v->num != 0;
    case JVal::STR: return !v->str.empty();
    case JVal::ARR: return !v->arr.empty();
    case JVal::OBJ: return !v->obj.empty();
    default: return false;
    }
}
int as_int(const JVal *v) { return v && v->type == JVal::NUM ? (int)v->num : 0; }
std::string as_str(const JVal *v) { return v && v->type == JVal::STR ? v->str : std::string(); }

// The packs' state slots and the declared names each binds to (first = canonical); include/ge_step.h GE_WW_* / GE_TT_*.
const char *const WW_SLOT_NAMES[GE_WW_SLOTS][3] = {
    {"role"}, {"team"}, {"is_alive"}, {"role_revealed"}, {"can_vote"}, {"has_secret_role"},
    {"night_action_eligible", "has_night_action"}, {"night_action_submitted"}, {"selected_target_id"},
    {"investigated_alignments", "known_alignments"}, {"wolf_chat_enabled"}};
const char *const TT_SLOT_NAMES[GE_TT_SLOTS][3] = {
    {"is_speaker"}, {"statements_submitted"}, {"lie_index"}, {"lie_revealed"}, {"can_vote"}, {"vote_choice"},
    {"has_voted"}, {"total_score"}, {"rounds_as_speaker"}, {"statements"}};

// which declared field each slot of the pack is (declaration.player_states decides; "" = not declared)
struct Binding {
    int pack = 0, n = 0;
    std::string declared[GE_MAX_SLOTS];
    // every string of the phase graph except the target conditions themselves (names, descriptions, action texts, branch
    // keys): what the Referee is told to do.  A field these texts name may be written during play (mentions()).
    std::string phase_text;
    void collect_text(const JVal *v, const std::string &key) {
        if (!v) return;
        if (v->type == JVal::OBJ) for (auto &kv : v->obj) { phase_text += kv.first; phase_text += '\n'; collect_text(kv.second.get(), kv.first); }
        else if (v->type == JVal::ARR) for (auto &x : v->arr) collect_text(x.get(), key);
        else if (v->type == JVal::STR && key != "condition") { phase_text += v->str; phase_text += '\n'; }
    }
    // whole-identifier, case-sensitive occurrence ("TIER 1 - PUBLIC" does not name a field `tier`)
    bool mentions(const std::string &field) const {
        auto ident = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; };
        if (field.empty()) return false;
        for (size_t at = phase_text.find(field); at != std::string::npos; at = phase_text.find(field, at + 1)) {
            const bool left = at == 0 || !ident(phase_text[at - 1]);
            const bool right = at + field.size() >= phase_text.size() || !ident(phase_text[at + field.size()]);
            if (left && right) return true;
        }
        return false;
    }
    const char *canonical(int slot) const { return pack == GE_PACK_WEREWOLF ? WW_SLOT_NAMES[slot][0] : TT_SLOT_NAMES[slot][0]; }
    // false: the declaration names one slot twice (two accepted names of the same slot) - `twice` says which
    bool bind(int pack_, const JVal *ps_def, std::string &twice) {
        pack = pack_;
        n = pack == GE_PACK_WEREWOLF ? (int)GE_WW_SLOTS : (int)GE_TT_SLOTS;
        for (int s = 0; s < n; s++) {
            const char *const *names = pack == GE_PACK_WEREWOLF ? WW_SLOT_NAMES[s] : TT_SLOT_NAMES[s];
            for (int k = 0; k < 3 && names[k]; k++)
                if (ps_def && ps_def->get(names[k])) {
                    if (!declared[s].empty()) { twice = declared[s] + " / " + names[k]; return false; }
                    declared[s] = names[k];
                }
        }
        return true;
    }
    // canonical slot name of a declared field, "" if the field is not a slot of the pack
    std::string slot_of(const std::string &field) const {
        for (int s = 0; s < n; s++) if (!declared[s].empty() && declared[s] == field) return canonical(s);
        return std::string();
    }
    const JVal *get(const JVal *obj, int slot) const { return obj && !declared[slot].empty() ? obj->get(declared[slot].c_str()) : nullptr; }
};

// base predicate index of `player.<slot> == <value>` inside a pack (canonical slot names), -1 if the pack has none
int base_of(int pack, const std::string &field, const std::string &sval, bool is_str) {
    if (pack == GE_PACK_WEREWOLF) {
        static const char *bools[] = {"is_alive", "can_vote", "role_revealed", "has_secret_role",
                                      "night_action_eligible", "night_action_submitted"};
        if (!is_str) {
            for (int i = 0; i < 6; i++) if (field == bools[i]) return i;
            if (field == "wolf_chat_enabled") return 7;             // derived slot: set with the team, never again
            return -1;
        }
        if (field == "team") { if (sval == "villagers") return 6; if (sval == "werewolves") return 7; return -1; }
        if (field == "role") return 7 + role_class(sval);
        return -1;
    }
    static const char *bools[] = {"is_speaker", "statements_submitted", "lie_revealed", "can_vote", "has_voted"};
    if (is_str) return -1;
    for (int i = 0; i < 5; i++) if (field == bools[i]) return i;
    return -1;
}

// ---- target_players.condition: the grammar of dsl_phases_generation_prompt.txt:120-132
//   cond   := clause { " or " clause }            (and binds tighter than or; no parentheses)
//   clause := term { " and " term }
//   term   := "player." field op value | "player." field ["not"] "in" "[" value {"," value} "]"
//   op     := == != < <= > >=          value := 'str' | "str" | true | false | integer
// compiled to an OR of AND-clauses of ge_literal.  Both shipped games only write `==` joined by `and`.

struct Atom { enum { BOOL, INT, STR } type = INT; bool b = false; long num = 0; std::string str; };

std::string squeeze(const std::string &s) {                    // whitespace runs -> one blank, trimmed
    std::string o;
    bool sp = true;
    for (char c : s) {
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { if (!sp) o += ' '; sp = true; }
        else { o += c; sp = false; }
    }
    while (!o.empty() && o.back() == ' ') o.pop_back();
    return o;
}

// splits on a blank-delimited keyword outside quotes and brackets
std::vector<std::string> split_kw(const std::string &s, const char *kw) {
    std::vector<std::string> out;
    const std::string pat = std::string(" ") + kw + " ";
    std::string low = lower(s);
    size_t start = 0;
    char quote = 0;
    int depth = 0;
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[i];
        if (quote) { if (c == quote) quote = 0; continue; }
        if (c == '\'' || c == '"') { quote = c; continue; }
        if (c == '[') depth++;
        else if (c == ']') depth--;
        else if (depth == 0 && low.compare(i, pat.size(), pat) == 0) {
            out.push_back(s.substr(start, i - start));
            start = i + pat.size();
            i = start - 1;
        }
    }
    out.push_back(s.substr(start));
    return out;
}

bool parse_atom(const std::string &text, Atom &a) {
    std::string t = squeeze(text);
    std::string l = lower(t);
    if (l == "true" || l == "false") { a.type = Atom::BOOL; a.b = l == "true"; return true; }
    if (t.size() >= 2 && (t[0] == '\'' || t[0] == '"') && t.back() == t[0]) { a.type = Atom::STR; a.str = t.substr(1, t.size() - 2); return true; }
    if (!t.empty()) {
        size_t i = t[0] == '-' ? 1 : 0;
        bool digits = i < t.size();
        for (size_t k = i; k < t.size(); k++) digits = digits && isdigit((unsigned char)t[k]);
        if (digits) { a.type = Atom::INT; a.num = strtol(t.c_str(), nullptr, 10); return true; }
    }
    return false;
}

// numeric field of a pack: GE_NUM_* and the largest value the record holds, -1 if the pack has none of that name
int num_field_of(int pack, const std::string &field, int &top) {
    if (pack == GE_PACK_WEREWOLF) {
        if (field == "selected_target_id") { top = 15; return GE_NUM_SELECTED_TARGET; }
        return -1;
    }
    if (field == "lie_index") { top = 3; return GE_NUM_LIE_INDEX; }
    if (field == "vote_choice") { top = 3; return GE_NUM_VOTE_CHOICE; }
    if (field == "total_score") { top = 255; return GE_NUM_TOTAL_SCORE; }
    if (field == "rounds_as_speaker") { top = 15; return GE_NUM_ROUNDS_AS_SPEAKER; }
    return -1;
}

bool is_bool_field(int pack, const std::string &field) { return base_of(pack, field, std::string(), false) >= 0; }

using Clause = std::vector<ge_literal>;

// is `field` one the rule pack models (a base predicate, an enum or a numeric field of its records)?
bool pack_models(int pack, const std::string &field) {
    int top;
    return is_bool_field(pack, field) || num_field_of(pack, field, top) >= 0 ||
           (pack == GE_PACK_WEREWOLF && (field == "role" || field == "team"));
}

// a term over a declared field the pack does not model: nobody ever writes it under the fixed policy, so every player
// keeps the template's value and the term is a constant.  1 = holds, 0 = does not, -1 = outside the grammar
int const_term(const JVal &have, const std::string &op, const std::vector<Atom> &vals) {
    auto same = [&](const Atom &v) {
        if (have.type == JVal::BOOL) return v.type == Atom::BOOL ? v.b == have.b : (v.type == Atom::INT && (v.num == 0 || v.num == 1) && (v.num != 0) == have.b);
        if (have.type == JVal::NUM) {
            if (v.type == Atom::BOOL) return (have.num == 0 || have.num == 1) && (have.num != 0) == v.b;
            return v.type == Atom::INT && (double)v.num == have.num;
        }
        return v.type == Atom::STR && v.str == have.str;
    };
    if (op == "==" || op == "!=" || op == "in" || op == "not in") {
        bool hit = false;
        for (auto &v : vals) hit = hit || same(v);
        return hit != (op == "!=" || op == "not in") ? 1 : 0;
    }
    if (have.type != JVal::NUM || vals[0].type != Atom::INT) return -1;
    const double k = (double)vals[0].num;
    if (op == "<") return have.num < k;
    if (op == "<=") return have.num <= k;
    if (op == ">") return have.num > k;
    return have.num >= k;
}

// one term -> the literals it stands for (several = an OR, for a numeric `in` over a list with gaps)
int parse_term(const Binding &bind, const JVal *tmpl, const std::string &part, std::vector<ge_literal> &options, std::string &why) {
    const int pack = bind.pack;
    std::string p = squeeze(part);
    if (p.compare(0, 7, "player.") != 0) { why = "unsupported condition term: " + p; return -1; }
    size_t i = 7;
    while (i < p.size() && (isalnum((unsigned char)p[i]) || p[i] == '_')) i++;
    std::string field = p.substr(7, i - 7);
    while (i < p.size() && p[i] == ' ') i++;
    std::string rest = p.substr(i), lrest = lower(rest), op;
    for (const char *cand : {"==", "!=", "<=", ">=", "<", ">", "not in ", "in "})
        if (lrest.compare(0, strlen(cand), cand) == 0) { op = cand; break; }
    if (field.empty() || op.empty()) { why = "unsupported condition term: " + p; return -1; }
    std::string rhs = squeeze(rest.substr(op.size()));
    while (!op.empty() && op.back() == ' ') op.pop_back();
    const bool in_op = op == "in" || op == "not in";
    std::vector<Atom> vals;
    if (in_op) {
        if (rhs.size() < 2 || rhs[0] != '[' || rhs.back() != ']') { why = "unsupported list literal: " + rhs; return -1; }
        std::string inner = rhs.substr(1, rhs.size() - 2), cur;
        char quote = 0;
        auto flush = [&]() -> bool {
            if (squeeze(cur).empty()) { cur.clear(); return true; }
            Atom a;
            if (!parse_atom(cur, a)) return false;
            vals.push_back(a);
            cur.clear();
            return true;
        };
        for (char c : inner) {
            if (quote) { cur += c; if (c == quote) quote = 0; continue; }
            if (c == '\'' || c == '"') { quote = c; cur += c; continue; }
            if (c == ',') { if (!flush()) { why = "unsupported literal in: " + p; return -1; } continue; }
            cur += c;
        }
        if (!flush()) { why = "unsupported literal in: " + p; return -1; }
        if (vals.empty()) { why = "empty list in: " + p; return -1; }
    } else {
        Atom a;
        if (!parse_atom(rhs, a)) { why = "unsupported literal in: " + p; return -1; }
        vals.push_back(a);
    }
    const bool neg = op == "!=" || op == "not in";
    const std::string declared_name = field;
    field = bind.slot_of(declared_name);                                // from here on: the canonical slot ("" = none)
    if (!pack_models(pack, field) && tmpl && declared_name != "name") {
        const JVal *have = tmpl->get(declared_name.c_str());
        if (have && (have->type == JVal::BOOL || (have->type == JVal::NUM && have->num == (double)(long)have->num) || have->type == JVal::STR)) {
            // a declared field outside the pack folds to a constant - unless the phase graph's own text names it: a field
            // the Referee is told to update (the generator prompt's `player.is_current_turn`,
            // dsl_phases_generation_prompt.txt:121) is state no rule pack carries
            if (bind.mentions(declared_name)) {
                why = "condition on '" + declared_name + "': the phases' text names this field (it may be written during play) and the rule pack does not model it: " + p;
                return -1;
            }
            const int holds = const_term(*have, op, vals);
            if (holds < 0) { why = "unsupported comparison on a non-numeric field: " + p; return -1; }
            ge_literal l;
            memset(&l, 0, sizeof l);
            l.kind = GE_LIT_BASE; l.bases = 0; l.neg = (uint8_t)holds;          // empty base set = never; negated = always
            options.push_back(l);
            return 0;
        }
    }
    int top = 0;
    const int nf = num_field_of(pack, field, top);
    bool all_int = true;
    for (auto &v : vals) all_int = all_int && v.type == Atom::INT;
    if (nf >= 0 && all_int) {
        auto clip = [&](long lo, long hi) {
            ge_literal l;
            memset(&l, 0, sizeof l);
            l.kind = GE_LIT_NUM; l.num_field = (uint8_t)nf;
            if (lo < 0) lo = 0;
            if (hi > top) hi = top;
            if (lo > hi) { lo = 1; hi = 0; }                          // never true
            l.lo = (uint8_t)lo; l.hi = (uint8_t)hi;
            return l;
        };
        if (op == "==" || op == "!=" || in_op) {
            std::vector<long> ks;
            for (auto &v : vals) ks.push_back(v.num);
            std::sort(ks.begin(), ks.end());
            ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
            std::vector<std::pair<long, long>> runs;                  // contiguous runs of the value list
            for (long k : ks) {
                if (!runs.empty() && k == runs.back().second + 1) runs.back().second = k;
                else runs.push_back({k, k});
            }
            if (neg && runs.size() > 1) { why = "unsupported: 'not in' over a non-contiguous list: " + p; return -1; }
            for (auto &r : runs) { ge_literal l = clip(r.first, r.second); l.neg = neg; options.push_back(l); }
        } else {
            const long k = vals[0].num;
            if (op == "<") options.push_back(clip(0, k - 1));
            else if (op == "<=") options.push_back(clip(0, k));
            else if (op == ">") options.push_back(clip(k + 1, top));
            else options.push_back(clip(k, top));
        }
        return 0;
    }
    if (!(op == "==" || op == "!=" || in_op)) { why = "unsupported comparison on a non-numeric field: " + p; return -1; }
    ge_literal l;
    memset(&l, 0, sizeof l);
    l.kind = GE_LIT_BASE;
    int flips = 0;                                                    // bit 0: some value keeps, bit 1: some value flips
    for (auto &v : vals) {
        int base = -1;
        bool flip = false;
        if (v.type == Atom::BOOL || (v.type == Atom::INT && (v.num == 0 || v.num == 1) && is_bool_field(pack, field))) {
            base = base_of(pack, field, std::string(), false);
            flip = v.type == Atom::BOOL ? !v.b : v.num == 0;
        } else if (v.type == Atom::STR) {
            base = base_of(pack, field, v.str, true);
        } else { why = "unsupported value in: " + p; return -1; }
        if (base < 0) { why = "condition field not in rule pack: " + p; return -1; }
        l.bases |= (uint16_t)(1u << base);
        flips |= flip ? 2 : 1;
    }
    if (flips == 3) { why = "unsupported: a boolean list with both values: " + p; return -1; }
    l.neg = (neg != (flips == 2)) ? 1 : 0;
    options.push_back(l);
    return 0;
}

// the whole condition -> row.clause[][] (+ term_base / term_neg and generic = 0 when it is a plain conjunction)
int parse_condition(const Binding &bind, const JVal *tmpl, const std::string &cond_in, ge_phase_row &row, std::string &why) {
    row.n_terms = 0; row.n_clauses = 0; row.generic = 0;
    const std::string cond = squeeze(cond_in);
    if (cond.empty()) return 0;
    {   // parentheses are outside the grammar (a '(' inside a quoted string is data)
        char quote = 0;
        for (char c : cond) {
            if (quote) { if (c == quote) quote = 0; continue; }
            if (c == '\'' || c == '"') quote = c;
            else if (c == '(' || c == ')') { why = "unsupported condition (parentheses): " + cond; return -1; }
        }
    }
    std::vector<Clause> clauses;
    for (const std::string &alt : split_kw(cond, "or")) {
        std::vector<Clause> partial(1);
        for (const std::string &part : split_kw(alt, "and")) {
            std::vector<ge_literal> options;
            if (parse_term(bind, tmpl, part, options, why) != 0) return -1;
            std::vector<Clause> next;
            for (auto &c : partial)
                for (auto &o : options) { Clause x = c; x.push_back(o); next.push_back(x); }
            partial.swap(next);
            if (partial.size() > GE_MAX_CLAUSES) { why = "too many condition alternatives"; return -1; }
        }
        for (auto &c : partial) clauses.push_back(c);
    }
    if (clauses.size() > GE_MAX_CLAUSES) { why = "too many condition alternatives"; return -1; }
    for (auto &c : clauses)
        if (c.size() > GE_MAX_TERMS) { why = "too many condition terms"; return -1; }
    row.n_clauses = (uint8_t)clauses.size();
    bool plain = clauses.size() == 1;
    for (size_t ci = 0; ci < clauses.size(); ci++) {
        row.clause_len[ci] = (uint8_t)clauses[ci].size();
        for (size_t li = 0; li < clauses[ci].size(); li++) {
            const ge_literal &l = clauses[ci][li];
            row.clause[ci][li] = l;
            plain = plain && l.kind == GE_LIT_BASE && l.bases != 0 && (l.bases & (l.bases - 1)) == 0;
        }
    }
    row.generic = plain ? 0 : 1;
    if (plain) {
        for (const ge_literal &l : clauses[0]) {
            int b = 0;
            while (!((l.bases >> b) & 1)) b++;
            row.term_base[row.n_terms] = (uint8_t)b;
            row.term_neg[row.n_terms] = l.neg;
            row.n_terms++;
        }
    }
    return 0;
}

// a clause holds the positive (neg = 0) / negative single-base literal `base`
bool clause_has(const ge_phase_row &r, int c, int base, int neg) {
    for (int j = 0; j < r.clause_len[c]; j++) {
        const ge_literal &l = r.clause[c][j];
        if (l.kind == GE_LIT_BASE && l.bases == (1u << base) && l.neg == neg) return true;
    }
    return false;
}

// `rows`: the table's phases with their effects, for keys that name a phase ("... follows Dawn Reveal ...")
int resolver_for(const std::string &key, const ge_phase_row *rows, int n_rows) {
    std::string k = lower(key);
    if (has(k, "no living werewol") || has(k, "all werewolves eliminated")) return GE_RES_WOLVES_ZERO;
    if (has(k, "outnumber")) return GE_RES_WOLVES_GE_VILLAGERS;
    if (has(k, "follows a day")) return GE_RES_FOLLOWS_DAY;
    if (has(k, "follows a night")) return GE_RES_FOLLOWS_NIGHT;
    if (has(k, "all players have completed")) return GE_RES_ALL_ROUNDS_DONE;
    if (k.compare(0, 9, "otherwise") == 0) return GE_RES_OTHERWISE;
    const size_t at = k.find("follows");
    if (at != std::string::npos) {
        // "follows <phase name>": what matters is which resolution that phase performs; the longest name wins
        const std::string tail = k.substr(at + 7);
        int best = -1;
        size_t best_len = 0;
        for (int i = 0; i < n_rows; i++) {
            const std::string nm = lower(rows[i].name);
            if (!nm.empty() && nm.size() > best_len && has(tail, nm.c_str())) { best = i; best_len = nm.size(); }
        }
        if (best >= 0 && rows[best].effect == GE_EFF_DAY_RESOLVE) return GE_RES_FOLLOWS_DAY;
        if (best >= 0 && rows[best].effect == GE_EFF_NIGHT_RESOLVE) return GE_RES_FOLLOWS_NIGHT;
    }
    return -1;
}

}  // namespace

static int compile_impl(const char *dsl_json, size_t len, int rounds, ge_game_table *out,
                        char *err_buf, size_t err_cap) {
    if (!dsl_json || !out || rounds < 1) return GE_ERR_ARG;
    Err err{err_buf, err_cap};
    if (err_buf && err_cap) err_buf[0] = 0;
    Parser ps{dsl_json, dsl_json + len, std::string()};
    JPtr root;
    if (!ps.value(root) || root->type != JVal::OBJ) return err.set("JSON: " + (ps.err.empty() ? std::string("not an object") : ps.err));
    const JVal *decl = root->get("declaration");
    const JVal *phases = root->get("phases");
    if (!decl || decl->type != JVal::OBJ || !phases || phases->type != JVal::OBJ || phases->obj.empty())
        return err.set("DSL needs top-level 'declaration' and 'phases'");

    ge_game_table t;
    memset(&t, 0, sizeof t);
    t.abi_version = GE_ABI_VERSION;
    t.rounds = rounds;
    t.min_players = as_int(decl->get("min_players"));

    const JVal *ps_def = decl->get("player_states");
    auto declared = [&](const char *f) { return ps_def && ps_def->get(f) != nullptr; };
    if (declared("role") && declared("team") && declared("is_alive"))
        t.pack = GE_PACK_WEREWOLF;
    else if (declared("is_speaker") && declared("lie_index") && declared("vote_choice") && declared("total_score"))
        t.pack = GE_PACK_TWO_TRUTHS;
    else
        return err.set("no rule pack matches declaration.player_states");

    Binding bind;
    std::string twice;
    if (!bind.bind(t.pack, ps_def, twice)) return err.set("two declared fields bind to one state slot: " + twice);
    for (int s = 0; s < bind.n; s++) copy_name(t.field_names[s], bind.declared[s]);
    bind.collect_text(phases, std::string());

    if (t.pack == GE_PACK_WEREWOLF) {
        const JVal *roles = decl->get("roles");
        if (roles && roles->type == JVal::ARR)
            for (auto &r : roles->arr) {
                std::string nm = as_str(r->get("name"));
                int c = role_class(nm);
                if (!t.role_names[c][0]) copy_name(t.role_names[c], nm);
            }
        for (int c = 1; c <= 4; c++)
            if (!t.role_names[c][0]) return err.set("werewolf pack needs Villager/Werewolf/Doctor/Detective roles");
    }

    // player_states_template: utils.py:603-609 ends up taking the first template entry
    const JVal *tmpl = nullptr;
    if (const JVal *pst = decl->get("player_states_template"))
        if (const JVal *tps = pst->get("player_states"))
            if (tps->type == JVal::OBJ && !tps->obj.empty()) {
                tmpl = tps->get("1");
                if (!tmpl || !truthy(tmpl)) tmpl = tps->obj.front().second.get();
            }
    if (!tmpl || tmpl->type != JVal::OBJ) return err.set("declaration.player_states_template.player_states is missing");
    uint8_t *f = t.init_fields;
    auto tf = [&](int slot) { return bind.get(tmpl, slot); };          // the template's value of a slot (nullptr: not declared)
    if (t.pack == GE_PACK_WEREWOLF) {
        std::string role = as_str(tf(GE_WW_ROLE)), team = as_str(tf(GE_WW_TEAM));
        f[0] = 0;
        for (int c = 1; c <= 4; c++) if (!role.empty() && role == t.role_names[c]) f[0] = (uint8_t)c;
        f[1] = team == "villagers" ? 1 : team == "werewolves" ? 2 : 0;
        const JVal *alive = tf(GE_WW_IS_ALIVE);
        f[2] = alive ? truthy(alive) : 1;
        f[3] = truthy(tf(GE_WW_ROLE_REVEALED)); f[4] = truthy(tf(GE_WW_CAN_VOTE));
        f[5] = truthy(tf(GE_WW_HAS_SECRET_ROLE)); f[6] = truthy(tf(GE_WW_NIGHT_ELIGIBLE));
        f[7] = truthy(tf(GE_WW_NIGHT_SUBMITTED)); f[8] = (uint8_t)as_int(tf(GE_WW_SELECTED_TARGET));
    } else {
        f[0] = truthy(tf(GE_TT_IS_SPEAKER)); f[1] = truthy(tf(GE_TT_STATEMENTS_SUBMITTED));
        f[2] = (uint8_t)as_int(tf(GE_TT_LIE_INDEX)); f[3] = truthy(tf(GE_TT_LIE_REVEALED));
        f[4] = truthy(tf(GE_TT_CAN_VOTE)); f[5] = (uint8_t)as_int(tf(GE_TT_VOTE_CHOICE));
        f[6] = truthy(tf(GE_TT_HAS_VOTED)); f[7] = (uint8_t)as_int(tf(GE_TT_TOTAL_SCORE));
        f[8] = (uint8_t)as_int(tf(GE_TT_ROUNDS_AS_SPEAKER));
    }

    if (phases->obj.size() > GE_MAX_PHASES) return err.set("too many phases");
    t.n_phases = (int32_t)phases->obj.size();
    std::vector<int> ids;
    for (auto &kv : phases->obj) {
        char *e = nullptr;
        long id = strtol(kv.first.c_str(), &e, 10);
        if (e == kv.first.c_str() || *e) return err.set("phase key is not an integer: " + kv.first);
        ids.push_back((int)id);
    }
    bool has0 = false;
    for (int id : ids) has0 = has0 || id == 0;
    if (!has0) return err.set("no phase with id 0 (AgentState.current_phase_id starts at 0)");

    for (int i = 0; i < t.n_phases; i++) {
        const JVal *ph = phases->obj[i].second.get();
        ge_phase_row &row = t.rows[i];
        row.phase_id = ids[i];
        char where[32];
        snprintf(where, sizeof where, "phase %d: ", ids[i]);
        std::string name = as_str(ph->get("name"));
        if (name.empty()) name = std::string("Phase ") + std::to_string(ids[i]);
        copy_name(row.name, name);
        const JVal *cc = ph->get("completion_criteria");
        std::string ctype = lower(as_str(cc ? cc->get("type") : nullptr));
        if (ctype.empty() || ctype == "ui_displayed") row.completion = GE_COMP_UI;
        else if (ctype == "timer") row.completion = GE_COMP_TIMER;
        else if (ctype == "player_action") row.completion = GE_COMP_ACTION;
        else return err.set(where + ("unknown completion type " + ctype));

        std::vector<std::string> tools;
        if (const JVal *acts = ph->get("actions"))
            if (acts->type == JVal::ARR)
                for (auto &a : acts->arr)
                    if (const JVal *tl = a->get("tools"))
                        if (tl->type == JVal::ARR)
                            for (auto &x : tl->arr) tools.push_back(as_str(x.get()));
        auto has_tool = [&](const char *n) { for (auto &x : tools) if (x == n) return true; return false; };
        std::string lname = lower(name);
        std::string text = lname + " " + lower(as_str(ph->get("description")));

        if (row.completion == GE_COMP_ACTION) {
            const JVal *tp = cc->get("target_players");
            std::string why;
            const JVal *wf = cc->get("wait_for");
            if (wf && !wf->is_null()) {
                // all three kinds mean "feedback from ALL target players" (prompt :138 "Completion Logic"); anything else is an error
                const std::string w = as_str(wf);
                if (w != "single_player_choice" && w != "all_players_action" && w != "multiple_players_action")
                    return err.set(where + ("unknown wait_for " + w));
            }
            if (parse_condition(bind, tmpl, as_str(tp ? tp->get("condition") : nullptr), row, why) != 0)
                return err.set(where + why);
            // the action kind, from the condition: every alternative is classified on its own and all must agree
            if (row.n_clauses == 0) return err.set(where + std::string("cannot classify the player action"));
            for (int c = 0; c < row.n_clauses; c++) {
                int act = GE_ACT_NONE;
                if (t.pack == GE_PACK_WEREWOLF) {
                    if (clause_has(row, c, 7 + ROLE_WEREWOLF, 0)) act = GE_ACT_WOLF_TARGET;
                    else if (clause_has(row, c, 7 + ROLE_DOCTOR, 0)) act = GE_ACT_DOCTOR_PROTECT;
                    else if (clause_has(row, c, 7 + ROLE_DETECTIVE, 0)) act = GE_ACT_DETECTIVE;
                    else if (clause_has(row, c, 7, 0)) act = GE_ACT_WOLF_TARGET;          // "all alive werewolves" written by team
                    else if (clause_has(row, c, 1, 0)) act = GE_ACT_DAY_VOTE;
                } else {
                    if (clause_has(row, c, 0, 1)) act = GE_ACT_TT_VOTE;
                    else if (clause_has(row, c, 0, 0))
                        act = (has_tool("createTextInputPanel") || has(lname, "statement")) ? GE_ACT_TT_STATEMENTS : GE_ACT_TT_LIE;
                }
                if (act == GE_ACT_NONE) return err.set(where + std::string("cannot classify the player action"));
                if (c > 0 && act != row.act) return err.set(where + std::string("the condition's alternatives describe different player actions"));
                row.act = (uint8_t)act;
            }
        }

        if (t.pack == GE_PACK_WEREWOLF) {
            if (has(lname, "role assignment") || has(text, "assign roles")) row.effect = GE_EFF_ASSIGN_ROLES;
            else if (has_tool("markPlayerDead")) {
                if (has(text, "night")) row.effect = GE_EFF_NIGHT_RESOLVE;
                else if (has(text, "vot")) row.effect = GE_EFF_DAY_RESOLVE;
            }
            if (row.act == GE_ACT_WOLF_TARGET && row.effect == GE_EFF_NONE) row.effect = GE_EFF_NIGHT_BEGIN;
        } else {
            if (has(lname, "round start")) row.effect = GE_EFF_TT_ROUND_START;
            else if (has(lname, "reveal")) row.effect = GE_EFF_TT_REVEAL;
            else if (has(lname, "scoring")) row.effect = GE_EFF_TT_SCORE;
        }

    }
    // branches second: a key may name another phase, whose effect must be known by then
    for (int i = 0; i < t.n_phases; i++) {
        const JVal *ph = phases->obj[i].second.get();
        ge_phase_row &row = t.rows[i];
        char where[32];
        snprintf(where, sizeof where, "phase %d: ", ids[i]);
        const JVal *nx = ph->get("next_phase");
        auto add_branch = [&](int res, const JVal *tgt) -> int {
            if (row.n_branches >= GE_MAX_BRANCHES) return -1;
            const JVal *idv = tgt ? tgt->get("id") : nullptr;
            if (!idv || idv->type != JVal::NUM) return -2;
            int tid = (int)idv->num, ti = -1;
            for (int k = 0; k < t.n_phases; k++) if (ids[k] == tid) ti = k;
            if (ti < 0) return -3;
            row.br_res[row.n_branches] = (uint8_t)res;
            row.br_target[row.n_branches] = (uint8_t)ti;
            row.n_branches++;
            return 0;
        };
        if (nx && nx->type == JVal::OBJ) {
            const JVal *idv = nx->get("id");
            if (idv && idv->type == JVal::NUM) {
                if (add_branch(GE_RES_ALWAYS, nx) != 0) return err.set(where + std::string("next_phase id not in phases"));
            } else {
                for (auto &kv : nx->obj) {
                    int res = resolver_for(kv.first, t.rows, t.n_phases);
                    if (res < 0) return err.set(where + ("no branch resolver for next_phase key '" + kv.first + "'"));
                    int rc = add_branch(res, kv.second.get());
                    if (rc == -1) return err.set(where + std::string("too many branches"));
                    if (rc != 0) return err.set(where + std::string("branch target id not in phases"));
                }
            }
        } else if (nx && !nx->is_null()) {
            return err.set(where + std::string("next_phase must be null or an object"));
        }
    }
    *out = t;
    return GE_OK;
}

// nothing throws across the C ABI: the parser and the compiler allocate (std::string / std::vector)
extern "C" int ge_table_compile_json(const char *dsl_json, size_t len, int rounds, ge_game_table *out,
                                     char *err_buf, size_t err_cap) {
    try {
        return compile_impl(dsl_json, len, rounds, out, err_buf, err_cap);
    } catch (const std::bad_alloc &) {
        return GE_ERR_NOMEM;
    } catch (...) {
        return GE_ERR_DSL;
    }
}
