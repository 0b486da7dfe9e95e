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

// base predicate index of `player.<field> == <value>` inside a pack, -1 if the pack has none
int base_of(int pack, const std::string &field, const std::string &sval, bool is_str) {
    if (pack == GE_PACK_WEREWOLF) {
        static const char *bools[] = {"is_alive", "can_vote", "role_revealed", "has_secret_role",
                                      "night_action_eligible", "night_action_submitted"};
        if (!is_str) { for (int i = 0; i < 6; i++) if (field == bools[i]) return i; return -1; }
        if (field == "team") { if (sval == "villagers") return 6; if (sval == "werewolves") return 7; return -1; }
        if (field == "role") return 7 + role_class(sval);
        return -1;
    }
    static const char *bools[] = {"is_speaker", "statements_submitted", "lie_revealed", "can_vote", "has_voted"};
    if (is_str) return -1;
    for (int i = 0; i < 5; i++) if (field == bools[i]) return i;
    return -1;
}

// `player.f == v and player.g == w`  (ww:247,279,310,390; tt phases 2,3,5)
int parse_condition(int pack, const std::string &cond, ge_phase_row &row, std::string &why) {
    size_t pos = 0;
    row.n_terms = 0;
    std::string s = cond;
    while (pos < s.size()) {
        size_t nxt = s.find(" and ", pos);
        std::string part = s.substr(pos, nxt == std::string::npos ? std::string::npos : nxt - pos);
        pos = nxt == std::string::npos ? s.size() : nxt + 5;
        size_t a = part.find_first_not_of(" \t\n"), z = part.find_last_not_of(" \t\n");
        if (a == std::string::npos) continue;
        part = part.substr(a, z - a + 1);
        if (part.compare(0, 7, "player.") != 0) { why = "unsupported condition term: " + part; return -1; }
        size_t i = 7;
        while (i < part.size() && (isalnum((unsigned char)part[i]) || part[i] == '_')) i++;
        std::string field = part.substr(7, i - 7);
        while (i < part.size() && part[i] == ' ') i++;
        bool neg;
        if (part.compare(i, 2, "==") == 0) neg = false;
        else if (part.compare(i, 2, "!=") == 0) neg = true;
        else { why = "unsupported operator in: " + part; return -1; }
        i += 2;
        while (i < part.size() && part[i] == ' ') i++;
        std::string lit = part.substr(i);
        bool is_str = false;
        std::string sval;
        if (lit.size() >= 2 && (lit[0] == '\'' || lit[0] == '"') && lit.back() == lit[0]) {
            is_str = true; sval = lit.substr(1, lit.size() - 2);
        } else {
            std::string l = lower(lit);
            if (l == "false") neg = !neg;
            else if (l != "true") { why = "unsupported literal in: " + part; return -1; }
        }
        int base = base_of(pack, field, sval, is_str);
        if (base < 0) { why = "condition field not in rule pack: " + part; return -1; }
        if (row.n_terms >= GE_MAX_TERMS) { why = "too many condition terms"; return -1; }
        row.term_base[row.n_terms] = (uint8_t)base;
        row.term_neg[row.n_terms] = neg ? 1 : 0;
        row.n_terms++;
    }
    return 0;
}

bool term_is(const ge_phase_row &r, int base, int neg) {
    for (int j = 0; j < r.n_terms; j++)
        if (r.term_base[j] == base && r.term_neg[j] == neg) return true;
    return false;
}

int resolver_for(const std::string &key) {
    std::string k = lower(key);
    if (has(k, "no living werewol") || has(k, "all werewolves eliminated")) return GE_RES_WOLVES_ZERO;
    if (has(k, "outnumber")) return GE_RES_WOLVES_GE_VILLAGERS;
    if (has(k, "follows a day")) return GE_RES_FOLLOWS_DAY;
    if (has(k, "follows a night")) return GE_RES_FOLLOWS_NIGHT;
    if (has(k, "all players have completed")) return GE_RES_ALL_ROUNDS_DONE;
    if (k.compare(0, 9, "otherwise") == 0) return GE_RES_OTHERWISE;
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
    if (declared("role") && declared("team") && declared("is_alive") && declared("selected_target_id"))
        t.pack = GE_PACK_WEREWOLF;
    else if (declared("is_speaker") && declared("lie_index") && declared("vote_choice") && declared("total_score"))
        t.pack = GE_PACK_TWO_TRUTHS;
    else
        return err.set("no rule pack matches declaration.player_states");

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
    if (t.pack == GE_PACK_WEREWOLF) {
        std::string role = as_str(tmpl->get("role")), team = as_str(tmpl->get("team"));
        f[0] = 0;
        for (int c = 1; c <= 4; c++) if (!role.empty() && role == t.role_names[c]) f[0] = (uint8_t)c;
        f[1] = team == "villagers" ? 1 : team == "werewolves" ? 2 : 0;
        const JVal *alive = tmpl->get("is_alive");
        f[2] = alive ? truthy(alive) : 1;
        f[3] = truthy(tmpl->get("role_revealed")); f[4] = truthy(tmpl->get("can_vote"));
        f[5] = truthy(tmpl->get("has_secret_role")); f[6] = truthy(tmpl->get("night_action_eligible"));
        f[7] = truthy(tmpl->get("night_action_submitted")); f[8] = (uint8_t)as_int(tmpl->get("selected_target_id"));
    } else {
        f[0] = truthy(tmpl->get("is_speaker")); f[1] = truthy(tmpl->get("statements_submitted"));
        f[2] = (uint8_t)as_int(tmpl->get("lie_index")); f[3] = truthy(tmpl->get("lie_revealed"));
        f[4] = truthy(tmpl->get("can_vote")); f[5] = (uint8_t)as_int(tmpl->get("vote_choice"));
        f[6] = truthy(tmpl->get("has_voted")); f[7] = (uint8_t)as_int(tmpl->get("total_score"));
        f[8] = (uint8_t)as_int(tmpl->get("rounds_as_speaker"));
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
            if (parse_condition(t.pack, as_str(tp ? tp->get("condition") : nullptr), row, why) != 0)
                return err.set(where + why);
            if (t.pack == GE_PACK_WEREWOLF) {
                if (term_is(row, 7 + ROLE_WEREWOLF, 0)) row.act = GE_ACT_WOLF_TARGET;
                else if (term_is(row, 7 + ROLE_DOCTOR, 0)) row.act = GE_ACT_DOCTOR_PROTECT;
                else if (term_is(row, 7 + ROLE_DETECTIVE, 0)) row.act = GE_ACT_DETECTIVE;
                else if (term_is(row, 1, 0)) row.act = GE_ACT_DAY_VOTE;
            } else {
                if (term_is(row, 0, 1)) row.act = GE_ACT_TT_VOTE;
                else if (term_is(row, 0, 0))
                    row.act = (has_tool("createTextInputPanel") || has(lname, "statement")) ? GE_ACT_TT_STATEMENTS : GE_ACT_TT_LIE;
            }
            if (row.act == GE_ACT_NONE) return err.set(where + std::string("cannot classify the player action"));
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
                    int res = resolver_for(kv.first);
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
