/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, one-room-at-a-time restatement of one turn of the reference's loop
 *   InitialRouterNode -> BotBehaviorNode -> PhaseNode -> RefereeNode -> ActionExecutor
 *   (/root/reference/agent/game_agent_v2.py:198, 468, 987, 619, 1243; graph v2:1571-1587)
 * with every LLM call replaced by the fixed policy of POLICY.md, over the canonical
 * integer projection of the reference's dict state.  The tool-call plumbing it
 * restates is agent/tools/backend_tools.py:204-225 (update_player_state: set one
 * field) and :285-344 (update_player_actions: append to the per-player log).
 *
 * Pinned: tests/test_oracle_golden.py checks it turn by turn against golden vectors
 * produced by driving the reference's own node coroutines (oracle/refharness).
 * The reference has no tests or fixtures for this path, so those
 * reference-run vectors are the pin.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (game_engine_amd) never does.
 *
 * Deliberately written differently from the HIP kernels: per-player byte arrays
 * and loops here, bitboards there.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { PACK_WW = 1, PACK_TT = 2 };
enum { COMP_UI = 0, COMP_TIMER = 1, COMP_ACTION = 2 };
enum { ACT_NONE, ACT_WOLF_TARGET, ACT_DOCTOR_PROTECT, ACT_DETECTIVE, ACT_DAY_VOTE,
       ACT_TT_STATEMENTS, ACT_TT_LIE, ACT_TT_VOTE };
enum { EFF_NONE, EFF_ASSIGN_ROLES, EFF_NIGHT_BEGIN, EFF_NIGHT_RESOLVE, EFF_DAY_RESOLVE,
       EFF_TT_ROUND_START, EFF_TT_REVEAL, EFF_TT_SCORE };
enum { RES_ALWAYS, RES_WOLVES_ZERO, RES_WOLVES_GE_VILLAGERS, RES_FOLLOWS_DAY,
       RES_FOLLOWS_NIGHT, RES_ALL_ROUNDS_DONE, RES_OTHERWISE };
enum { ROLE_NONE, ROLE_VILLAGER, ROLE_WEREWOLF, ROLE_DOCTOR, ROLE_DETECTIVE };
enum { TEAM_NONE, TEAM_VILLAGERS, TEAM_WEREWOLVES };

/* werewolf player fields */
enum { W_ROLE, W_TEAM, W_ALIVE, W_REVEALED, W_CAN_VOTE, W_SECRET, W_ELIG, W_SUB, W_TARGET,
       W_ACTED, W_CHOICE };
/* two-truths player fields */
enum { T_SPEAKER, T_SUBMITTED, T_LIE, T_REVEALED, T_CAN_VOTE, T_VOTE, T_HAS_VOTED, T_SCORE,
       T_ROUNDS, T_ACTED, T_CHOICE };
#define F_ACTED 9
#define F_CHOICE 10

/* one literal of a condition's clause form (oracle/dsl_table.py Literal):
 * kind 1: the player has ANY base predicate of the bit set `bases`; kind 2: lo <= numeric field <= hi */
typedef struct {
    uint8_t kind, neg, num_field, pad;
    uint16_t bases;
    uint8_t lo, hi;
} orc_literal;

typedef struct {
    uint8_t completion, act, effect, n_terms, n_branches, pad[3];
    uint8_t term_base[4], term_neg[4];
    uint8_t br_res[4], br_target[4];
    int32_t phase_id;
    /* target_players.condition as OR of AND-clauses (always filled; n_terms/term_* are not read) */
    uint8_t n_clauses, clause_len[4], pad2[3];
    orc_literal clause[4][4];
} orc_phase;

typedef struct {
    int32_t pack, n_phases, rounds, pad;
    uint8_t init_fields[12];
    uint8_t pad2[4];
    orc_phase ph[32];
} orc_table;

typedef struct {
    uint8_t phase, prev, phase0_done, n;
    int32_t end_turn;
    int32_t games;            /* restart mode: games completed on this slot before the current one */
    uint8_t p[16][12];
    uint8_t det[16];          /* 0 unknown, 1 villagers, 2 werewolves (the detective's memory) */
    /* what the most recent turn logged and decided (not part of the projection) */
    uint8_t ev_from, ev_to, ev_restarted, ev_pad;
    uint16_t ev_newly, ev_pad2;
    uint8_t ev_choice[16];
} orc_room;

/* ---- RNG: POLICY.md §RNG (restated from the text, cf. oracle/rng.py) ---- */
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x;
}
static uint32_t room_key(uint64_t seed, uint64_t room) {
    uint32_t k = mix32((uint32_t)seed ^ 0x243F6A88u);
    k = mix32(k ^ (uint32_t)(seed >> 32));
    k = mix32(k ^ (uint32_t)room);
    k = mix32(k ^ (uint32_t)(room >> 32));
    return k;
}
static uint32_t turn_key(uint32_t rk, uint32_t turn) { return mix32(rk ^ (turn * 0x9E3779B9u)); }
static uint32_t deal_key(uint32_t rk, uint32_t game) { return mix32(rk ^ 0x44454C31u ^ (game * 0x9E3779B9u)); }
static uint32_t draw(uint32_t tk, uint32_t idx) { return mix32(tk + (idx + 1u) * 0x9E3779B9u); }
static uint32_t pick(uint32_t d, uint32_t k) { return (uint32_t)(((uint64_t)d * k) >> 32); }

/* i-th (0-based) player, ascending, for whom cand[] is set; -1 if none */
static int nth_candidate(const uint8_t *cand, int n, int i) {
    for (int k = 0; k < n; k++)
        if (cand[k] && i-- == 0) return k;
    return -1;
}

static int is_alive(const orc_table *tb, const orc_room *r, int i) {
    return tb->pack == PACK_WW ? r->p[i][W_ALIVE] : 1;
}

static int base_true(const orc_table *tb, const orc_room *r, int i, int base) {
    const uint8_t *f = r->p[i];
    if (tb->pack == PACK_WW) {
        switch (base) {
        case 0: return f[W_ALIVE];   case 1: return f[W_CAN_VOTE]; case 2: return f[W_REVEALED];
        case 3: return f[W_SECRET];  case 4: return f[W_ELIG];     case 5: return f[W_SUB];
        case 6: return f[W_TEAM] == TEAM_VILLAGERS; case 7: return f[W_TEAM] == TEAM_WEREWOLVES;
        default: return f[W_ROLE] == base - 7;          /* 8..11 -> role 1..4 */
        }
    }
    switch (base) {
    case 0: return f[T_SPEAKER]; case 1: return f[T_SUBMITTED]; case 2: return f[T_REVEALED];
    case 3: return f[T_CAN_VOTE]; default: return f[T_HAS_VOTED];
    }
}

/* a numeric player field a condition may compare (index: dsl_table.WW_NUM / TT_NUM) */
static int num_value(const orc_table *tb, const orc_room *r, int i, int field) {
    const uint8_t *f = r->p[i];
    if (tb->pack == PACK_WW) return f[W_TARGET];
    switch (field) {
    case 1: return f[T_LIE]; case 2: return f[T_VOTE]; case 3: return f[T_SCORE]; default: return f[T_ROUNDS];
    }
}

static int literal_true(const orc_table *tb, const orc_room *r, int i, const orc_literal *l) {
    int ok = 0;
    if (l->kind == 1) {
        for (int b = 0; b < 16; b++)
            if (((l->bases >> b) & 1) && base_true(tb, r, i, b)) ok = 1;
    } else {
        const int v = num_value(tb, r, i, l->num_field);
        ok = v >= l->lo && v <= l->hi;
    }
    return ok != (l->neg != 0);
}

/* completion_criteria.target_players.condition AND alive; the condition is an OR of AND-clauses
 * (an empty condition = every living player) */
static int is_target(const orc_table *tb, const orc_phase *ph, const orc_room *r, int i) {
    if (!is_alive(tb, r, i)) return 0;
    if (ph->n_clauses == 0) return 1;
    for (int c = 0; c < ph->n_clauses; c++) {
        int all = 1;
        for (int t = 0; t < ph->clause_len[c]; t++)
            if (!literal_true(tb, r, i, &ph->clause[c][t])) { all = 0; break; }
        if (all) return 1;
    }
    return 0;
}

/* most votes wins, ties -> lowest id; 0 when nobody voted */
static int plurality(const int *cnt, int n) {
    int best = 0, arg = 0;
    for (int k = 1; k <= n; k++)
        if (cnt[k] > best) { best = cnt[k]; arg = k; }
    return arg;
}

static void kill_player(orc_room *r, int id) {
    uint8_t *f = r->p[id - 1];
    f[W_ALIVE] = 0; f[W_CAN_VOTE] = 0; f[W_ELIG] = 0; f[W_REVEALED] = 1;
}

void orc_room_init(const orc_table *tb, int n_players, orc_room *r) {
    memset(r, 0, sizeof *r);
    r->n = (uint8_t)n_players;
    r->end_turn = -1;
    for (int i = 0; i < n_players; i++) memcpy(r->p[i], tb->init_fields, 12);
}

/* One turn = one graph run of the reference. */
void orc_room_step(const orc_table *tb, orc_room *r, uint64_t seed, uint64_t room, uint32_t turn, uint32_t human_mask) {
    const int n = r->n;
    const int p = r->phase;
    const orc_phase *ph = &tb->ph[p];
    const uint32_t tk = turn_key(room_key(seed, room), turn);
    uint8_t newly[16] = {0};
    r->ev_from = r->ev_to = (uint8_t)p;
    r->ev_newly = 0;
    memset(r->ev_choice, 0, sizeof r->ev_choice);

    /* ---- BotBehaviorNode (v2:468-617): due bots act, one action per player per visit */
    if (ph->completion == COMP_ACTION) {
        uint8_t alive[16], wolf[16];
        int known_wolf_alive = -1;
        for (int i = 0; i < n; i++) {
            alive[i] = (uint8_t)is_alive(tb, r, i);
            wolf[i] = tb->pack == PACK_WW && r->p[i][W_TEAM] == TEAM_WEREWOLVES;
        }
        for (int i = n - 1; i >= 0; i--)
            if (tb->pack == PACK_WW && r->det[i] == 2 && alive[i]) known_wolf_alive = i;
        /* decisions are taken on the state at node entry, so collect first, apply after */
        uint8_t choice[16] = {0};
        for (int i = 0; i < n; i++) {
            if (!is_target(tb, ph, r, i) || r->p[i][F_ACTED]) continue;
            if ((human_mask >> i) & 1u) continue;      /* host-driven player: never acted for (bot_behavior prompt :3) */
            uint32_t d = draw(tk, (uint32_t)i);
            if ((d & 3u) == 0) continue;
            uint8_t cand[16];
            int k = 0, c = 0;
            switch (ph->act) {
            case ACT_WOLF_TARGET:
                for (int j = 0; j < n; j++) k += cand[j] = alive[j] && !wolf[j];
                break;
            case ACT_DOCTOR_PROTECT:
                for (int j = 0; j < n; j++) k += cand[j] = alive[j];
                break;
            case ACT_DETECTIVE:
                for (int j = 0; j < n; j++) k += cand[j] = alive[j] && j != i && r->det[j] == 0;
                if (!k) for (int j = 0; j < n; j++) k += cand[j] = alive[j] && j != i;
                break;
            case ACT_DAY_VOTE:
                if (wolf[i]) {
                    for (int j = 0; j < n; j++) k += cand[j] = alive[j] && !wolf[j];
                } else if (r->p[i][W_ROLE] == ROLE_DETECTIVE && known_wolf_alive >= 0) {
                    memset(cand, 0, sizeof cand); cand[known_wolf_alive] = 1; k = 1;
                } else {
                    for (int j = 0; j < n; j++) k += cand[j] = alive[j] && j != i;
                }
                break;
            case ACT_TT_STATEMENTS: c = 1; break;
            case ACT_TT_LIE:
            case ACT_TT_VOTE: c = 1 + (int)pick(d, 3); break;
            default: break;
            }
            if (ph->act >= ACT_WOLF_TARGET && ph->act <= ACT_DAY_VOTE) {
                if (!k) for (int j = 0; j < n; j++) k += cand[j] = alive[j];
                c = nth_candidate(cand, n, (int)pick(d, (uint32_t)k)) + 1;
            }
            choice[i] = (uint8_t)c;
            newly[i] = 1;
        }
        for (int i = 0; i < n; i++)
            if (newly[i]) {
                r->p[i][F_ACTED] = 1; r->p[i][F_CHOICE] = choice[i];
                r->ev_newly |= (uint16_t)(1u << i); r->ev_choice[i] = choice[i];
            }
    }

    /* ---- PhaseNode (v2:987-1241) */
    if (p == 0 && !r->phase0_done) {          /* phase-0 guard v2:1025-1052: Referee skipped */
        r->phase0_done = 1;
        return;
    }
    int q = p;
    if (ph->n_branches) {
        int complete = 1;
        if (ph->completion == COMP_ACTION)
            for (int i = 0; i < n; i++)
                if (is_target(tb, ph, r, i) && !r->p[i][F_ACTED]) complete = 0;
        if (complete) {
            int w = 0, g = 0, all_done = 1;
            for (int i = 0; i < n; i++) {
                if (tb->pack == PACK_WW) {
                    if (r->p[i][W_ALIVE] && r->p[i][W_TEAM] == TEAM_WEREWOLVES) w++;
                    if (r->p[i][W_ALIVE] && r->p[i][W_TEAM] == TEAM_VILLAGERS) g++;
                } else if (r->p[i][T_ROUNDS] < tb->rounds) all_done = 0;
            }
            int prev_eff = tb->ph[r->prev].effect;
            for (int b = 0; b < ph->n_branches; b++) {
                int ok;
                switch (ph->br_res[b]) {
                case RES_WOLVES_ZERO: ok = w == 0; break;
                case RES_WOLVES_GE_VILLAGERS: ok = w >= g; break;
                case RES_FOLLOWS_DAY: ok = prev_eff == EFF_DAY_RESOLVE; break;
                case RES_FOLLOWS_NIGHT: ok = prev_eff == EFF_NIGHT_RESOLVE; break;
                case RES_ALL_ROUNDS_DONE: ok = all_done; break;
                default: ok = 1; break;
                }
                if (ok) { q = ph->br_target[b]; break; }
            }
        }
    }

    /* ---- RefereeNode (v2:619-803): (A) record this turn's actions */
    for (int i = 0; i < n; i++) {
        if (!newly[i]) continue;
        uint8_t *f = r->p[i];
        int c = f[F_CHOICE];
        switch (ph->act) {
        case ACT_DETECTIVE:
            r->det[c - 1] = r->p[c - 1][W_TEAM] == TEAM_WEREWOLVES ? 2 : 1;
            /* fallthrough */
        case ACT_WOLF_TARGET:
        case ACT_DOCTOR_PROTECT: f[W_SUB] = 1; f[W_TARGET] = (uint8_t)c; break;
        case ACT_TT_STATEMENTS: f[T_SUBMITTED] = 1; break;
        case ACT_TT_LIE: f[T_LIE] = (uint8_t)c; break;
        case ACT_TT_VOTE: f[T_VOTE] = (uint8_t)c; f[T_HAS_VOTED] = 1; break;
        default: break;
        }
    }
    if (q == p) return;

    /* ---- RefereeNode (B): effect of entering q */
    const orc_phase *qh = &tb->ph[q];
    switch (qh->effect) {
    case EFF_ASSIGN_ROLES: {
        uint8_t rem[16];
        int left = n, nw = n / 4 > 1 ? n / 4 : 1;
        const uint32_t dk = deal_key(room_key(seed, room), (uint32_t)r->games);   /* roles are dealt per game */
        for (int i = 0; i < n; i++) { rem[i] = 1; r->p[i][W_ROLE] = ROLE_VILLAGER; }
        for (int j = 0; j < nw + 2 && left > 0; j++) {
            int i = nth_candidate(rem, n, (int)pick(draw(dk, 16u + (uint32_t)j), (uint32_t)left));
            r->p[i][W_ROLE] = j < nw ? ROLE_WEREWOLF : (j == nw ? ROLE_DOCTOR : ROLE_DETECTIVE);
            rem[i] = 0; left--;
        }
        for (int i = 0; i < n; i++) {
            uint8_t *f = r->p[i];
            f[W_TEAM] = f[W_ROLE] == ROLE_WEREWOLF ? TEAM_WEREWOLVES : TEAM_VILLAGERS;
            f[W_SECRET] = f[W_ELIG] = f[W_ROLE] != ROLE_VILLAGER;
        }
        break;
    }
    case EFF_NIGHT_BEGIN:
        for (int i = 0; i < n; i++) { r->p[i][W_SUB] = 0; r->p[i][W_TARGET] = 0; }
        break;
    case EFF_NIGHT_RESOLVE: {
        int cnt[17] = {0}, protect = 0;
        for (int i = 0; i < n; i++) {
            const uint8_t *f = r->p[i];
            if (!f[W_ALIVE]) continue;
            if (f[W_ROLE] == ROLE_WEREWOLF) cnt[f[W_TARGET]]++;
            if (f[W_ROLE] == ROLE_DOCTOR) protect = f[W_TARGET];
        }
        int victim = plurality(cnt, n);
        if (victim && victim != protect) kill_player(r, victim);
        break;
    }
    case EFF_DAY_RESOLVE: {
        int cnt[17] = {0};
        for (int i = 0; i < n; i++)
            if (r->p[i][W_ALIVE] && r->p[i][F_ACTED]) cnt[r->p[i][F_CHOICE]]++;
        int victim = plurality(cnt, n);
        if (victim) kill_player(r, victim);
        break;
    }
    case EFF_TT_ROUND_START: {
        int speaker = -1;
        for (int i = 0; i < n && speaker < 0; i++)
            if (r->p[i][T_ROUNDS] < tb->rounds) speaker = i;
        for (int i = 0; i < n; i++) {
            uint8_t *f = r->p[i];
            f[T_SPEAKER] = i == speaker; f[T_CAN_VOTE] = i != speaker;
            f[T_SUBMITTED] = f[T_LIE] = f[T_REVEALED] = f[T_VOTE] = f[T_HAS_VOTED] = 0;
        }
        break;
    }
    case EFF_TT_REVEAL:
        for (int i = 0; i < n; i++) if (r->p[i][T_SPEAKER]) r->p[i][T_REVEALED] = 1;
        break;
    case EFF_TT_SCORE: {
        int s = -1, fooled = 0;
        for (int i = 0; i < n && s < 0; i++) if (r->p[i][T_SPEAKER]) s = i;
        if (s < 0) break;
        for (int i = 0; i < n; i++) {
            if (i == s || !r->p[i][T_HAS_VOTED]) continue;
            if (r->p[i][T_VOTE] == r->p[s][T_LIE]) r->p[i][T_SCORE]++; else fooled++;
        }
        r->p[s][T_SCORE] = (uint8_t)(r->p[s][T_SCORE] + fooled);
        r->p[s][T_ROUNDS]++;
        break;
    }
    default: break;
    }
    /* a new visit starts with an empty action log */
    for (int i = 0; i < n; i++) { r->p[i][F_ACTED] = 0; r->p[i][F_CHOICE] = 0; }
    r->prev = (uint8_t)p;
    r->phase = (uint8_t)q;
    r->ev_to = (uint8_t)q;
    if (!qh->n_branches && r->end_turn < 0) r->end_turn = turn < 0xFFFEu ? (int32_t)turn : 0xFFFE;
}

/* rooms[i] is room (first_room + i); turns first_turn .. first_turn+n_turns-1 */
void orc_run(const orc_table *tb, uint64_t seed, uint64_t first_room, uint64_t n_rooms,
             uint32_t first_turn, uint32_t n_turns, orc_room *rooms, int threads, int restart, uint32_t human_mask) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_rooms; i++)
        for (uint32_t t = 0; t < n_turns; t++) {
            orc_room *r = &rooms[i];
            int restarted = 0;
            if (restart && !tb->ph[r->phase].n_branches) {
                restarted = 1;
                /* steady state: a finished room becomes a NEW room on the same slot; the clock
                 * (turn index) keeps running, so the new game draws fresh random numbers */
                int32_t g = r->games < 0xFFFF ? r->games + 1 : r->games;
                orc_room_init(tb, r->n, r);
                r->games = g;
            }
            orc_room_step(tb, r, seed, first_room + (uint64_t)i, first_turn + t, human_mask);
            r->ev_restarted = (uint8_t)restarted;
        }
    (void)threads;
}

/* A host-driven player's action logged between turns: joins this visit's log and gets the
 * Referee's record effect (POLICY.md §3), as the reference does with a human's message at the
 * start of the next graph run (agent/tools/utils.py:310-358).  0 ok, -1 not allowed. */
int orc_inject_action(const orc_table *tb, orc_room *r, int player_id, int choice) {
    const orc_phase *ph = &tb->ph[r->phase];
    const int n = r->n;
    if (player_id < 1 || player_id > n || ph->completion != COMP_ACTION) return -1;
    const int i = player_id - 1;
    if (!is_target(tb, ph, r, i) || r->p[i][F_ACTED]) return -1;
    if (tb->pack == PACK_WW) {
        if (choice < 1 || choice > n || !r->p[choice - 1][W_ALIVE]) return -1;
    } else if (ph->act == ACT_TT_STATEMENTS ? choice != 1 : (choice < 1 || choice > 3)) return -1;
    uint8_t *f = r->p[i];
    f[F_ACTED] = 1; f[F_CHOICE] = (uint8_t)choice;
    switch (ph->act) {
    case ACT_DETECTIVE: r->det[choice - 1] = r->p[choice - 1][W_TEAM] == TEAM_WEREWOLVES ? 2 : 1; /* fallthrough */
    case ACT_WOLF_TARGET:
    case ACT_DOCTOR_PROTECT: f[W_SUB] = 1; f[W_TARGET] = (uint8_t)choice; break;
    case ACT_TT_STATEMENTS: f[T_SUBMITTED] = 1; break;
    case ACT_TT_LIE: f[T_LIE] = (uint8_t)choice; break;
    case ACT_TT_VOTE: f[T_VOTE] = (uint8_t)choice; f[T_HAS_VOTED] = 1; break;
    default: break;
    }
    return 0;
}

int orc_sizeof_room(void) { return (int)sizeof(orc_room); }
int orc_sizeof_table(void) { return (int)sizeof(orc_table); }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
