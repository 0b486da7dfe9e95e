// ge_kernels.inl — every kernel of libge_step.so (device code only; included by ge_step.hip inside its anonymous namespace):
// the step kernels (ge_step_kernel<KIND, LOWOCC, GENERIC, SINGLE>, ge_step_kernel_mixed), summary, fill, deal-cache clear, action
// injection.  The turn itself is ge_device.h, the record layout ge_layout.h.  game_engine_amd/_lib.py::kernel_source_hash covers
// exactly these three files and the Makefile's flags: what a committed counter profile (profiles/pmc_*.json) is tied to.
// LDS of a step block: the table image (DevTable's leading IMG_* bytes: [phase rows][ord8][nth8][spread8][tally64], the last
// three in the large-batch builds only), the restart template (large-batch), then one WaveLds per wavefront
// single-turn launches: a wavefront about to store its record issues ahead of the others (it leaves, and its slot goes to a new wavefront
// whose loads then start earlier): -0.6 % (profiles/r05_ab_k1_prio.txt; raising NEW wavefronts' priority until their loads are out costs 3 - 10 %)
#ifndef GE_STORE_PRIO
#define GE_STORE_PRIO 3
#endif
constexpr uint32_t LDS_ROWS = sizeof(DevRow) * GE_MAX_PHASES;
constexpr uint32_t LDS_ORD8 = 1024;
static_assert(LDS_ROWS == IMG_ORD8 && LDS_ROWS + LDS_ORD8 == IMG_NTH8, "LDS image offsets");


struct SegDev {
    uint32_t *base;            // planes of this segment
    uint64_t rooms;            // real rooms
    uint64_t rooms_padded;     // multiple of 256: plane stride
    uint64_t first_global;     // global index of the segment's room 0
    uint32_t kind, n_players, nw, rounds;
    uint32_t phase0_idx, block_begin, table_idx, words;
    uint32_t human_mask, pad0;
    uint64_t local_first;      // index of the segment's room 0 inside the batch
    uint32_t init_words[12];   // the initial record (player_states_template, phase 0)
    alignas(16) uint32_t init_regs[20];    // the same in the kernels' register form (WWR::to_regs / TT::to_regs): the restart template, read with scalar loads
    uint32_t term_mask;        // bit r = table row r is terminal (no next_phase branch)
    uint32_t done0;            // two-truths: tt_done_mask of the initial record
    uint32_t *trace;           // GE_FLAG_TRACE: [turn in launch][rooms_padded] x 4 words, else null
    uint32_t *deal_side;       // Werewolf x 12: prepared role deals of the single-turn launches, [rooms_padded] x 2 words (else null); see run_ww
};

struct StepArgs {
    const uint32_t *turn_dev;  // launches replayed from a hipGraph: turn0 is relative to this device word (else null)
    unsigned long long *stamps; // GE_STAMPS diagnostic build: 4 segment sums + wave-turn count (else null)
    uint32_t n_seg, turn0, n_turns, seed_key, block_threads, restart, trace, lowocc;
    uint32_t cond_off;         // GENERIC builds: byte offset of the literal image (DevTable::cond_img) in a block's LDS, behind everything else
    uint32_t rooms_per_block;  // rooms a block steps: blockDim.x, or 32 of its 64 lanes (ge_step.hip launch_geometry: half-filled lone wavefronts)
    uint32_t block_begin[GE_MAX_SEGMENTS];
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// NT: non-temporal (streaming) accesses.  The step kernels touch every record once per launch; measured on the single-turn
// launches in their sustained regime (bench.py hbm_streaming, % of 8 TB/s, profiles/r03_ab_nontemporal.txt): Two-Truths x 4
// 59.4 -> 63.0 and Werewolf x 12 58.4 -> 60.0 with both loads and stores streaming (either alone: nothing); Werewolf x 8
// 57.8 -> 58.7 with streaming stores only (with both: 57.5).  Fused launches do not notice.  So the step kernels stream their
// stores, and their loads except for the 8-word Werewolf record (run_ww / run_tt); summary, fill and injection do not.
// Round 5 (profiles/r05_ab_ww8_nt_loads.txt): for that record it depends on where the state lives.  While it fits the 256 MiB
// Infinity Cache a plain load is better (1 M rooms 62.0 against 59.8 %, 8 M rooms = 256 MiB 82.5 against 74.9 %); once it streams from
// HBM the streaming load is (16 M rooms 74.2 -> 77.9 %, 33 M 70.9 -> 73.2 %), and in a mixed batch too (C5 share 55.4 -> 57.5 %).
// The other layouts turn over too (profiles/r05_ab_nt_others.txt): Two-Truths x 4 streaming at 24 MiB (61.2 against 59.9 %), plain from 96 MiB
// (75.7 against 69.7 %) to 252 MiB (85.8 against 72.8 %), streaming again beyond the cache; Werewolf x 12 level at 80 MiB, plain at 160 MiB
// (74.1 against 70.2 %), streaming beyond (400 MiB: 72.5 against 66.6 %).  So the large-batch single-turn kernels of the shipped games exist
// in both forms (the LD template argument; ge_step.hip record_loads picks by layout and size of the state), and a mixed batch's single-turn
// launches stream.  (A wave-uniform branch between two load sequences inside one kernel was tried first: the optimiser merges the arms'
// loads and drops the hint unless inline-asm statements keep them apart, and behind any such statement it fetches the launch's turn word
// with a vector load instead of a scalar one - 4 % of the launch.)
constexpr bool stream_loads(int ld, bool layout_default) { return ld == 0 ? layout_default : ld == 2; }
template <int WORDS, bool NT = false>
__device__ __forceinline__ void load_words(const uint32_t *base, uint64_t rooms_padded, uint64_t room, uint32_t *w) {
    constexpr int NP = (WORDS + 3) / 4;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        // global (address space 1), not flat, accesses: the base pointer was itself loaded from memory
        const char *plane = reinterpret_cast<const char *>(base) + plane_offset(rooms_padded, j);
        if (WORDS - 4 * j >= 4) {
            const auto *p = &((const __attribute__((address_space(1))) u32x4 *)(uintptr_t)plane)[room];
            const u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        } else {
            const auto *p = &((const __attribute__((address_space(1))) u32x2 *)(uintptr_t)plane)[room];
            const u32x2 v = NT ? __builtin_nontemporal_load(p) : *p;
            w[4 * j] = v.x; w[4 * j + 1] = v.y;
        }
    }
}

template <int WORDS, bool NT = false>
__device__ __forceinline__ void store_words(uint32_t *base, uint64_t rooms_padded, uint64_t room, const uint32_t *w) {
    constexpr int NP = (WORDS + 3) / 4;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        char *plane = reinterpret_cast<char *>(base) + plane_offset(rooms_padded, j);
        if (WORDS - 4 * j >= 4) {
            u32x4 v; v.x = w[4 * j]; v.y = w[4 * j + 1]; v.z = w[4 * j + 2]; v.w = w[4 * j + 3];
            auto *p = &((__attribute__((address_space(1))) u32x4 *)(uintptr_t)plane)[room];
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        } else {
            u32x2 v; v.x = w[4 * j]; v.y = w[4 * j + 1];
            auto *p = &((__attribute__((address_space(1))) u32x2 *)(uintptr_t)plane)[room];
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

// event trace (GE_FLAG_TRACE): one 16-byte record per room and turn of the launch, coalesced
__device__ __forceinline__ void store_event(uint32_t *trace, uint64_t rooms_padded, uint32_t t, uint64_t room, uint32_t turn,
                                            uint32_t p, uint32_t q, uint32_t restarted, uint32_t newly, uint64_t choice) {
    u32x4 v;
    v.x = turn; v.y = p | (q << 8) | (restarted << 16) | (newly << 20);
    v.z = (uint32_t)choice; v.w = (uint32_t)(choice >> 32);
    // streaming store: written once, read by the host (traced 1 M x 8: +6.0 -> +4.6 % over the untraced turn, profiles/r03_ab_nontemporal.txt)
    __builtin_nontemporal_store(v, &((__attribute__((address_space(1))) u32x4 *)(uintptr_t)trace)[(uint64_t)t * rooms_padded + room]);
}

// The large-batch turn loops take the restart template (SegDev::init_regs) from a copy in the block's LDS when a room
// restarts (uniform-address reads inside the restart branch).  A/B on MI355X, us/turn at 64 fused turns
// (profiles/r03_ab_restart_template.txt): template in scalar registers across the loop - spills to VGPR lanes; scalar-cache
// load inside the branch 8.12 / 21.28 / 4.88 (1 M x 8 / 2 M x 12 / 1 M Two-Truths x 4); LDS copy 7.90 / 21.09 / 4.68.
constexpr uint32_t LDS_S0 = 128;       // 20 words of init_regs, padded

// Fills the block's LDS tables: DevTable starts with rows | ord8 | nth8 | spread8 | tally64 in the LDS order, so the first
// N16 16-byte elements are one linear copy (64 = the phase rows, 128 = + ord8, 256 = + nth8, 448 = + the vote tables), one
// element per thread and pass.  All passes' loads are issued before the first LDS write (a load -> wait -> write loop
// serialises one L2 round trip per pass in front of every wavefront of a single-turn launch); the large-batch builds also
// copy the restart template behind the image.
template <uint32_t N16, bool WITH_S0, int GENERIC = false, bool SPLIT = false>
__device__ __forceinline__ void load_rows(DevRow *rows, const DevTable *tables, uint32_t table_idx, const SegDev *sg, uint32_t cond_off = 0u) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(tables + table_idx);
    if (GENERIC) {
        // the generic rows' literal image (ge_layout.h CondLit): cond_n16 elements, a few hundred bytes for a typical DSL
        const u32x4 *ci = reinterpret_cast<const u32x4 *>(tables[table_idx].cond_img);
        u32x4 *cd = reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(rows) + cond_off);
        const uint32_t n16 = __builtin_amdgcn_readfirstlane(tables[table_idx].cond_n16);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) cd[i] = ci[i];
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(rows);
    const uint32_t bd = blockDim.x, tid = threadIdx.x;
    // element i of the image -> where it goes in LDS: the phase rows' halves into two arrays (ge_device.h lds_row), the rest as is
    auto at = [](uint32_t i) -> uint32_t { return (SPLIT && GE_ROWS_SPLIT && i < 2u * GE_MAX_PHASES) ? ((i & 1u) * GE_MAX_PHASES + (i >> 1)) : i; };
    // the restart template behind the image (large-batch turn loops; a single-turn build reads it through the scalar cache)
    const u32x4 t0 = WITH_S0 ? reinterpret_cast<const u32x4 *>(sg->init_regs)[tid < 5u ? tid : 4u] : u32x4{0u, 0u, 0u, 0u};
    if (bd >= 512u) {                                          // wave-uniform: a single-turn launch's larger block - one pass (N16 <= 448)
        const u32x4 t = src[tid < N16 ? tid : N16 - 1u];
        if (tid < N16) dst[at(tid)] = t;
    } else if (bd == 256u) {                                   // the block size of every large batch
        constexpr uint32_t P = (N16 + 255u) / 256u;
        u32x4 t[P];
#pragma unroll
        for (uint32_t p = 0; p < P; p++) { const uint32_t i = p * 256u + tid; t[p] = src[i < N16 ? i : N16 - 1u]; }   // loads are not predicated (clamped index)
#pragma unroll
        for (uint32_t p = 0; p < P; p++) { const uint32_t i = p * 256u + tid; if (i < N16) dst[at(i)] = t[p]; }
    } else {
        for (uint32_t base = 0; base < N16; base += bd) {
            const uint32_t i = base + tid;
            if (i < N16) dst[at(i)] = src[i];
        }
    }
    if (WITH_S0 && tid < 5u) dst[IMG_END / 16u + tid] = t0;
    __syncthreads();
}

// The restart template (SegDev::init_regs) through the scalar cache.  The address is an opaque scalar: the loads are
// s_load_dwordx8/x16, and they stay where they are written - inside the restart branch of a large-batch turn loop they are
// not hoisted out of it: up to 19 scalar registers live across the whole loop spill to VGPR lanes there (v_writelane /
// v_readlane per restart), while a scalar-cache hit per restart costs a wavefront with 5-7 neighbours on its SIMD nothing.
template <int N>
__device__ __forceinline__ void load_init_regs(const SegDev &sg, uint32_t *ir) {
    // whole 16-byte groups (init_regs has 20 words): a ragged tail would be fetched with vector loads.  The opaque zero
    // offset is what keeps the loads in place (an address the optimiser cannot prove loop-invariant)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    uint32_t zero = 0;
    asm volatile("" : "+s"(zero));
    const u32x4_t *ip = reinterpret_cast<const u32x4_t *>(sg.init_regs) + zero;
#pragma unroll
    for (int j = 0; j < (N + 3) / 4; j++) {
        const u32x4_t v = ip[j];
        if (4 * j < N) ir[4 * j] = __builtin_amdgcn_readfirstlane(v.x);
        if (4 * j + 1 < N) ir[4 * j + 1] = __builtin_amdgcn_readfirstlane(v.y);
        if (4 * j + 2 < N) ir[4 * j + 2] = __builtin_amdgcn_readfirstlane(v.z);
        if (4 * j + 3 < N) ir[4 * j + 3] = __builtin_amdgcn_readfirstlane(v.w);
    }
}

// a 12-player game lasts ~65 turns against ~40 for 8 players: deals are prepared half as often there
// (profiles/r03_ab_deal_period_12.txt: every 16th / 32nd / 64th turn = 18.37 / 18.21 / 19.53 us per turn at 2 M x 12)
template <int NB> constexpr uint32_t deal_period() { return NB <= 8 ? GE_DEAL_PERIOD : 2u * GE_DEAL_PERIOD; }

template <int NB, bool LOWOCC, int GENERIC, bool SINGLE, int LD = 0>
__device__ __forceinline__ void run_ww(const SegDev *__restrict__ sgp, const StepArgs &a, DevRow *rows, void *lw,
                                       uint8_t *nth8, const DevTable *__restrict__ tables, uint64_t room_in) {
    const SegDev &sg = *sgp;
    using L = WWLayout<NB>;
    using B = WwBuild<NB, LOWOCC, SINGLE>;
    // lanes past the end of the segment stay in the wavefront (the action queue is a wave-wide
    // collective); they shadow room 0 with no actions and store nothing
    const bool valid = room_in < sg.rooms;
    const uint64_t room = valid ? room_in : 0;
    uint32_t w[L::WORDS];
    load_words<L::WORDS, stream_loads(LD, NB > 8)>(sg.base, sg.rooms_padded, room, w);   // in flight while the block fills its LDS tables; streaming: see load_words
    // the slot -> player table of the action queue sits right behind the phase rows (step_lds_bytes)
    const uint32_t *ord8 = reinterpret_cast<const uint32_t *>(reinterpret_cast<unsigned char *>(rows) + LDS_ROWS);
    const DevRow *rows_t = rows;
    const uint8_t *nth8_t = nth8;
    const unsigned char *cond_t = reinterpret_cast<const unsigned char *>(rows) + a.cond_off;
    if (single_global<LOWOCC, SINGLE>(0)) {
        // GE_SINGLE_GLOBAL: a single-turn launch reads the tables where they lie in global memory (DevTable leads with the LDS image)
        // instead of every block copying them into LDS behind a barrier first (ge_device.h: which builds do)
        const unsigned char *img = reinterpret_cast<const unsigned char *>(tables + sg.table_idx);
        rows_t = reinterpret_cast<const DevRow *>(img);
        ord8 = reinterpret_cast<const uint32_t *>(img + IMG_ORD8);
        nth8_t = img + IMG_NTH8;
        cond_t = reinterpret_cast<const unsigned char *>(tables[sg.table_idx].cond_img);
    } else {
        load_rows<B::TABLE ? IMG_END / 16u : B::ORD ? 128u : 64u, !LOWOCC && !SINGLE, GENERIC, !LOWOCC && !SINGLE>(rows, tables, sg.table_idx, sgp, a.cond_off);
    }
    WWR<NB> s;
    uint32_t cache;
    if (NB <= 8 && !SINGLE) {
        // one opaque value per word: as elements of the loaded 16-byte vectors the packed predicate words would stay
        // <2 x i32> values through the turn loop's phis, and the 64-bit register tuples that makes cost a v_mov_b64
        // per pair and turn at the loop's back edge (the lone-wavefront build pays a full issue slot for each)
#pragma unroll
        for (int j = 0; j < L::WORDS; j++) asm volatile("" : "+v"(w[j]));
    }
    ww_load_regs<NB>(w, s, cache);                            // N <= 8: the record is the register form (ge_layout.h)
    const uint32_t ALL = (1u << sg.n_players) - 1u;
    const uint32_t rk = room_key_from(a.seed_key, sg.first_global + room);
    // the fresh room a finished one is recycled into: wave-uniform, already in register form (SegDev::init_regs).  The
    // lone-wavefront build keeps it in scalar registers across the turn loop; the large-batch builds fetch it when a room restarts
    auto fresh_room = [&]() {
        uint32_t ir[20];
        if (!LOWOCC && !SINGLE) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(rows) + IMG_END / 16u;
#pragma unroll
            for (int j = 0; j < (WWR<NB>::NREGS + 3) / 4; j++) { const u32x4 v = p[j]; ir[4 * j] = v.x; ir[4 * j + 1] = v.y; ir[4 * j + 2] = v.z; ir[4 * j + 3] = v.w; }
        } else {
            load_init_regs<WWR<NB>::NREGS>(sg, ir);
        }
        WWR<NB> s0;
        s0.from_regs(ir);
        return s0;
    };
    const uint32_t term_mask = __builtin_amdgcn_readfirstlane(sg.term_mask);
    const uint32_t turn0 = a.turn0 + (a.turn_dev ? *a.turn_dev : 0u);   // uniform (scalar load)
    CondShape cs = {0u, 0u, 0u, 0u, 0u};
    if (GENERIC) {
        const DevTable &tb = tables[sg.table_idx];
        cs = CondShape{(uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_shape), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_g[0]), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_g[1]),
                       (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_fields[0]), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_fields[1])};
    }
    // Role deals are keyed by (room, game index), so they can be prepared before the turn that applies them.  Entering
    // the role-assignment phase is rare per room (once a game) but in a wavefront of 64 rooms some room does it on ~80 %
    // of the turns; instead of running the deal for that one lane, every GE_DEAL_PERIOD-th turn all lanes without a
    // prepared deal get their next one together (ww_turn, in an LDS wait shadow; a game is longer than the period).
    // N <= 8: the record carries a prepared deal across launches, so this works for any number of turns per launch
    // (the period then counts absolute turns); Werewolf x 12 records have no spare bits - only launches of >= 16 turns
    // deal ahead there.
    Deal deal;
    deal_from_cache<NB, B::DEAL_FORM>(cache, s, ALL, deal);
    const bool ahead = NB <= 8 || a.n_turns >= 16u;
    const uint32_t deal_phase = NB <= 8 ? turn0 : 0u;
    uint32_t tk = turn_key(rk, turn0);                        // this turn's key; ww_turn leaves the next turn's (computed in an LDS wait shadow)
    Stamps stamps;
    if (GE_STAMPS) { stamps.start(); stamps.clocks_start(); }
    const WwCtx ctx = {rows_t, CondCtx{cond_t, cs}, lw, nth8_t, ord8, valid, sg.n_players, sg.nw, sg.phase0_idx, rk, sg.human_mask, term_mask};
    if constexpr (SINGLE) {
        // one turn, no loop: the row is fetched after the restart decision (terminal rows are a bit mask), nothing is
        // prepared for a next turn
        uint32_t restarted = 0;
        if (a.restart && ((term_mask >> s.phase) & 1u)) {        // recycle a finished room
            const uint32_t g = s.games;
            s = fresh_room();
            s.games = g < 0xFFFFu ? g + 1u : g;
            restarted = 1;
        }
        DevRow row = lds_row<!LOWOCC && !SINGLE>(rows_t, s.phase);
        const uint32_t p = s.phase;
        uint32_t ev_newly = 0;
        uint64_t ev_choice = 0;
        const bool trace = a.trace != 0u;
        // Werewolf x 12 has no room in its record for a prepared deal (ge_layout.h), and a wavefront of single-turn launches paid
        // a whole deal (~130 vector instructions) whenever one of its 64 rooms was assigned roles - on ~2 of 3 turns, 8 % of the
        // launch (profiles/r04_ab_deal12_upper_bound.txt).  So those deals live in a side plane beside the records: {DealPk, game
        // index | DEAL_VALID} per room, a cache of a pure function of (seed, global room, game index) whose tag is the whole game
        // index - a stale entry (in `deal.gv`: the game it was made for) can only miss, never mislead.  Every deal_period-th turn all lanes load their entry, the ones
        // without a deal for their next game compute it together, and all store; on the other turns only the lanes whose row
        // has a branch into a role assignment load theirs (a few scattered 8-byte loads per wavefront, issued here, used at the
        // end of the turn).  A miss deals on the spot (ww_apply_effect), as before.
        const bool side = NB > 8 && sg.deal_side != nullptr;                            // wave-uniform
        const bool refill = side && (turn0 & (deal_period<NB>() - 1u)) == 0u;
        auto *side_p = &((__attribute__((address_space(1))) u32x2 *)(uintptr_t)sg.deal_side)[room];
        if (NB > 8 && side) {
            uint32_t x = ((row.r3 >> 5) & 0x07070707u) ^ (0x01010101u * (uint32_t)EFF_ASSIGN_ROLES);   // a zero byte = a branch whose target assigns roles
            const bool may_assign = ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;
            if (refill || may_assign) {
                // taken as it is: the tag is compared where the deal is used (ww_prepare_deal, ww_apply_effect), at the end of the
                // turn - a compare here would make the wavefront wait for this load before it has done anything else
                const u32x2 v = *side_p;
                deal.a = v.x;
                deal.gv = v.y;
            }
        }
        const bool deal_now = NB <= 8 ? (turn0 & (GE_DEAL_PERIOD - 1u)) == 0u : refill;   // wave-uniform
        ww_turn<NB, LOWOCC, GENERIC, true>(s, row, ctx, turn0, tk, trace, deal, deal_now, ev_newly, ev_choice, nullptr);
        if (NB > 8 && refill && valid) {
            u32x2 v; v.x = deal.a; v.y = deal.gv == (deal_next_game<NB>(s) | DEAL_VALID) ? deal.gv : 0u;
            *side_p = v;
        }
        if (trace && valid) store_event(sg.trace, sg.rooms_padded, 0u, room, turn0, p, s.phase, restarted, ev_newly, ev_choice);
    } else {
        DevRow row = lds_row<!LOWOCC && !SINGLE>(rows_t, s.phase);
        WWR<NB> s0;
        DevRow row0 = row;
        if (LOWOCC) { s0 = fresh_room(); row0 = lds_row<!LOWOCC && !SINGLE>(rows, sg.phase0_idx); }
        // the turn loop; the lone-wavefront build compiles it once per trace setting: the event-trace branches (two per turn,
        // both wave-uniform and almost always taken) cost a lone wavefront an instruction-fetch bubble each
        auto turns = [&](auto trace_c) {
            constexpr bool KNOWN = decltype(trace_c)::value != 2;
            const bool trace = KNOWN ? decltype(trace_c)::value == 1 : a.trace != 0u;
            for (uint32_t t = 0; t < a.n_turns; t++) {
                uint32_t restarted = 0;
                if (a.restart && ((row.r0 >> 11) & 7u) == 0u) {      // recycle a finished room
                    const uint32_t g = s.games;
                    s = LOWOCC ? s0 : fresh_room();
                    s.games = g < 0xFFFFu ? g + 1u : g;
                    row = LOWOCC ? row0 : lds_row<!LOWOCC && !SINGLE>(rows, ctx.phase0_idx);
                    restarted = 1;
                }
                const uint32_t p = s.phase;
                uint32_t ev_newly = 0;
                uint64_t ev_choice = 0;
                const bool deal_now = ahead && ((deal_phase + t) & (deal_period<NB>() - 1u)) == 0u;    // wave-uniform
                ww_turn<NB, LOWOCC, GENERIC, false>(s, row, ctx, turn0 + t, tk, trace, deal, deal_now, ev_newly, ev_choice, (GE_STAMPS && a.stamps) ? &stamps : nullptr);
                if (trace && valid) store_event(sg.trace, sg.rooms_padded, t, room, turn0 + t, p, s.phase, restarted, ev_newly, ev_choice);
            }
        };
        if (B::TPL_TRACE) {                                       // (two copies of the loop cost the large-batch build registers)
            if (a.trace) turns(std::integral_constant<int, 1>{}); else turns(std::integral_constant<int, 0>{});
        } else {
            turns(std::integral_constant<int, 2>{});
        }
    }
    if (GE_STAMPS && a.stamps && (threadIdx.x & 63u) == 0u) {
        stamps.mark(3);
        for (int k = 0; k < 4; k++) atomicAdd(a.stamps + k, stamps.acc[k]);
        atomicAdd(a.stamps + 4, (unsigned long long)a.n_turns);
        atomicAdd(a.stamps + 5, (unsigned long long)(__builtin_amdgcn_s_memtime() - stamps.m0));       // a wavefront's life in shader cycles ...
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(a.stamps + 6, (unsigned long long)(r1 - stamps.r0));                                 // ... and in 100 MHz ticks
        if (GE_STAMPS == 2 && a.n_turns >= 256u) {
            // the launch's timeline: per wavefront {start, end} on the chip-wide 100 MHz clock and where it ran (HW_ID, XCC_ID)
            unsigned long long *log = a.stamps + 8 + 4ull * (((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
            log[0] = stamps.r0; log[1] = r1;
            log[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            log[3] = __builtin_amdgcn_s_memtime() - stamps.m0;
        }
    }
    if (!valid) return;
#if GE_STORE_PRIO
    if (SINGLE) __builtin_amdgcn_s_setprio(GE_STORE_PRIO);    // a finished wavefront leaves: its slot goes to a new one, whose loads then start earlier
#endif
    ww_store_regs<NB>(s, deal_to_cache<NB, B::DEAL_FORM>(deal, s), w);
    store_words<L::WORDS, true>(sg.base, sg.rooms_padded, room, w);
}

// Two-Truths: which builds route bot actions through the wavefront queue.  Many wavefronts per SIMD:
// always (x4: 5.35 -> 4.87 us/turn at 1M rooms).  Lone wavefront: the two extra LDS round trips cost more
// than the per-lane loop saves for few players (x4: 0.91 vs 1.03 us/turn at 65 536 rooms).
#ifndef GE_TT_LOW_QUEUE_MIN
#define GE_TT_LOW_QUEUE_MIN 5
#endif
constexpr bool tt_uses_queue(int nb, bool lowocc) { return !lowocc || nb >= GE_TT_LOW_QUEUE_MIN; }

template <int NB, bool LOWOCC, int GENERIC, bool SINGLE, int LD = 0>
__device__ __forceinline__ void run_tt(const SegDev *__restrict__ sgp, const StepArgs &a, DevRow *rows, void *lw, uint8_t *nth8,
                                       const DevTable *__restrict__ tables, uint64_t room_in) {
    constexpr bool QUEUE = tt_uses_queue(NB, LOWOCC);
    const SegDev &sg = *sgp;
    using L = TTLayout<NB>;
    const bool valid = room_in < sg.rooms;
    const uint64_t room = valid ? room_in : 0;
    uint32_t w[L::WORDS];
    load_words<L::WORDS, stream_loads(LD, true)>(sg.base, sg.rooms_padded, room, w);
    const DevRow *rows_t = rows;
    const uint8_t *nth8_t = nth8;
    const unsigned char *cond_t = reinterpret_cast<const unsigned char *>(rows) + a.cond_off;
    if (single_global<LOWOCC, SINGLE>(2)) {                    // see run_ww
        const unsigned char *img = reinterpret_cast<const unsigned char *>(tables + sg.table_idx);
        rows_t = reinterpret_cast<const DevRow *>(img);
        nth8_t = img + IMG_NTH8;
        cond_t = reinterpret_cast<const unsigned char *>(tables[sg.table_idx].cond_img);
    } else {
        load_rows<(QUEUE && !LOWOCC) ? 256u : 64u, QUEUE && !LOWOCC && !SINGLE, GENERIC, !LOWOCC && !SINGLE>(rows, tables, sg.table_idx, sgp, a.cond_off);
    }
    TT<NB> s;
    L::unpack(w, s);
    const uint32_t rk = room_key_from(a.seed_key, sg.first_global + room);
    // the restart template, already unpacked (see run_ww)
    auto fresh_room = [&]() {
        uint32_t ir[20];
        if (QUEUE && !LOWOCC && !SINGLE) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(rows) + IMG_END / 16u;
#pragma unroll
            for (int j = 0; j < (TT<NB>::NREGS + 3) / 4; j++) { const u32x4 v = p[j]; ir[4 * j] = v.x; ir[4 * j + 1] = v.y; ir[4 * j + 2] = v.z; ir[4 * j + 3] = v.w; }
        } else {
            load_init_regs<TT<NB>::NREGS>(sg, ir);
        }
        TT<NB> s0;
        s0.from_regs(ir);
        return s0;
    };
    const uint32_t done0 = __builtin_amdgcn_readfirstlane(sg.done0);
    const uint32_t term_mask = __builtin_amdgcn_readfirstlane(sg.term_mask);
    const uint32_t turn0 = a.turn0 + (a.turn_dev ? *a.turn_dev : 0u);
    CondShape cs = {0u, 0u, 0u, 0u, 0u};
    if (GENERIC) {
        const DevTable &tb = tables[sg.table_idx];
        cs = CondShape{(uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_shape), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_g[0]), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_g[1]),
                       (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_fields[0]), (uint32_t)__builtin_amdgcn_readfirstlane(tb.cond_fields[1])};
    }
    const CondCtx cc = {cond_t, cs};
    uint32_t done = tt_done_mask<NB>(s.rounds, sg.rounds);   // who has spoken all agreed rounds (ge_device.h)
    if constexpr (SINGLE) {
        uint32_t restarted = 0;
        if (a.restart && ((term_mask >> s.phase) & 1u)) {
            const uint32_t g = s.games;
            s = fresh_room();
            s.games = g < 0xFFFFu ? g + 1u : g;
            done = done0;
            restarted = 1;
        }
        DevRow row = lds_row<!LOWOCC && !SINGLE>(rows_t, s.phase);
        const uint32_t p = s.phase;
        uint32_t ev_newly = 0;
        uint64_t ev_choice = 0;
        tt_turn<NB, QUEUE, !LOWOCC, GENERIC, true>(s, done, row, rows_t, cc, lw, nth8_t, valid, sg.n_players, sg.rounds, sg.phase0_idx, rk, turn0, a.trace != 0u, sg.human_mask, term_mask, ev_newly, ev_choice);
        if (a.trace && valid) store_event(sg.trace, sg.rooms_padded, 0u, room, turn0, p, s.phase, restarted, ev_newly, ev_choice);
    } else {
        DevRow row = lds_row<!LOWOCC && !SINGLE>(rows_t, s.phase);
        TT<NB> s0;
        DevRow row0 = row;
        if (LOWOCC) { s0 = fresh_room(); row0 = lds_row<!LOWOCC && !SINGLE>(rows, sg.phase0_idx); }
        for (uint32_t t = 0; t < a.n_turns; t++) {
            uint32_t restarted = 0;
            if (a.restart && ((row.r0 >> 11) & 7u) == 0u) {
                const uint32_t g = s.games;
                s = LOWOCC ? s0 : fresh_room();
                s.games = g < 0xFFFFu ? g + 1u : g;
                row = LOWOCC ? row0 : lds_row<!LOWOCC && !SINGLE>(rows, sg.phase0_idx);
                done = done0;
                restarted = 1;
            }
            const uint32_t p = s.phase;
            uint32_t ev_newly = 0;
            uint64_t ev_choice = 0;
            tt_turn<NB, QUEUE, !LOWOCC, GENERIC, false>(s, done, row, rows, cc, lw, nth8, valid, sg.n_players, sg.rounds, sg.phase0_idx, rk, turn0 + t, a.trace != 0u, sg.human_mask, term_mask, ev_newly, ev_choice);
            if (a.trace && valid) store_event(sg.trace, sg.rooms_padded, t, room, turn0 + t, p, s.phase, restarted, ev_newly, ev_choice);
        }
    }
    if (!valid) return;
#if GE_STORE_PRIO
    if (SINGLE) __builtin_amdgcn_s_setprio(GE_STORE_PRIO);
#endif
    L::pack(s, w);
    store_words<L::WORDS, true>(sg.base, sg.rooms_padded, room, w);
}

// One launch advances every room of every segment by a.n_turns turns.  Blocks are
// segment-homogeneous (segments are padded to whole blocks), so a mixed Werewolf /
// Two-Truths batch diverges per block, never inside a wavefront.  Segment descriptors live in
// device memory and are read with a block-uniform index (scalar loads): indexing the kernel
// arguments dynamically would push them through scratch.
template <int KIND, bool LOWOCC, int GENERIC, bool SINGLE = false, int LD = 0>
__device__ __forceinline__ void run_kind(const SegDev *__restrict__ sg, const StepArgs &a, DevRow *rows, void *lw,
                                         uint8_t *nth8, const DevTable *__restrict__ tables, uint64_t room) {
    if (KIND == K_WW8) run_ww<8, LOWOCC, GENERIC, SINGLE, LD>(sg, a, rows, lw, nth8, tables, room);
    else if (KIND == K_WW12) run_ww<12, LOWOCC, GENERIC, SINGLE, LD>(sg, a, rows, lw, nth8, tables, room);
    else if (KIND == K_TT4) run_tt<4, LOWOCC, GENERIC, SINGLE, LD>(sg, a, rows, lw, nth8, tables, room);
    else if (KIND == K_TT8) run_tt<8, LOWOCC, GENERIC, SINGLE, LD>(sg, a, rows, lw, nth8, tables, room);
    else run_tt<12, LOWOCC, GENERIC, SINGLE, LD>(sg, a, rows, lw, nth8, tables, room);
}

// LDS of a step block is sized at launch (a 64-room block must not pay for four wavefronts' queues, or
// LDS, not registers, caps the wavefronts per CU)
constexpr uint32_t LDS_NTH8 = IMG_END - IMG_NTH8;             // nth8 + the vote tables
static_assert(offsetof(DevTable, ord8) == IMG_ORD8 && offsetof(DevTable, nth8) == IMG_NTH8 && offsetof(DevTable, spread8) == IMG_SPREAD8 &&
              offsetof(DevTable, tally64) == IMG_TALLY && offsetof(DevTable, n_phases) == IMG_END, "DevTable leads with the LDS image");
static_assert(sizeof(WaveLds) % 16 == 0 && sizeof(WaveLdsLow) % 16 == 0 && LDS_ROWS % 16 == 0 && LDS_ORD8 % 16 == 0, "LDS sections stay 16-byte aligned");

inline uint32_t step_lds_bytes(bool queue, bool lowocc, uint32_t block_threads) {
    if (!queue) return LDS_ROWS;                              // Two-Truths N <= 4: phase rows only
    return LDS_ROWS + LDS_ORD8 + (lowocc ? 0u : LDS_NTH8 + LDS_S0) + (uint32_t)(lowocc ? sizeof(WaveLdsLow) : sizeof(WaveLds)) * (block_threads / 64u);
}
// (GENERIC builds: the literal image of the table's generic rows follows at StepArgs::cond_off = this size, cond_bytes more)

extern __shared__ __align__(16) unsigned char ge_lds[];

// single-kind batch (the benchmark configurations): one instantiation per record layout, so each
// gets its own register allocation
// GENERIC (0 / 1; Two-Truths fused builds also 2 / 3 = the table's shape is 1 x 1 / 1 x 2, ge_device.h tt_cond_generic): some row of the table has a generic target condition (DevCond); single-game batches get both forms of those
// builds as well, a mixed batch with a generic table runs the large-batch form at every size
// Minimum wavefronts per SIMD asked of the register allocator for the large-batch Werewolf builds (tuning constants;
// tools/ab_switches.sh builds other values).  Werewolf x 12: 7 = 72 VGPRs, no scratch (round 2 held it to 6 = 80 VGPRs with a
// 2-register spill); Werewolf x 8 needs 62 VGPRs = 8 wavefronts per SIMD without being asked.
#ifndef GE_WW12_WAVES
#define GE_WW12_WAVES 7
#endif
#ifndef GE_WW8_WAVES
#define GE_WW8_WAVES 1
#endif
// the large-batch builds for tables with generic target conditions (every layout); 7 would mean at most 72 VGPRs
#ifndef GE_GENERIC_WAVES
#define GE_GENERIC_WAVES 1
#endif
// SINGLE: the launch is one turn (a.n_turns == 1) (run_ww / run_tt)
template <int KIND, bool LOWOCC, int GENERIC, bool SINGLE, int LD = 0>
__device__ __forceinline__ void step_body(const SegDev *__restrict__ segs, const DevTable *__restrict__ tables, const StepArgs &a) {
    constexpr bool WWK = KIND == K_WW8 || KIND == K_WW12 || tt_uses_queue(KIND == K_TT4 ? 4 : KIND == K_TT8 ? 8 : 12, LOWOCC);   // uses the action queue
    DevRow *rows = reinterpret_cast<DevRow *>(ge_lds);
    uint8_t *nth8 = ge_lds + LDS_ROWS + LDS_ORD8;
    auto *wl = reinterpret_cast<typename WaveLdsOf<LOWOCC>::type *>(ge_lds + LDS_ROWS + LDS_ORD8 + (LOWOCC ? 0u : LDS_NTH8 + LDS_S0));
    // (a lane past rooms_per_block holds no room: it shadows room 0 like a lane past the end of the segment)
    const uint64_t room = (!LOWOCC || threadIdx.x < a.rooms_per_block) ? (uint64_t)blockIdx.x * (LOWOCC ? a.rooms_per_block : blockDim.x) + threadIdx.x : ~0ull;
    run_kind<KIND, LOWOCC, GENERIC, SINGLE, LD>(segs, a, rows, WWK ? &wl[threadIdx.x >> 6] : nullptr, nth8, tables, room);
}
// LD (the large-batch single-turn kernels of the shipped games only): 1 / 2 = the record with plain / streaming loads instead of the layout's
// default - chosen at launch by the size of the state (load_words, ge_step.hip record_loads)
template <int KIND, bool LOWOCC, int GENERIC = false, bool SINGLE = false, int LD = 0>
__global__ void __launch_bounds__(SINGLE ? 1024 : 256, SINGLE ? 8 : (!LOWOCC && GENERIC) ? GE_GENERIC_WAVES : (KIND == K_WW12 && !LOWOCC) ? GE_WW12_WAVES : (KIND == K_WW8 && !LOWOCC) ? GE_WW8_WAVES : 1) ge_step_kernel(const SegDev *__restrict__ segs, const DevTable *__restrict__ tables, const StepArgs a) {
    step_body<KIND, LOWOCC, GENERIC, SINGLE, LD>(segs, tables, a);
}

// mixed batch: several segments (games / player counts) in one launch.  SINGLE: the launch is one turn - each kind's single-turn
// form (no turn loop, the restart template through the scalar cache, Werewolf x 12 deals from the side plane), 8 wavefronts per SIMD
template <bool LOWOCC, int GENERIC = false, bool SINGLE = false>
__global__ void __launch_bounds__(256, SINGLE ? 8 : !LOWOCC ? (GENERIC ? GE_GENERIC_WAVES : GE_WW12_WAVES) : 1) ge_step_kernel_mixed(const SegDev *__restrict__ segs, const DevTable *__restrict__ tables, const StepArgs a) {
    DevRow *rows = reinterpret_cast<DevRow *>(ge_lds);
    uint8_t *nth8 = ge_lds + LDS_ROWS + LDS_ORD8;
    auto *wl = reinterpret_cast<typename WaveLdsOf<LOWOCC>::type *>(ge_lds + LDS_ROWS + LDS_ORD8 + (LOWOCC ? 0u : LDS_NTH8 + LDS_S0));
    const uint32_t bid = blockIdx.x;
    uint32_t si = 0;
    for (uint32_t k = 1; k < a.n_seg; k++)
        if (bid >= a.block_begin[k]) si = k;
    si = __builtin_amdgcn_readfirstlane(si);
    const SegDev *sg = segs + si;
    const uint64_t room = (uint64_t)(bid - a.block_begin[si]) * blockDim.x + threadIdx.x;
    void *lw = &wl[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];    // wave-uniform: kept in a scalar register across the kind switch
    switch (sg->kind) {
    case K_WW8: run_kind<K_WW8, LOWOCC, GENERIC, SINGLE, (SINGLE && !LOWOCC) ? 2 : 0>(sg, a, rows, lw, nth8, tables, room); break;   // streaming record loads (load_words)
    case K_WW12: run_kind<K_WW12, LOWOCC, GENERIC, SINGLE>(sg, a, rows, lw, nth8, tables, room); break;
    case K_TT4: run_kind<K_TT4, LOWOCC, GENERIC, SINGLE>(sg, a, rows, lw, nth8, tables, room); break;
    case K_TT8: run_kind<K_TT8, LOWOCC, GENERIC, SINGLE>(sg, a, rows, lw, nth8, tables, room); break;
    default: run_kind<K_TT12, LOWOCC, GENERIC, SINGLE>(sg, a, rows, lw, nth8, tables, room); break;
    }
}

// ---- summary: per-room contributions -> wavefront shuffle reduce -> LDS -> one atomic per block
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int WORDS> __device__ __forceinline__ uint32_t fold_words(uint32_t h, const uint32_t *w) {
#pragma unroll
    for (int j = 0; j < WORDS; j++) h = mix32(h ^ w[j]);
    return h;
}

constexpr uint32_t SUM_CHUNKS = 16;

struct RoomStats { uint32_t finished, village, wolves, alive, end_turn, games; };

template <int NB> __device__ __forceinline__ RoomStats stats_ww(const uint32_t *w, const DevRow *rows, uint32_t *hist_score) {
    WW<NB> s; WWLayout<NB>::unpack(w, s);
    RoomStats r;
    r.finished = ((rows[s.phase].r0 >> 11) & 7u) == 0u;
    const uint32_t wv = __popc(s.alive & s.team_w);
    r.village = r.finished && wv == 0; r.wolves = r.finished && wv != 0;
    r.alive = __popc(s.alive); r.end_turn = s.end_turn; r.games = s.games;
    (void)hist_score;
    return r;
}
template <int NB> __device__ __forceinline__ RoomStats stats_tt(const uint32_t *w, const DevRow *rows, uint32_t n, uint32_t *hist_score) {
    TT<NB> s; TTLayout<NB>::unpack(w, s);
    RoomStats r;
    r.finished = ((rows[s.phase].r0 >> 11) & 7u) == 0u;
    r.village = 0; r.wolves = 0; r.alive = n; r.end_turn = s.end_turn; r.games = s.games;
#pragma unroll
    for (int i = 0; i < NB; i++) {                               // static indices: no scratch
        const uint32_t sc = (s.score[i / 4] >> (8 * (i % 4))) & 255u;
        if ((uint32_t)i < n) atomicAdd(&hist_score[sc < 15 ? sc : 15], 1u);
    }
    return r;
}

__global__ void __launch_bounds__(256) ge_summary_kernel(const StepArgs a, const SegDev *__restrict__ segs,
                                                         const DevTable *__restrict__ tables,
                                                         unsigned long long *__restrict__ out) {
    __shared__ DevRow rows[GE_MAX_PHASES];
    __shared__ uint32_t h_end[16], h_score[16];
    __shared__ unsigned long long acc[7];
    uint32_t si = 0;
    for (uint32_t k = 1; k < a.n_seg; k++)
        if (blockIdx.x >= a.block_begin[k]) si = k;
    si = __builtin_amdgcn_readfirstlane(si);
    const SegDev sg = segs[si];
    if (threadIdx.x < GE_MAX_PHASES) rows[threadIdx.x] = tables[sg.table_idx].rows[threadIdx.x];
    if (threadIdx.x < 16) { h_end[threadIdx.x] = 0; h_score[threadIdx.x] = 0; }
    if (threadIdx.x < 7) acc[threadIdx.x] = 0;
    __syncthreads();
    // each block walks SUM_CHUNKS consecutive 256-room chunks of its segment, so that the per-block
    // global atomics (a few dozen, all blocks on the same words) stay rare
    RoomStats r = {0, 0, 0, 0, 0, 0};
    uint64_t ck = 0;
    for (uint32_t c = 0; c < SUM_CHUNKS; c++) {
        const uint64_t room = ((uint64_t)(blockIdx.x - a.block_begin[si]) * SUM_CHUNKS + c) * blockDim.x + threadIdx.x;
        if (room >= sg.rooms) break;
        const uint64_t g = sg.first_global + room;
        const uint32_t h0 = mix32((uint32_t)g ^ mix32((uint32_t)(g >> 32) ^ 0xA5A5A5A5u));
        uint32_t h = h0;
        RoomStats q;
        switch (sg.kind) {
        case K_WW8: { uint32_t w[8]; load_words<8>(sg.base, sg.rooms_padded, room, w); q = stats_ww<8>(w, rows, h_score); w[7] &= WWLayout<8>::CHECKSUM_MASK7; h = fold_words<8>(h0, w); break; }   // (the prepared-deal cache is not state)
        case K_WW12: { uint32_t w[10]; load_words<10>(sg.base, sg.rooms_padded, room, w); q = stats_ww<12>(w, rows, h_score); h = fold_words<10>(h0, w); break; }
        case K_TT4: { uint32_t w[6]; load_words<6>(sg.base, sg.rooms_padded, room, w); q = stats_tt<4>(w, rows, sg.n_players, h_score); h = fold_words<6>(h0, w); break; }
        case K_TT8: { uint32_t w[8]; load_words<8>(sg.base, sg.rooms_padded, room, w); q = stats_tt<8>(w, rows, sg.n_players, h_score); h = fold_words<8>(h0, w); break; }
        default: { uint32_t w[12]; load_words<12>(sg.base, sg.rooms_padded, room, w); q = stats_tt<12>(w, rows, sg.n_players, h_score); h = fold_words<12>(h0, w); break; }
        }
        ck += (uint64_t)h | ((uint64_t)mix32(h ^ 0x5BD1E995u) << 32);
        if (q.finished) atomicAdd(&h_end[(q.end_turn >> 3) < 15 ? (q.end_turn >> 3) : 15], 1u);
        r.finished += q.finished; r.village += q.village; r.wolves += q.wolves; r.alive += q.alive;
        r.end_turn += q.finished ? q.end_turn : 0u; r.games += q.games;
    }
    const uint64_t v0 = wave_sum(r.finished), v1 = wave_sum(r.village), v2 = wave_sum(r.wolves);
    const uint64_t v3 = wave_sum(r.alive), v4 = wave_sum(r.end_turn), v5 = wave_sum(ck);
    const uint64_t v6 = wave_sum(r.games);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], v0); atomicAdd(&acc[1], v1); atomicAdd(&acc[2], v2);
        atomicAdd(&acc[3], v3); atomicAdd(&acc[4], v4); atomicAdd(&acc[5], v5); atomicAdd(&acc[6], v6);
    }
    __syncthreads();
    // out: [0] finished [1] village [2] wolves [3] alive [4] sum_end [5..20] end hist [21..36] score hist [37] checksum
    if (threadIdx.x < 5) atomicAdd(&out[threadIdx.x], acc[threadIdx.x]);
    if (threadIdx.x == 5) atomicAdd(&out[37], acc[5]);
    if (threadIdx.x == 6) atomicAdd(&out[38], acc[6]);
    if (threadIdx.x >= 64 && threadIdx.x < 80 && h_end[threadIdx.x - 64]) atomicAdd(&out[5 + threadIdx.x - 64], (unsigned long long)h_end[threadIdx.x - 64]);
    if (threadIdx.x >= 128 && threadIdx.x < 144 && h_score[threadIdx.x - 128]) atomicAdd(&out[21 + threadIdx.x - 128], (unsigned long long)h_score[threadIdx.x - 128]);
}

// ---- reset: every room record <- the DSL's initial record (player_states_template, phase 0)
__global__ void __launch_bounds__(256) ge_fill_kernel(const SegDev *__restrict__ segs, uint32_t n_seg) {
    for (uint32_t k = 0; k < n_seg; k++) {
        const SegDev &sg = segs[k];
        const int np = planes_of((int)sg.words);
        for (uint64_t room = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; room < sg.rooms_padded; room += (uint64_t)gridDim.x * blockDim.x) {
            if (sg.deal_side) { u32x2 z; z.x = 0u; z.y = 0u; ((__attribute__((address_space(1))) u32x2 *)(uintptr_t)sg.deal_side)[room] = z; }
            for (int j = 0; j < np; j++) {
                char *plane = reinterpret_cast<char *>(sg.base) + plane_offset(sg.rooms_padded, j);
                if ((int)sg.words - 4 * j >= 4) {
                    u32x4 v; v.x = sg.init_words[4 * j]; v.y = sg.init_words[4 * j + 1]; v.z = sg.init_words[4 * j + 2]; v.w = sg.init_words[4 * j + 3];
                    ((__attribute__((address_space(1))) u32x4 *)(uintptr_t)plane)[room] = v;
                } else {
                    u32x2 v; v.x = sg.init_words[4 * j]; v.y = sg.init_words[4 * j + 1];
                    ((__attribute__((address_space(1))) u32x2 *)(uintptr_t)plane)[room] = v;
                }
            }
        }
    }
}

// ---- the prepared-deal caches (Werewolf x 8: word 7 of the records, upper half; Werewolf x 12: the side plane) <- empty.  A cached deal is a function of
// (seed, global room index, game index); records copied in raw from somewhere else (ge_batch_state) may carry deals of
// another seed or room range, so ge_batch_set_turn - the call that completes a raw restore - drops them all.
__global__ void __launch_bounds__(256) ge_clear_deal_cache(const SegDev *__restrict__ segs, uint32_t n_seg) {
    for (uint32_t k = 0; k < n_seg; k++) {
        const SegDev &sg = segs[k];
        if (sg.deal_side)                                                     // Werewolf x 12: the side plane of prepared deals
            for (uint64_t room = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; room < sg.rooms_padded; room += (uint64_t)gridDim.x * blockDim.x) {
                u32x2 z; z.x = 0u; z.y = 0u;
                ((__attribute__((address_space(1))) u32x2 *)(uintptr_t)sg.deal_side)[room] = z;
            }
        if (sg.kind != K_WW8) continue;
        uint32_t *plane1 = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sg.base) + plane_offset(sg.rooms_padded, 1));
        for (uint64_t room = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; room < sg.rooms_padded; room += (uint64_t)gridDim.x * blockDim.x)
            plane1[4u * room + 3u] &= WWLayout<8>::CHECKSUM_MASK7;            // word 7 = plane 1, element word 3
    }
}

// ---- host-driven players: a batch of logged actions (ge_batch_inject_actions), one thread per distinct room.
// What the reference does with a human's message at the start of the next graph run
// (agent/tools/utils.py:310-358 -> bt:285-344) + the Referee's record effect (POLICY.md §3).
struct InjectArgs {
    const uint64_t *rooms;     // sorted by room (stable): action k of the sorted order
    const uint32_t *players, *choices;
    const uint32_t *group;     // group g = sorted actions [group[g], group[g + 1])
    int32_t *status;           // per sorted action
    uint32_t n_groups, n_seg;
};

// base predicate `base` of player bit `bit` (POLICY.md §3 numbering)
template <int NB> __device__ __forceinline__ bool ww_base(const WW<NB> &s, uint32_t base, uint32_t bit) {
    const uint32_t n2 = ~s.rb2;
    switch (base) {
    case 0: return s.alive & bit; case 1: return s.can_vote & bit; case 2: return s.revealed & bit; case 3: return s.secret & bit;
    case 4: return s.elig & bit; case 5: return s.sub & bit; case 6: return s.team_v & bit; case 7: return s.team_w & bit;
    case 8: return s.rb0 & ~s.rb1 & n2 & bit; case 9: return ~s.rb0 & s.rb1 & n2 & bit;
    case 10: return s.rb0 & s.rb1 & n2 & bit; default: return s.rb2 & ~s.rb1 & ~s.rb0 & bit;
    }
}
template <int NB> __device__ __forceinline__ bool tt_base(const TT<NB> &s, uint32_t base, uint32_t bit) {
    // selects, not a switch: the compiler turns a switch over five masks into a table in scratch memory
    uint32_t m = s.has_voted;
    m = sel32(base == 0u, s.speaker, m); m = sel32(base == 1u, s.submitted, m);
    m = sel32(base == 2u, s.revealed, m); m = sel32(base == 3u, s.can_vote, m);
    return (m & bit) != 0u;
}
// the row's j-th term as (base, negated): r1 encodes {word, shift} into the packed predicate words
__device__ __forceinline__ void row_term(const DevRow &row, uint32_t j, uint32_t fpw, uint32_t &base, bool &neg) {
    const uint32_t e = (row.r1 >> (8u * j)) & 255u;
    base = (e >> 5) * fpw + (e & 31u) / (32u / fpw);
    neg = (row.r0 >> (16u + j)) & 1u;
}

// the clause form of a condition for ONE player of an unpacked record (host-driven players' actions)
template <typename BASE, typename NUM>
__device__ __forceinline__ bool clauses_hold(const DevCond &c, BASE base_true, NUM num_value) {
    const uint32_t ncl = c.meta & 7u;
    if (ncl == 0u) return true;
    for (uint32_t k = 0; k < ncl; k++) {
        const uint32_t len = (c.meta >> (4 + 4 * k)) & 7u;
        bool all = true;
        for (uint32_t l = 0; l < len && all; l++) {
            const uint32_t w = c.lit[k][l];
            bool ok = false;
            if (((w >> 28) & 3u) == 1u) {
                for (uint32_t b = 0; b < 16u; b++)
                    if (((w >> b) & 1u) && base_true(b)) ok = true;
            } else {
                const uint32_t v = num_value((w >> 16) & 7u);
                ok = v >= (w & 0xFFu) && v <= ((w >> 8) & 0xFFu);
            }
            all = ok != (((w >> 30) & 1u) != 0u);
        }
        if (all) return true;
    }
    return false;
}

template <int NB> __device__ __forceinline__ int inject_ww(WW<NB> &s, const DevRow &row, const DevCond &cond, uint32_t n, uint32_t player, uint32_t choice) {
    using nib_t = typename WW<NB>::nib_t;
    if (player < 1 || player > n) return GE_ERR_ARG;
    if ((row.r0 & 3u) != COMP_ACTION) return GE_ERR_ARG;
    const uint32_t bit = 1u << (player - 1u);
    if (!(s.alive & bit)) return GE_ERR_ARG;
    if (row.r0 & ROW_GENERIC) {
        if (!clauses_hold(cond, [&](uint32_t b) { return ww_base<NB>(s, b, bit); },
                          [&](uint32_t) { return (uint32_t)(s.sel >> (4u * (player - 1u))) & 15u; })) return GE_ERR_ARG;
    }
    const uint32_t nt = (row.r0 & ROW_GENERIC) ? 0u : (row.r0 >> 8) & 7u;
    for (uint32_t j = 0; j < nt; j++) {
        uint32_t base; bool neg;
        row_term(row, j, NB <= 8 ? 4u : 2u, base, neg);
        if (ww_base<NB>(s, base, bit) == neg) return GE_ERR_ARG;
    }
    if (s.acted & bit) return GE_ERR_ARG;
    if (choice < 1 || choice > n || !(s.alive & (1u << (choice - 1u)))) return GE_ERR_ARG;   // targets must be alive
    const uint32_t sh = 4u * (player - 1u);
    const nib_t clr = ~(nib_t(15) << sh), put = nib_t(choice) << sh;
    s.acted |= bit;
    s.choice = (s.choice & clr) | put;
    const uint32_t act = (row.r0 >> 2) & 7u;
    if (act == ACT_DETECTIVE) {
        const uint32_t tb = 1u << (choice - 1u);
        s.det_v &= ~tb; s.det_w &= ~tb;
        if (s.team_w & tb) s.det_w |= tb; else s.det_v |= tb;
    }
    if (act >= ACT_WOLF_TARGET && act <= ACT_DETECTIVE) { s.sub |= bit; s.sel = (s.sel & clr) | put; }
    return GE_OK;
}

template <int NB> __device__ __forceinline__ int inject_tt(TT<NB> &s, const DevRow &row, const DevCond &cond, uint32_t n, uint32_t player, uint32_t choice) {
    if (player < 1 || player > n) return GE_ERR_ARG;
    if ((row.r0 & 3u) != COMP_ACTION) return GE_ERR_ARG;
    const uint32_t bit = 1u << (player - 1u);
    if (row.r0 & ROW_GENERIC) {
        const uint32_t i = player - 1u;
        if (!clauses_hold(cond, [&](uint32_t b) { return tt_base<NB>(s, b, bit); },
                          [&](uint32_t f) -> uint32_t {
                              if (f == 1u) return (s.lie >> (2u * i)) & 3u;
                              if (f == 2u) return (s.vote >> (2u * i)) & 3u;
                              if (f == 4u) return (uint32_t)(s.rounds >> (4u * i)) & 15u;
                              uint32_t sc = s.score[0];                            // no dynamic index: the record stays in registers
                              if (NB > 4) sc = sel32(i >= 4u, s.score[NB > 4 ? 1 : 0], sc);
                              if (NB > 8) sc = sel32(i >= 8u, s.score[NB > 8 ? 2 : 0], sc);
                              return (sc >> (8u * (i % 4u))) & 255u;
                          })) return GE_ERR_ARG;
    }
    const uint32_t nt = (row.r0 & ROW_GENERIC) ? 0u : (row.r0 >> 8) & 7u;
    for (uint32_t j = 0; j < nt; j++) {
        uint32_t base; bool neg;
        row_term(row, j, 2u, base, neg);
        if (tt_base<NB>(s, base, bit) == neg) return GE_ERR_ARG;
    }
    if (s.acted & bit) return GE_ERR_ARG;
    const uint32_t act = (row.r0 >> 2) & 7u;
    if (act == ACT_TT_STATEMENTS ? choice != 1u : (choice < 1u || choice > 3u)) return GE_ERR_ARG;
    const uint32_t sh = 2u * (player - 1u), clr = ~(3u << sh), put = choice << sh;
    s.acted |= bit;
    s.choice = (s.choice & clr) | put;
    if (act == ACT_TT_STATEMENTS) s.submitted |= bit;
    else if (act == ACT_TT_LIE) s.lie = (s.lie & clr) | put;
    else if (act == ACT_TT_VOTE) { s.vote = (s.vote & clr) | put; s.has_voted |= bit; }
    return GE_OK;
}

template <int NB> __device__ __forceinline__ void inject_group_ww(const SegDev &sg, const DevTable *tables, uint64_t room, const InjectArgs &a, uint32_t lo, uint32_t hi) {
    using L = WWLayout<NB>;
    uint32_t w[L::WORDS];
    load_words<L::WORDS>(sg.base, sg.rooms_padded, room, w);
    WW<NB> s;
    L::unpack(w, s);
    const DevRow &row = tables[sg.table_idx].rows[s.phase];
    const DevCond &cond = tables[sg.table_idx].conds[s.phase];   // read in place: a private copy of the clause table is indexed dynamically and would live in scratch
    for (uint32_t k = lo; k < hi; k++) a.status[k] = inject_ww<NB>(s, row, cond, sg.n_players, a.players[k], a.choices[k]);
    L::pack(s, w);
    store_words<L::WORDS>(sg.base, sg.rooms_padded, room, w);
}
template <int NB> __device__ __forceinline__ void inject_group_tt(const SegDev &sg, const DevTable *tables, uint64_t room, const InjectArgs &a, uint32_t lo, uint32_t hi) {
    using L = TTLayout<NB>;
    uint32_t w[L::WORDS];
    load_words<L::WORDS>(sg.base, sg.rooms_padded, room, w);
    TT<NB> s;
    L::unpack(w, s);
    const DevRow &row = tables[sg.table_idx].rows[s.phase];
    const DevCond &cond = tables[sg.table_idx].conds[s.phase];   // read in place: a private copy of the clause table is indexed dynamically and would live in scratch
    for (uint32_t k = lo; k < hi; k++) a.status[k] = inject_tt<NB>(s, row, cond, sg.n_players, a.players[k], a.choices[k]);
    L::pack(s, w);
    store_words<L::WORDS>(sg.base, sg.rooms_padded, room, w);
}

__global__ void __launch_bounds__(64) ge_inject_kernel(const InjectArgs a, const SegDev *__restrict__ segs, const DevTable *__restrict__ tables) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_groups) return;
    const uint32_t lo = a.group[g], hi = a.group[g + 1];
    const uint64_t room = a.rooms[lo];
    const SegDev *sg = nullptr;
    for (uint32_t k = 0; k < a.n_seg; k++)
        if (room >= segs[k].local_first && room < segs[k].local_first + segs[k].rooms) sg = segs + k;
    if (!sg) {
        for (uint32_t k = lo; k < hi; k++) a.status[k] = GE_ERR_RANGE;
        return;
    }
    const uint64_t r = room - sg->local_first;
    switch (sg->kind) {
    case K_WW8: inject_group_ww<8>(*sg, tables, r, a, lo, hi); break;
    case K_WW12: inject_group_ww<12>(*sg, tables, r, a, lo, hi); break;
    case K_TT4: inject_group_tt<4>(*sg, tables, r, a, lo, hi); break;
    case K_TT8: inject_group_tt<8>(*sg, tables, r, a, lo, hi); break;
    default: inject_group_tt<12>(*sg, tables, r, a, lo, hi); break;
    }
}

// last node of a captured sequence of step launches: the device-side turn base moves on, so the same
// graph can be replayed for the next n turns
__global__ void ge_turn_bump(uint32_t *turn, uint32_t n) { *turn += n; }
