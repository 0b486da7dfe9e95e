// ge_device.h — one turn of one room, in registers, for gfx950.
//
// What a turn is: one graph run of the reference,
//   BotBehaviorNode (agent/game_agent_v2.py:468) -> PhaseNode (:987) -> RefereeNode (:619),
// with the LLM decisions fixed by POLICY.md.  Lane = room: a 64-wide wavefront advances 64
// independent rooms; the per-room rules (conditions, tallies, eliminations, scoring) are bit-parallel
// over N-bit masks (ge_layout.h).  The one badly balanced part, the bots' actions, is spread over
// the wavefront through a work queue in LDS (WaveLds below).
// Integer / branchy code: no MFMA (there is no contraction here), VALU + LDS (the phase table,
// 32 B per row, the queue, a 2 KB n-th-set-bit table).
#pragma once
#include <hip/hip_runtime.h>
#include "ge_layout.h"

namespace ge {

enum { COMP_UI = 0, COMP_TIMER = 1, COMP_ACTION = 2 };
enum { ACT_NONE = 0, ACT_WOLF_TARGET, ACT_DOCTOR_PROTECT, ACT_DETECTIVE, ACT_DAY_VOTE,
       ACT_TT_STATEMENTS, ACT_TT_LIE, ACT_TT_VOTE };
enum { EFF_NONE = 0, EFF_ASSIGN_ROLES, EFF_NIGHT_BEGIN, EFF_NIGHT_RESOLVE, EFF_DAY_RESOLVE,
       EFF_TT_ROUND_START, EFF_TT_REVEAL, EFF_TT_SCORE };
enum { RES_ALWAYS = 0, RES_WOLVES_ZERO, RES_WOLVES_GE_VILLAGERS, RES_FOLLOWS_DAY, RES_FOLLOWS_NIGHT,
       RES_ALL_ROUNDS_DONE, RES_OTHERWISE };

constexpr uint32_t GOLDEN = 0x9E3779B9u;

// Compile-time switches.  Every one selects between shipped builds or is a tuning constant; tools/ab_switches.sh builds
// the non-default value of each and runs the parity subset on it.  (What used to be A/B switches for measured-and-rejected
// variants is gone from the source: the numbers live in profiles/r02_ab_*.txt and in git history.)
// GE_STAMPS=1: diagnostic build (tools/stamps.py) - s_memtime stamps at points of the werewolf turn where no LDS operation
//   is outstanding anyway, accumulated per wavefront; never in the product build.  GE_STAMPS=2: only the two clocks at a
//   wavefront's start and end - s_memtime (shader cycles) against s_memrealtime (constant 100 MHz): the shader clock the
//   turn loop really ran at, un-profiled (tools/clock_probe.py)
// GE_DEAL_EARLY=1: the large-batch fused builds prepare the next deal (every GE_DEAL_PERIOD-th turn) at the head of the turn, not inside
//   the action queue's second LDS shadow: there the compiler duplicated the ~130-instruction deal into both continuations of the queue
//   and carried the prepared deal's words through two register sets, copied over on every turn.  Werewolf x 8 large-batch build 803 ->
//   676 vector instructions, 1 M rooms -0.8 %, Werewolf x 12 -0.5 % (profiles/r05_ab_valu_price.txt; the lone-wavefront build keeps the
//   shadow: a lone wavefront has nothing else to run during an LDS round trip)
// GE_SINGLE_GLOBAL: which single-turn builds read the table image where it lies in global memory (through the vector cache) instead of
//   filling the block's LDS with it behind a barrier first - bit 0 Werewolf lone-wavefront, 1 Werewolf large-batch, 2 Two-Truths
//   lone-wavefront, 3 Two-Truths large-batch.  A lone wavefront's single turn is a chain of latencies, and fill -> barrier -> LDS read is a
//   longer one than a cached load: sustained us per single-turn launch at 65 536 rooms (profiles/r05_ab_single_global.txt) Werewolf x 8
//   3.73 -> 3.38 (-9.5 %), x 12 4.46 -> 4.29, Two-Truths x 4 3.12 -> 2.85, x 12 4.22 -> 3.97, a mixed 16 384 + 16 384 batch 3.59 -> 3.35.
//   The large-batch builds keep the LDS tables: a Werewolf turn looks up a table per queue slot and 64 scattered addresses cost the
//   vector cache far more than LDS (1 M x 8 13.5 -> 15.7 us, 2 M x 12 31.9 -> 41, 1 GiB of records 376 -> 486); Two-Truths x 4 gains 2 % at
//   1 M rooms but loses 13 % once the records stream from HBM (33 M rooms 273 -> 309 us), x 8 loses 5 %.
#ifndef GE_SINGLE_GLOBAL
#define GE_SINGLE_GLOBAL 5
#endif
template <bool LOWOCC, bool SINGLE> constexpr bool single_global(int game_bit) { return SINGLE && ((GE_SINGLE_GLOBAL >> (game_bit + (LOWOCC ? 0 : 1))) & 1) != 0; }
#ifndef GE_DEAL_EARLY
#define GE_DEAL_EARLY 1
#endif
#ifndef GE_STAMPS
#define GE_STAMPS 0
#endif
// LDS bank layout A/B (round 5, profiles/r05_ab_lds_banks.txt): GE_RES_PACKED - the queue's result words of a Werewolf x 8 room 8
//   bytes apart instead of 16; GE_ROWS_SPLIT - the block's phase rows as two arrays of 16-byte halves (lds_row)
#ifndef GE_RES_PACKED
#define GE_RES_PACKED 1
#endif
// Issue priority (profiles/r05_ab_queue_prio.txt).  The large-batch fused builds are bound by latency on two half-busy shared units (vector
//   pipe ~60 %, LDS array ~60 %: DESIGN.md 4 "Attribution"), so which wavefront issues next matters: a wavefront inside the queue's dependent
//   LDS round trips (GE_QUEUE_PRIO, s_setprio from the context write to the result read) or inside a vote resolution's table lookups
//   (GE_RESOLVE_PRIO) issues ahead of wavefronts doing independent vector work.  GE_RES_ATOMIC64: a queue slot of a Werewolf x 8 room returns its
//   result with one 64-bit LDS atomic on the packed pair instead of two 32-bit ones.  Together -2.6 % at 1 M Werewolf x 8, -3.3 % Two-Truths x 4.
#ifndef GE_QUEUE_PRIO
#define GE_QUEUE_PRIO 3
#endif
#ifndef GE_RES_ATOMIC64
#define GE_RES_ATOMIC64 1
#endif
#ifndef GE_RESOLVE_PRIO
#define GE_RESOLVE_PRIO 2
#endif
#ifndef GE_ROWS_SPLIT
#define GE_ROWS_SPLIT 1
#endif
// GE_DEAL_PERIOD: role deals are prepared ahead every GE_DEAL_PERIOD-th turn (a power of two; a game is longer, and a room
//   whose deal is not ready when it needs one deals on the spot).  8 / 16 / 32 measured: profiles/r02_ab_deal_shadow.txt
#ifndef GE_DEAL_PERIOD
#define GE_DEAL_PERIOD 16
#endif

// What distinguishes the two shipped builds of the werewolf turn (chosen per launch from the batch size, ge_step.hip
// fill_args): LOWOCC = at most one wavefront per SIMD (C2).  A lone wavefront pays an issue slot of >= 4 cycles for
// every instruction whatever its type, a bubble for every branch and the full latency of every dependent LDS round trip,
// so that build is branch-lean and computes; with many wavefronts per SIMD the kernel is VALU-bound and prefers LDS
// tables and skip-branches.  Each entry was an A/B on MI355X (profiles/r02_ab_*.txt):
// SINGLE = the launch advances every room by exactly one turn (max_fuse = 1: interactive / traced stepping, and the
// launch that really streams the state through HBM every turn): no turn loop, nothing prepared for a next turn.
template <int NB, bool LOWOCC, bool SINGLE = false> struct WwBuild {
    static constexpr bool ORD = NB <= 8;                    // queue slots find their player through the ord8 table (r02_ab_shadow_ord)
    static constexpr bool ONE_ATOMIC = NB <= 8 && LOWOCC;   // one result atomic per slot; the room derives who acted from the non-zero choice nibbles (r02_ab_choose_one_atomic)
    static constexpr bool SHADOW = LOWOCC || NB <= 8;       // action-independent work inside the queue's two LDS round trips (r02_ab_occupancy: N > 8 large-batch is better off without)
    static constexpr bool ONEHOT = LOWOCC && NB <= 8;       // queue slots carry the action kind one-hot: selects by v_bfe_i32 masks + v_bfi, no compares (r02_ab_onehot_swar_pk)
    static constexpr bool PIN_CHOICE = LOWOCC && NB > 8;    // the slot's choice computed outside its `acts this turn` region (r02_ab_sel_pin)
    static constexpr bool SEL_OPEN = LOWOCC;                // completion test without short-circuit evaluation (r02_ab_open_tpldeal)
    static constexpr bool TPL_TRACE = LOWOCC;               // the turn loop compiled once per trace setting (r02_ab_branch_diet)
    static constexpr int DEAL_FORM = SINGLE ? 2 : LOWOCC ? 0 : NB <= 8 ? 1 : 2;   // DEAL_PACKED / DEAL_MASKS / DEAL_WORDS, see struct Deal
    static constexpr bool DEAL_EARLY = GE_DEAL_EARLY && !LOWOCC && !SINGLE;   // the deal is prepared at the head of the turn, not in the queue's second LDS shadow (see ww_turn)
    static constexpr bool TABLE = !LOWOCC;                  // n-th-set-bit through the 2 KB LDS table instead of ~15 VALU instructions
};

// ---- POLICY.md §RNG: stateless 32-bit counter hash
GE_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
GE_HD uint32_t room_key(uint32_t seed_lo, uint32_t seed_hi, uint64_t room) {
    uint32_t k = mix32(seed_lo ^ 0x243F6A88u);
    k = mix32(k ^ seed_hi);
    k = mix32(k ^ (uint32_t)room);
    k = mix32(k ^ (uint32_t)(room >> 32));
    return k;
}
// the two seed rounds of room_key are the same for every room: done once on the host
GE_HD uint32_t seed_key(uint32_t seed_lo, uint32_t seed_hi) { return mix32(mix32(seed_lo ^ 0x243F6A88u) ^ seed_hi); }
GE_HD uint32_t room_key_from(uint32_t sk, uint64_t room) { return mix32(mix32(sk ^ (uint32_t)room) ^ (uint32_t)(room >> 32)); }
GE_HD uint32_t turn_key(uint32_t rk, uint32_t turn) { return mix32(rk ^ (turn * GOLDEN)); }
GE_HD uint32_t draw(uint32_t tk, uint32_t idx) { return mix32(tk + (idx + 1u) * GOLDEN); }
// role picks are keyed by the room's game index, not by the turn in which they are applied
GE_HD uint32_t deal_key(uint32_t rk, uint32_t game) { return mix32(rk ^ 0x44454C31u ^ (game * GOLDEN)); }

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t pick(uint32_t d, uint32_t k) { return __umulhi(d, k); }
__device__ __forceinline__ uint32_t popc(uint32_t x) { return (uint32_t)__popc(x); }
__device__ __forceinline__ uint32_t ctz(uint32_t x) { return (uint32_t)__ffs((int)x) - 1u; }

// position of the n-th (0-based) set bit of a <=16-bit mask; the bit must exist
template <int NB> __device__ __forceinline__ uint32_t nth_set_bit(uint32_t m, uint32_t n) {
    uint32_t pos = 0, c;
    if (NB > 8) { c = popc(m & 0xFFu); if (n >= c) { n -= c; pos = 8; m >>= 8; } }
    c = popc(m & 0xFu); if (n >= c) { n -= c; pos += 4; m >>= 4; }
    c = popc(m & 0x3u); if (n >= c) { n -= c; pos += 2; m >>= 2; }
    c = m & 1u;         if (n >= c) { pos += 1; }
    return pos;
}

// NB <= 8, the same from nibble prefix counts: spread the mask to one bit per nibble; (x + 7 - n) * 0x11111111 puts
// (number of set bits at positions <= j) + 7 - n into nibble j (<= 15: no carry), whose bit 3 says "more than n set bits
// up to here"; the lowest such nibble is the n-th set bit.  12 instructions (one quarter-rate) instead of 22.
// (the bit may not exist - a stale queue slot, a deal pick that is not taken - and the caller drops the result then: the
// sentinel keeps the count-trailing-zeros defined; it folds into the AND as one v_and_or_b32)
__device__ __forceinline__ uint32_t nth_set_bit_swar8(uint32_t m, uint32_t n) {
    uint32_t x = m & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu; x = (x | (x << 6)) & 0x03030303u; x = (x | (x << 3)) & 0x11111111u;
    const uint32_t t = (x + 7u - n) * 0x11111111u;
    return (uint32_t)__builtin_ctz((t & 0x88888888u) | 0x80000000u) >> 2;
}

// the same through a 2 KB LDS table nth8[mask][n] (mask: 8 bits): one LDS read instead of ~15 VALU
// instructions.  Pays at many wavefronts per SIMD (VALU-bound); a lone wavefront would only add
// LDS latency to a dependent chain, so it keeps the computed form (`lowocc`).
template <int NB> __device__ __forceinline__ uint32_t nth_set_bit_lds(const uint8_t *nth8, uint32_t m, uint32_t n) {
    if (NB <= 8) return nth8[(m & 0xFFu) * 8u + (n & 7u)];
    const uint32_t lo = m & 0xFFu, c = popc(lo);
    const bool in_lo = n < c;
    const uint32_t idx = in_lo ? lo * 8u + n : ((m >> 8) & 0xFFu) * 8u + ((n - c) & 7u);
    return (in_lo ? 0u : 8u) + nth8[idx];
}

// host side: the table's content (nth8[mask][n] = position of the n-th set bit of the 8-bit mask)
inline void fill_nth8_host(uint8_t *nth8) {
    for (uint32_t m = 0; m < 256u; m++) {
        uint32_t x = m;
        for (uint32_t n = 0; n < 8u; n++) {
            uint32_t pos = 0;
            while (x && !((x >> pos) & 1u)) pos++;
            nth8[m * 8u + n] = (uint8_t)(x ? pos : 0u);
            x &= x - 1u;
        }
    }
}

// spread8 / tally64 (DevTable): see plurality()
inline void fill_vote_luts_host(uint32_t *spread8, uint64_t *tally64) {
    for (uint32_t m = 0; m < 256u; m++) {
        uint32_t v = 0;
        for (uint32_t i = 0; i < 8u; i++)
            if ((m >> i) & 1u) v |= 0xFu << (4u * i);
        spread8[m] = v;
        tally64[m] = (uint64_t(1) << (4u * (m & 15u))) + (uint64_t(1) << (4u * (m >> 4)));
    }
}

// ord8[mask]: nibble r = position of the r-th set bit of the 8-bit mask (0 past the last one)
inline void fill_ord8_host(uint32_t *ord8) {
    for (uint32_t m = 0; m < 256u; m++) {
        uint32_t v = 0, r = 0;
        for (uint32_t i = 0; i < 8u; i++)
            if ((m >> i) & 1u) v |= i << (4u * r++);
        ord8[m] = v;
    }
}

// x has at most bit 0 of each nibble set: widen every such bit to a full 0xF nibble.  ORs of shifts,
// not (x << 4) - x: the compiler turns that into v_mul_lo_u32 by 15, a quarter-rate instruction
// (the empty asm hides the intermediate from the optimiser, which would otherwise prove the bits disjoint
// and fold the ORs back into that multiply)
__device__ __forceinline__ uint32_t nib_fill(uint32_t x) { x |= x << 1; asm("" : "+v"(x)); return x | (x << 2); }
__device__ __forceinline__ uint64_t nib_fill(uint64_t x) { x |= x << 1; asm("" : "+v"(x)); return x | (x << 2); }

// 1-based id with the most votes among `voters`, ties -> lowest id, 0 if nobody voted.
// votes: one nibble per player (1-based target id, 0 = none).  Counters are nibbles too
// (<= 12 voters), so the whole tally is one or two registers.
// LUT (large-batch builds): `img` = the block's LDS table image; voters are spread through spread8 and the counters come
// two votes at a time from tally64 - 6 lookups instead of 12 shift-and-add steps for 12 players (the resolution is ~30 % of
// a Werewolf x 12 turn's vector instructions: some room of a wavefront resolves on every turn).  A lone wavefront would
// only add LDS round trips to its dependency chain and keeps the computed form.
template <int NB, typename nib_t, bool LUT = false>
__device__ __forceinline__ uint32_t plurality(nib_t votes, uint32_t voters, const void *img = nullptr) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pk = [](uint32_t x) { return __builtin_bit_cast(u16x2, x); };
    if (LUT) {
        const uint32_t *spread8 = reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(img) + IMG_SPREAD8);
        const uint2 *tally64 = reinterpret_cast<const uint2 *>(static_cast<const unsigned char *>(img) + IMG_TALLY);
        uint32_t v_lo = (uint32_t)votes & spread8[voters & 0xFFu], v_hi = 0;
        if (NB > 8) v_hi = (uint32_t)((uint64_t)votes >> 32) & spread8[(voters >> 8) & 0xFFu];
        // the two halves of the counters never carry into each other (<= 12 votes per nibble): plain 32-bit adds
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint2 t = tally64[(v_lo >> (8 * k)) & 0xFFu]; lo += t.x; hi += t.y; }
        if (NB > 8) {
#pragma unroll
            for (int k = 0; k < 2; k++) { const uint2 t = tally64[(v_hi >> (8 * k)) & 0xFFu]; lo += t.x; hi += t.y; }
        }
        // keys count << 4 | 15 - id, two per 32-bit register (see below): ids 1..7 from the low word's nibbles, 8..12 from the high word's
        const uint32_t a = (lo & 0x00F000F0u) | 0x000A000Eu;                                    // ids 1, 5
        const uint32_t b = ((lo >> 4) & 0x00F000F0u) | 0x0009000Du;                             // ids 2, 6
        const uint32_t c = ((lo >> 8) & 0x00F000F0u) | 0x0008000Cu;                             // ids 3, 7
        const uint32_t d = ((lo >> 12) & 0x000000F0u) | ((hi & 15u) << 20) | 0x0007000Bu;       // ids 4, 8
        u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(pk(a), pk(b)), __builtin_elementwise_max(pk(c), pk(d)));
        if (NB > 8) {
            const uint32_t e = (hi & 0x000000F0u) | ((hi << 12) & 0x00F00000u) | 0x00050006u;       // ids 9, 10
            const uint32_t f = ((hi >> 8) & 0x000000F0u) | ((hi << 4) & 0x00F00000u) | 0x00030004u; // ids 11, 12
            m = __builtin_elementwise_max(m, __builtin_elementwise_max(pk(e), pk(f)));
        }
        const uint32_t key = m.x > m.y ? m.x : m.y;
        return (key >> 4) ? 15u - (key & 15u) : 0u;
    }
    // keep only the voters' nibbles: spread the voter bits to nibble position 0, times 15
    nib_t vm;
    if (NB <= 8) {
        uint32_t x = voters & 0xFFu;
        x = (x | (x << 12)) & 0x000F000Fu; x = (x | (x << 6)) & 0x03030303u; x = (x | (x << 3)) & 0x11111111u;
        vm = (nib_t)nib_fill(x);
    } else {
        uint64_t x = voters & 0xFFFFu;
        x = (x | (x << 24)) & 0x000000FF000000FFull; x = (x | (x << 12)) & 0x000F000F000F000Full;
        x = (x | (x << 6)) & 0x0303030303030303ull; x = (x | (x << 3)) & 0x1111111111111111ull;
        vm = (nib_t)nib_fill(x);
    }
    const nib_t v = votes & vm;
    // tally: counter k (a nibble) counts votes for player id k; nibble 0 collects "no vote"
    uint32_t key = 0;                                          // max over k of (count << 4 | 15 - k): ties -> lowest id
    if (NB <= 8) {
        // 32-bit counters: a vote for player 8 would be nibble 8; its shift (32) wraps to nibble 0,
        // which nobody reads, and player 8 is counted from bit 3 of the vote nibbles instead
        const uint32_t v32 = (uint32_t)v;
        uint32_t tally = 0;
#pragma unroll
        for (int i = 0; i < NB; i++) tally += 1u << ((4u * ((v32 >> (4 * i)) & 15u)) & 31u);
        {
            // two keys per register, 16 bits each: counters 1 / 5 sit at bits 4..7 of the two halves already (count << 4),
            // 2 / 6, 3 / 7 and 4 after a shift; player 8's count comes from bit 3 of the vote nibbles
            const uint32_t c8 = popc(v32 & 0x88888888u);
            const uint32_t a = (tally & 0x00F000F0u) | 0x000A000Eu;                 // ids 1, 5: 15 - id = 14, 10
            const uint32_t b = ((tally >> 4) & 0x00F000F0u) | 0x0009000Du;          // ids 2, 6
            const uint32_t c = ((tally >> 8) & 0x00F000F0u) | 0x0008000Cu;          // ids 3, 7
            const uint32_t d = ((tally >> 12) & 0x000000F0u) | (c8 << 20) | 0x0007000Bu;   // ids 4, 8
            const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(pk(a), pk(b)), __builtin_elementwise_max(pk(c), pk(d)));
            key = m.x > m.y ? m.x : m.y;
        }
    } else {
        uint64_t tally = 0;
#pragma unroll
        for (int i = 0; i < NB; i++) tally += uint64_t(1) << (4u * ((uint32_t)(v >> (4 * i)) & 15u));
        // keys count << 4 | 15 - id, two per register as above: ids 1..7 from the low word's nibbles, 8..12 from the high word's
        const uint32_t lo = (uint32_t)tally, hi = (uint32_t)(tally >> 32);
        const uint32_t a = (lo & 0x00F000F0u) | 0x000A000Eu;                                    // ids 1, 5
        const uint32_t b = ((lo >> 4) & 0x00F000F0u) | 0x0009000Du;                             // ids 2, 6
        const uint32_t c = ((lo >> 8) & 0x00F000F0u) | 0x0008000Cu;                             // ids 3, 7
        const uint32_t d = ((lo >> 12) & 0x000000F0u) | ((hi & 15u) << 20) | 0x0007000Bu;       // ids 4, 8
        const uint32_t e = (hi & 0x000000F0u) | ((hi << 12) & 0x00F00000u) | 0x00050006u;       // ids 9, 10
        const uint32_t f = ((hi >> 8) & 0x000000F0u) | ((hi << 4) & 0x00F00000u) | 0x00030004u; // ids 11, 12
        const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(pk(a), pk(b)), __builtin_elementwise_max(pk(c), pk(d))),
                                                  __builtin_elementwise_max(pk(e), pk(f)));
        key = m.x > m.y ? m.x : m.y;
    }
    return (key >> 4) ? 15u - (key & 15u) : 0u;
}

// Per-wavefront LDS scratch of the bot-action work queue.  Lane = room leaves the action step
// badly balanced: the mean number of due (room, player) actions is ~1 per room and turn, but a
// room on the first turn of a vote has 8-12, and a per-lane loop runs max-over-lanes iterations.
// So the wavefront compacts all due actions of its 64 rooms into one queue in LDS, every lane
// takes one item per round (reading the owning room's context from LDS), and results go back
// with LDS atomic ORs.  Wavefront-private: no block barrier, only wave-level ordering.
struct WaveLds {
    uint4 ctx[64];            // {alive | team_w<<16 | act<<28, known-or-r_det | lo_kw<<16, due mask | first slot<<16 | lane<<26, turn key}
    uint4 res[64];            // {go mask, choice nibbles lo, choice nibbles hi, -}
    uint8_t queue[64 * 13];   // slot -> lane of the owning room; 64 * 12 slots at most, read in rounds of 64
};
// The lone-wavefront build pays ~150 cycles for every dependent LDS round trip and has LDS to spare, so
// there a slot holds the owning room's whole context (one read per item instead of lane id -> context).
struct WaveLdsLow {
    uint4 slot[64 * 13 + 80];  // + scratch for rooms without a due bot (lane .. lane + 12)
    uint4 res[64];
};
template <bool LOWOCC> struct WaveLdsOf { using type = WaveLds; };
template <> struct WaveLdsOf<true> { using type = WaveLdsLow; };

// A phase row from the block's LDS table image.  The image keeps the rows' two 16-byte halves in two arrays (half h of row r at
// element h * 32 + r; load_rows permutes while it copies): a wavefront's lanes read 18 or so different rows at once, and 32 bytes
// apart rows r, r + 8, r + 16 share their banks (ds_read_b128: bank = dword address mod 64) - every row had one or two partners in
// the shipped tables; 16 bytes apart only r and r + 16 do.
// SPLIT = the large-batch fused builds (A/B, profiles/r05_ab_lds_banks.txt: with the packed results -2.2 % at 1 M Werewolf x 8
// rooms, -3 % on the C5 mix; a lone wavefront has no bank conflicts with itself to lose and pays 1 % for the second address, a
// single-turn launch reads one row per room and gains nothing).
template <bool SPLIT>
__device__ __forceinline__ DevRow lds_row(const DevRow *rows, uint32_t idx) {
    if (!(SPLIT && GE_ROWS_SPLIT)) return rows[idx];
    const uint4 *h = reinterpret_cast<const uint4 *>(rows);
    const uint4 a = h[idx], b = h[GE_MAX_PHASES + idx];
    return DevRow{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum of a small per-lane count over the wavefront, and the wave total (uniform)
__device__ __forceinline__ void wave_excl_scan(uint32_t cnt, uint32_t &off, uint32_t &total) {
    // inclusive scan by DPP: inside each row of 16 lanes (row_shr 1, 2, 4, 8), then
    // row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows 0-1 -> 2-3 (row_bcast:31)
    uint32_t v = cnt;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    off = v - cnt;
    total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct Stamps {
    unsigned long long last, acc[4];
    __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); acc[0] = acc[1] = acc[2] = acc[3] = 0; }
    __device__ __forceinline__ void mark(int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[k] += now - last; last = now; }
    unsigned long long m0, r0;      // GE_STAMPS = 2
    __device__ __forceinline__ void clocks_start() { m0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
};

// candidate choice of one bot action (POLICY.md §3); shared by the per-lane loop and the queue.
// Written as mask arithmetic (sel = b ^ ((a ^ b) & -cond)), not ?: chains: the compiler turned some of those into
// exec-mask regions, each a pair of scalar instructions and a branch that a lone wavefront pays in full.
__device__ __forceinline__ uint32_t sel32(bool c, uint32_t a, uint32_t b) { return b ^ ((a ^ b) & (0u - (uint32_t)c)); }

template <int NB, bool TABLE>
__device__ __forceinline__ uint32_t ww_choose(uint32_t act, uint32_t i, uint32_t d, uint32_t alive, uint32_t team_w,
                                              uint32_t known, uint32_t lo_kw, uint32_t r_det, const uint8_t *nth8) {
    const uint32_t me = 1u << i;
    const uint32_t others = alive & ~me, non_wolf = alive & ~team_w;
    const uint32_t fresh = others & ~known;
    uint32_t cand;
    if (!TABLE) {
        // lone-wavefront build: mask arithmetic (profiles/r02_ab_choose_one_atomic.txt: C2 1.383 -> 1.331 us/turn; the
        // large-batch build loses 2 % with it and keeps the ?: form)
        const uint32_t det_c = sel32(fresh != 0u, fresh, others);                       // ACT_DETECTIVE
        const uint32_t vote = sel32((team_w & me) != 0u, non_wolf, sel32((r_det & me) != 0u && lo_kw != 0u, lo_kw, others));   // ACT_DAY_VOTE
        cand = alive;                                                                   // ACT_DOCTOR_PROTECT
        cand = sel32(act == ACT_WOLF_TARGET, non_wolf, cand);
        cand = sel32(act == ACT_DETECTIVE, det_c, cand);
        cand = sel32(act == ACT_DAY_VOTE, vote, cand);
        cand = sel32(cand != 0u, cand, alive);
    } else {
        cand = alive;                                                                   // ACT_DOCTOR_PROTECT
        cand = act == ACT_WOLF_TARGET ? non_wolf : cand;
        cand = act == ACT_DETECTIVE ? (fresh ? fresh : others) : cand;
        const uint32_t vote = (team_w & me) ? non_wolf : (((r_det & me) && lo_kw) ? lo_kw : others);
        cand = act == ACT_DAY_VOTE ? vote : cand;
        cand = cand ? cand : alive;
    }
    const uint32_t idx = pick(d, popc(cand));
    return (TABLE ? nth_set_bit_lds<NB>(nth8, cand, idx) : NB <= 8 ? nth_set_bit_swar8(cand, idx) : nth_set_bit<NB>(cand, idx)) + 1u;
}

// 0 / ~0 from bit `pos` of x (v_bfe_i32), and a select by such a mask (v_bfi_b32): no compare, no VCC
// (the empty asm keeps the optimiser from recognising the mask as a sign-extended compare and turning the select back
// into v_cmp + v_cndmask)
__device__ __forceinline__ uint32_t bit_mask(uint32_t x, uint32_t pos) {
    uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)x, pos, 1u);
    asm("" : "+v"(m));
    return m;
}
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }

// ww_choose for a queue slot of the lone-wavefront build, N <= 8 (WwBuild::ONEHOT).  x = alive | team_w << 16 | kind << 28 with
// the action kind ONE-HOT (bit 28 wolf target, 29 doctor, 30 detective, 31 day vote); `det_voter`: by day, the Detective's
// bit if it knows a living werewolf (else 0) - the owning room resolves that, so the slot needs no compare for it.
__device__ __forceinline__ uint32_t ww_choose_onehot8(uint32_t x, uint32_t i, uint32_t d, uint32_t known, uint32_t lo_kw, uint32_t det_voter) {
    const uint32_t alive = x & 0xFFFFu, team_w = (x >> 16) & 0xFFFu;
    const uint32_t me = 1u << i;
    const uint32_t others = alive & ~me, non_wolf = alive & ~team_w;
    const uint32_t fresh = others & ~known;
    const uint32_t det_c = sel32(fresh != 0u, fresh, others);                                   // ACT_DETECTIVE
    const uint32_t vote = bfi(bit_mask(team_w, i), non_wolf, bfi(bit_mask(det_voter, i), lo_kw, others));   // ACT_DAY_VOTE
    uint32_t cand = alive;                                                                      // ACT_DOCTOR_PROTECT
    cand = bfi(bit_mask(x, 28), non_wolf, cand);
    cand = bfi(bit_mask(x, 30), det_c, cand);
    cand = bfi(bit_mask(x, 31), vote, cand);
    cand = sel32(cand != 0u, cand, alive);
    const uint32_t idx = pick(d, popc(cand));
    return nth_set_bit_swar8(cand, idx) + 1u;
}

// nibble mask (0xF per player) of the non-zero nibbles of x
__device__ __forceinline__ uint32_t nib_nonzero(uint32_t x) {
    uint32_t m = x | (x >> 1); m |= m >> 2; m &= 0x11111111u; return nib_fill(m);
}
__device__ __forceinline__ uint64_t nib_nonzero(uint64_t x) {
    uint64_t m = x | (x >> 1); m |= m >> 2; m &= 0x1111111111111111ull; return nib_fill(m);
}

// A room's role deal (POLICY.md §3 ASSIGN_ROLES): nw werewolves, then a Doctor, then a Detective, by repeated
// n-th-set-bit sampling of the players not yet dealt; the rest are Villagers.  Picks are keyed by (room, game index), not by
// the turn that applies them, so a deal can be prepared ahead - in registers during a fused launch, and for N <= 8 across
// launches in the record's spare half-word (ge_layout.h, WWLayout<8>).  Three in-register forms (WwBuild::DEAL_FORM):
//   DEAL_MASKS   a / b / c / rem = werewolves / Doctor / Detective / Villagers as player masks - the lone-wavefront build,
//                where turning them into predicate words sits in an LDS wait shadow and costs nothing;
//   DEAL_WORDS   a / b / c = the three packed predicate words an assignment writes (N <= 8, large-batch fused build:
//                the conversion is paid once per deal, not once per turn; profiles/r02_ab_onehot_swar_pk.txt);
//   DEAL_PACKED  a = DealPk, one register - Werewolf x 12 large-batch (register pressure: that build is held to 80 VGPRs)
//                and every single-turn build; expanded only by the lanes that assign.
// gv = game index | 1 << 31 while the deal is valid, else 0: "ready for this game" is one compare.
enum { DEAL_MASKS = 0, DEAL_WORDS = 1, DEAL_PACKED = 2 };
struct Deal { uint32_t a, b, c, rem, gv; };
constexpr uint32_t DEAL_VALID = 1u << 31;

// werewolves mask | Doctor's index << WB | Detective's index << (WB + IB) | 1 << OK_SH.  16 bits for N <= 8 (the record's cache)
template <int NB> struct DealPk {
    static constexpr uint32_t WB = NB <= 8 ? 8 : 12, IB = NB <= 8 ? 3 : 4;
    static constexpr uint32_t WM = (1u << WB) - 1u, IM = (1u << IB) - 1u, DOC_SH = WB, DET_SH = WB + IB, OK_SH = WB + 2 * IB;
    static __device__ __forceinline__ uint32_t pack(uint32_t wolves, uint32_t doc, uint32_t det) {
        return wolves | (ctz(doc | 0x80000000u) & IM) << DOC_SH | (ctz(det | 0x80000000u) & IM) << DET_SH | 1u << OK_SH;
    }
    static __device__ __forceinline__ void unpack(uint32_t pk, uint32_t all, uint32_t &wolves, uint32_t &doc, uint32_t &det, uint32_t &rem) {
        wolves = pk & WM; doc = 1u << ((pk >> DOC_SH) & IM); det = 1u << ((pk >> DET_SH) & IM);
        rem = all & ~(wolves | doc | det);
    }
};

// the packed predicate words a role assignment writes, from the four masks
template <int NB> __device__ __forceinline__ void role_words(uint32_t wolves, uint32_t doc, uint32_t det, uint32_t rem, uint32_t all, WWR<NB> &w) {
    const uint32_t special = all & ~rem;
#pragma unroll
    for (int k = 0; k < WWR<NB>::NW; k++) w.W[k] = 0;
    w.template set<F_VIL>(rem); w.template set<F_WOLF>(wolves); w.template set<F_DOC>(doc); w.template set<F_DET>(det);
    w.template set<F_TEAM_W>(wolves); w.template set<F_TEAM_V>(all & ~wolves);
    w.template set<F_SECRET>(special); w.template set<F_ELIG>(special);
}

template <int NB, int FORM>
__device__ __forceinline__ void deal_set(Deal &d, uint32_t wolves, uint32_t doc, uint32_t det, uint32_t rem, uint32_t all, uint32_t game) {
    if (FORM == DEAL_WORDS) {
        WWR<NB> w;
        role_words<NB>(wolves, doc, det, rem, all, w);
        d.a = w.W[0]; d.b = w.W[1]; d.c = w.W[2]; d.rem = 0u;
    } else if (FORM == DEAL_PACKED) {
        d.a = DealPk<NB>::pack(wolves, doc, det); d.b = d.c = d.rem = 0u;
    } else {
        d.a = wolves; d.b = doc; d.c = det; d.rem = rem;
    }
    d.gv = game | DEAL_VALID;
}

// TABLE: n-th-set-bit from the LDS table (large-batch builds).
// N > 8 does without any n-th-set-bit search: the idx-th player not yet dealt = idx, moved up past every earlier pick at or
// below it in ascending order; the earlier picks are kept sorted by a min / max chain.  4 (pick number) instructions per
// pick instead of a two-level table lookup (~13) or a binary search (~22): a deal is 33 - 70 vector instructions shorter,
// which is what a Werewolf x 12 single-turn launch pays on most turns (its record has no room for a prepared deal).
template <int NB, bool TABLE, int FORM>
__device__ __forceinline__ void deal_roles(Deal &d, uint32_t dk, uint32_t game, uint32_t n, uint32_t nw, const uint8_t *nth8) {
    uint32_t rem = (1u << n) - 1u, wolves = 0, doc = 0, det = 0;
    uint32_t prev[4] = {255u, 255u, 255u, 255u};           // N > 8: the earlier picks, ascending
#pragma unroll
    for (uint32_t j = 0; j < (NB > 8 ? 5u : 4u); j++) {           // nw + 2 picks, nw <= NB / 4
        // every pick takes exactly one player (n >= 4 >= nw + 2 for every admitted n), so j picks leave n - j: a wave-uniform
        // count instead of a popcount of `rem` per pick
        const uint32_t k = n - j;
        const bool on = j < nw + 2u;                       // selects, not branches (see WwBuild)
        const uint32_t idx = pick(draw(dk, 16u + j), k);
        uint32_t pos;
        if (NB > 8) {
            pos = idx;
#pragma unroll
            for (uint32_t i = 0; i < j && i < 4u; i++) pos += pos >= prev[i] ? 1u : 0u;
            uint32_t x = on ? pos : 255u;                  // insert (a pick that is not taken sorts last and moves nobody)
#pragma unroll
            for (uint32_t i = 0; i < j && i < 4u; i++) { const uint32_t lo = prev[i] < x ? prev[i] : x; x = prev[i] < x ? x : prev[i]; prev[i] = lo; }
            if (j < 4u) prev[j] = x;
        } else {
            pos = TABLE ? nth_set_bit_lds<NB>(nth8, rem, idx) : nth_set_bit_swar8(rem, idx);
        }
        const uint32_t bit = on ? (1u << (pos & 15u)) : 0u;
        rem &= ~bit;
        wolves |= j < nw ? bit : 0u;
        doc = j == nw ? bit : doc;
        det = j == nw + 1u ? bit : det;
    }
    deal_set<NB, FORM>(d, wolves, doc, det, rem, (1u << n) - 1u, game);
}

// the packed predicate words a role assignment writes, from the prepared deal
template <int NB, int FORM> __device__ __forceinline__ void deal_words(const Deal &deal, uint32_t all, WWR<NB> &dealt) {
    if (FORM == DEAL_WORDS) {
        dealt.W[0] = deal.a; dealt.W[1] = deal.b; dealt.W[2] = deal.c;
    } else if (FORM == DEAL_PACKED) {
        uint32_t wolves, doc, det, rem;
        DealPk<NB>::unpack(deal.a, all, wolves, doc, det, rem);
        role_words<NB>(wolves, doc, det, rem, all, dealt);
    } else {
        role_words<NB>(deal.a, deal.b, deal.c, deal.rem, all, dealt);
    }
}

// the game whose deal a room needs next: this game's while it has no roles yet, else the next game's
template <int NB> __device__ __forceinline__ uint32_t deal_next_game(const WWR<NB> &s) {
    const bool has_roles = (NB <= 8 ? s.W[2] : (s.W[WWR<NB>::NW - 2] | s.W[WWR<NB>::NW - 1])) != 0u;
    return has_roles ? (s.games < 0xFFFFu ? s.games + 1u : s.games) : s.games;
}

// record cache <-> registers (N <= 8; Werewolf x 12 records have no spare bits: nothing is kept across launches).  The
// cache is written only when it is the deal of deal_next_game(the stored state), so the loader need not store the game.
template <int NB, int FORM> __device__ __forceinline__ void deal_from_cache(uint32_t cache, const WWR<NB> &s, uint32_t all, Deal &d) {
    d.a = d.b = d.c = d.rem = d.gv = 0u;
    if (NB > 8) return;
    const uint32_t g = deal_next_game<NB>(s);
    const bool ok = (cache >> DealPk<NB>::OK_SH) & 1u;
    if (FORM == DEAL_PACKED) {
        d.a = cache;
    } else {
        uint32_t wolves, doc, det, rem;
        DealPk<NB>::unpack(cache, all, wolves, doc, det, rem);
        deal_set<NB, FORM>(d, wolves, doc, det, rem, all, g);
    }
    d.gv = ok ? (g | DEAL_VALID) : 0u;
}
template <int NB, int FORM> __device__ __forceinline__ uint32_t deal_to_cache(const Deal &d, const WWR<NB> &s) {
    if (NB > 8) return 0u;
    uint32_t pk;
    if (FORM == DEAL_PACKED) pk = d.a;
    else if (FORM == DEAL_WORDS) pk = DealPk<NB>::pack((d.c >> 8) & 0xFFu, (d.c >> 16) & 0xFFu, d.c >> 24);   // W[2] = r_vil | r_wolf | r_doc | r_det
    else pk = DealPk<NB>::pack(d.a, d.b, d.c);
    return d.gv == (deal_next_game<NB>(s) | DEAL_VALID) ? (pk & 0xFFFFu) : 0u;
}

// ------------------------------------------------------------------ generic target conditions
// The slow path of `target_players.condition`: an OR of AND-clauses of literals, for DSLs that use the rest of the grammar
// (or, in [..], numeric comparisons - agent/prompt/dsl_phases_generation_prompt.txt:106-150).  Only the GENERIC kernel
// builds contain it, and only the lanes whose current row is flagged ROW_GENERIC use its result; the shipped games never do.
// The literals come from the block's LDS copy of the table's literal image (ge_layout.h CondLit): every generic row padded
// to the table's common shape, so the loops below are rolled and wave-uniform - two scalar counters, no per-lane clause
// count or length, no unrolled 4 x 4 slots (what made round 3's form spill 72 - 711 scalar registers).

// the even bits of x (bit 2i -> bit i), NF fields (12 at most)
template <int NF = 12> __device__ __forceinline__ uint32_t even_bits(uint32_t x) {
    x &= 0x00555555u;
    x = (x | (x >> 1)) & 0x00333333u; x = (x | (x >> 2)) & 0x000F0F0Fu;
    if (NF > 4) x = (x | (x >> 4)) & 0x00FF00FFu;
    if (NF > 8) x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// v in [lo, hi] for small fields, several players per word (SWAR).  Every field sits in a byte (half-word) of its own with
// the top bit free: (v | G) - lo keeps the top bit iff v >= lo, (hi | G) - v iff v <= hi, and no borrow crosses a field.
// lo4 = lo in every byte, hi4 = hi | 0x80 in every byte (prepared by the host: ge_step.hip build_cond_image).
__device__ __forceinline__ uint32_t in_range_bytes(uint32_t v, uint32_t lo4, uint32_t hi4) {
    return ((v | 0x80808080u) - lo4) & (hi4 - v) & 0x80808080u;
}

// a nibble array (selected_target_id, rounds_as_speaker): player mask of the nibbles in range
template <int NB> __device__ __forceinline__ uint32_t range_nibbles(uint64_t arr, uint32_t lo4, uint32_t hi4) {
    const uint32_t w0 = (uint32_t)arr;
    if (NB <= 4) {
        const uint32_t v = (w0 & 0x0F0Fu) | (((w0 >> 4) & 0x0F0Fu) << 16);          // bytes: players 0, 2, 1, 3
        const uint32_t q = in_range_bytes(v, lo4, hi4) >> 7;
        return (q | (q >> 15) | (q >> 6) | (q >> 21)) & 0xFu;
    }
    const uint32_t ie = in_range_bytes(w0 & 0x0F0F0F0Fu, lo4, hi4), io = in_range_bytes((w0 >> 4) & 0x0F0F0F0Fu, lo4, hi4);
    const uint32_t p = (ie >> 7) | (io >> 6);                    // bits 0 / 1 of byte k = players 2k / 2k + 1
    uint32_t r = (p | (p >> 6) | (p >> 12) | (p >> 18)) & 0xFFu;
    if (NB > 8) {
        const uint32_t w1 = (uint32_t)(arr >> 32) & 0xFFFFu;
        const uint32_t v = (w1 | (w1 << 12)) & 0x0F0F0F0Fu;      // bytes: players 8, 10, 9, 11
        const uint32_t q = in_range_bytes(v, lo4, hi4) >> 7;
        r |= ((q | (q >> 15) | (q >> 6) | (q >> 21)) & 0xFu) << 8;
    }
    return r;
}

// a 2-bit array (lie_index, vote_choice): the range as its set of allowed values - a0 = allowed(0) in the even bits |
// allowed(1) in the odd bits, a1 = allowed(2) | allowed(3) likewise; two bit-selects pick the entry of every field
template <int NB> __device__ __forceinline__ uint32_t range_2bit(uint32_t x, uint32_t a0, uint32_t a1) {
    const uint32_t b0 = x & 0x00555555u, b1 = (x >> 1) & 0x00555555u;
    const uint32_t H = b1 | (b1 << 1);
    const uint32_t t = (H & a1) | (~H & a0);                     // even bits: b1 ? allowed(2) : allowed(0); odd: b1 ? allowed(3) : allowed(1)
    return even_bits<NB>((b0 & (t >> 1)) | (~b0 & t));
}

// a byte array (total_score, up to 255): half-word lanes, two players per word; lo2 = lo in both halves, hi2 = hi | 0x8000
template <int NB> __device__ __forceinline__ uint32_t range_score(const uint32_t *score, uint32_t lo2, uint32_t hi2) {
    uint32_t r = 0;
#pragma unroll
    for (int w = 0; w < (NB + 3) / 4; w++) {
        const uint32_t a = score[w] & 0x00FF00FFu, b = (score[w] >> 8) & 0x00FF00FFu;          // players 4w, 4w + 2 | 4w + 1, 4w + 3
        const uint32_t ia = ((a | 0x80008000u) - lo2) & (hi2 - a) & 0x80008000u;
        const uint32_t ib = ((b | 0x80008000u) - lo2) & (hi2 - b) & 0x80008000u;
        const uint32_t q = ((ia >> 15) & 1u) | ((ib >> 14) & 2u) | ((ia >> 29) & 4u) | ((ib >> 28) & 8u);
        r |= q << (4 * w);
    }
    return r;
}

// what a launch keeps in scalar registers about the table's generic rows
struct CondCtx {
    const unsigned char *img;  // the block's LDS copy of DevTable::cond_img
    CondShape cs;              // ge_layout.h
};

// OR over clauses of AND over literals; `lit(d, e, g, f)` -> the literal's player mask before negation (d = the literal's
// first four words, e = its second four: Werewolf x 12 only; g, f = the slot's wave-uniform flags, CondShape).  One rolled
// loop over the table's NCL x LEN slots with one scalar counter; the slot flags are shifted out of two scalar registers.
// (Measured and rejected, profiles/r04_generic_probe.txt: issuing the next literal's LDS read before evaluating the current
// one, and an unrolled form for up to four slots with the reads grouped - more live registers, the scalar spills back.)
template <int STRIDE, typename LIT>
__device__ __forceinline__ uint32_t eval_cond_image(const CondCtx &cc, uint32_t row_r0, uint32_t all, LIT lit) {
    const uint32_t n = (cc.cs.shape & 7u) * ((cc.cs.shape >> 4) & 7u);                        // wave-uniform
    const unsigned char *p = cc.img + ((row_r0 >> ROW_COND_SLOT_SHIFT) & 31u) * (n * (uint32_t)STRIDE);
    uint64_t G = ((uint64_t)cc.cs.g_hi << 32) | cc.cs.g_lo, F = ((uint64_t)cc.cs.f_hi << 32) | cc.cs.f_lo;
    uint32_t T = 0, m = all;
#pragma nounroll
    for (uint32_t i = n; i != 0u; i--) {
        const uint32_t g = (uint32_t)G & 15u, f = (uint32_t)F & 15u;
        G >>= 4; F >>= 4;
        const uint4 d = *reinterpret_cast<const uint4 *>(p);
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        if (STRIDE > 16) e = *reinterpret_cast<const uint4 *>(p + 16);
        const uint32_t x = lit(d, e, g, f);
        m &= x ^ (d.x >> 16);                                                                 // negated: 0xFFFF in the top half of w
        p += STRIDE;
        if (g & 4u) { T |= m; m = all; }                                                      // the clause ends here
    }
    return T & all;
}

// The same for a table whose shape is known when the kernel is built (the GENERIC template argument of the Two-Truths kernels: 2 =
// 1 clause x 1 literal, 3 = 1 x 2 - what a generated DSL's conditions mostly are; ge_step.hip picks the build from
// DevTable::cond_shape, the rolled loop above serves every other shape): no loop, no scalar counter, every slot's LDS read in
// flight before the first literal is evaluated, clause ends known statically.  A Two-Truths turn is ~270 instructions, of which
// the rolled walk of two slots was ~60: x 1.27 -> x 1.24 of the shipped game's time (profiles/r05_generic_probe.txt).  A 2 x 2
// form was built and measured too: 20 - 43 scalar spills and no gain (x 1.61 either way) - such tables take the rolled walk.
template <int STRIDE, int NCL, int LEN, typename LIT>
__device__ __forceinline__ uint32_t eval_cond_image_fixed(const CondCtx &cc, uint32_t row_r0, uint32_t all, LIT lit) {
    constexpr uint32_t n = NCL * LEN;
    static_assert(n <= 8, "slot flags of a fixed shape come from the low words");
    const unsigned char *p = cc.img + ((row_r0 >> ROW_COND_SLOT_SHIFT) & 31u) * (n * (uint32_t)STRIDE);
    uint4 d[n];
#pragma unroll
    for (uint32_t i = 0; i < n; i++) d[i] = *reinterpret_cast<const uint4 *>(p + i * STRIDE);
    uint32_t T = 0;
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)NCL; c++) {
        uint32_t m = all;
#pragma unroll
        for (uint32_t l = 0; l < (uint32_t)LEN; l++) {
            const uint32_t i = c * LEN + l;
            const uint32_t x = lit(d[i], make_uint4(0u, 0u, 0u, 0u), (cc.cs.g_lo >> (4u * i)) & 15u, (cc.cs.f_lo >> (4u * i)) & 15u);
            m &= x ^ (d[i].x >> 16);
        }
        T |= m;
    }
    return T & all;
}

// (an input made opaque INSIDE the block that uses it: the compare's loop-invariant half - a third of its instructions - is
// otherwise hoisted in front of the loop and runs every turn, whether or not any slot of the table compares that field)
__device__ __forceinline__ uint32_t pin(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

template <int NB> __device__ __forceinline__ uint32_t ww_cond_generic(const WWR<NB> &s, const CondCtx &cc, uint32_t row_r0, uint32_t all) {
    constexpr int STRIDE = NB <= 8 ? (int)sizeof(CondLit) : (int)sizeof(CondLit12);
    return eval_cond_image<STRIDE>(cc, row_r0, all, [&](const uint4 &d, const uint4 &e, uint32_t sg, uint32_t) -> uint32_t {   // d = {w, m0, m1, m2}, e = {m3, m4, m5, -}
        const bool any_base = sg & 1u, any_num = sg & 2u, any_conj = sg & 8u;       // wave-uniform
        uint32_t x = 0;
        if (any_base) {
            uint32_t g;
            if (NB <= 8) {
                g = (s.W[0] & d.y) | (s.W[1] & d.z) | (s.W[2] & d.w);
                g |= g >> 16; g |= g >> 8;
            } else {
                g = (s.W[0] & d.y) | (s.W[1] & d.z) | (s.W[2] & d.w) | (s.W[WWR<NB>::NW > 3 ? 3 : 0] & e.x) |
                    (s.W[WWR<NB>::NW > 4 ? 4 : 0] & e.y) | (s.W[WWR<NB>::NW > 5 ? 5 : 0] & e.z);
                g |= g >> 16;
            }
            x = g;
        }
        if (any_conj) {                                          // single predicates ANDed: permute and fold, like a row's own terms (ww_targets)
            uint32_t X;
            if (NB <= 8) {
                X = (__builtin_amdgcn_perm(s.W[1], s.W[0], d.y) & __builtin_amdgcn_perm(s.W[2], s.W[2], d.z)) ^ d.w;
                X &= X >> 16; X &= X >> 8;
            } else {
                X = (__builtin_amdgcn_perm(s.W[1], s.W[0], d.y) & __builtin_amdgcn_perm(s.W[WWR<NB>::NW > 3 ? 3 : 0], s.W[2], d.z) &
                     __builtin_amdgcn_perm(s.W[WWR<NB>::NW - 1], s.W[WWR<NB>::NW - 2], d.w)) ^ e.x;
                X &= X >> 16;
            }
            x = any_base ? bfi(bit_mask(d.x, 8u), X, x) : X;     // (a conj-only slot's neutral literals are conj literals too)
        }
        if (any_num) {                                           // GE_NUM_SELECTED_TARGET is the pack's only numeric field
            const uint32_t r = range_nibbles<NB>(NB <= 8 ? (uint64_t)pin((uint32_t)s.sel) : ((uint64_t)pin((uint32_t)((uint64_t)s.sel >> 32)) << 32) | pin((uint32_t)s.sel), d.y, d.z);
            x = bfi(bit_mask(d.x, 0u), r, x);                     // by the literal's own kind (a neutral literal of another row sits in this slot as a base set)
        }
        return x;
    });
}

// GSHAPE: the GENERIC template argument - 1 = any shape (rolled walk), 2 / 3 = the table is 1 x 1 / 1 x 2
template <int NB, int GSHAPE = 1> __device__ __forceinline__ uint32_t tt_cond_generic(const TT<NB> &s, const CondCtx &cc, uint32_t row_r0, uint32_t all) {
    const uint32_t W0 = s.speaker | (s.submitted << 16), W1 = s.revealed | (s.can_vote << 16), W2 = s.has_voted;
    auto literal = [&](const uint4 &d, const uint4 &, uint32_t g, uint32_t flds) -> uint32_t {
        // wave-uniform: what the slot holds in some row of the table - a base set, and / or a range over which fields
        const bool any_base = g & 1u, any_conj = g & 8u;
        uint32_t x = 0;
        if (any_base) {
            uint32_t v = (W0 & d.y) | (W1 & d.z) | (W2 & d.w);
            v |= v >> 16;
            x = v;
        }
        if (any_conj) {                                          // two single predicates ANDed: permute and fold (tt_turn's own terms)
            uint32_t X = (__builtin_amdgcn_perm(W1, W0, d.y) & __builtin_amdgcn_perm(W2, W2, d.z)) ^ d.w;
            X &= X >> 16;
            x = any_base ? bfi(bit_mask(d.x, 8u), X, x) : X;
        }
        if (flds) {
            // the literal's field one-hot in w bits 4..7: bit-selects, no compares and no exec-mask regions
            if (flds & 1u) x = bfi(bit_mask(d.x, 4u), range_2bit<NB>(pin(s.lie), d.y, d.z), x);  // GE_NUM_LIE_INDEX
            if (flds & 2u) x = bfi(bit_mask(d.x, 5u), range_2bit<NB>(pin(s.vote), d.y, d.z), x); // GE_NUM_VOTE_CHOICE
            if (flds & 4u) {                                                                     // GE_NUM_TOTAL_SCORE
                uint32_t sc[(NB + 3) / 4];
#pragma unroll
                for (int w = 0; w < (NB + 3) / 4; w++) sc[w] = pin(s.score[w]);
                x = bfi(bit_mask(d.x, 6u), range_score<NB>(sc, d.y, d.z), x);
            }
            if (flds & 8u)                                                                       // GE_NUM_ROUNDS_AS_SPEAKER
                x = bfi(bit_mask(d.x, 7u), range_nibbles<NB>(NB <= 8 ? (uint64_t)pin((uint32_t)s.rounds) : ((uint64_t)pin((uint32_t)(s.rounds >> 32)) << 32) | pin((uint32_t)s.rounds), d.y, d.z), x);
        }
        return x;
    };
    if (GSHAPE == 2) return eval_cond_image_fixed<(int)sizeof(CondLit), 1, 1>(cc, row_r0, all, literal);
    if (GSHAPE == 3) return eval_cond_image_fixed<(int)sizeof(CondLit), 1, 2>(cc, row_r0, all, literal);
    return eval_cond_image<(int)sizeof(CondLit)>(cc, row_r0, all, literal);
}

// ------------------------------------------------------------------ werewolf
// One turn of a room (= one graph run of the reference, v2:1571-1587) is
//   ww_targets()        who must act (PhaseNode's reading of completion_criteria.target_players, v2:1087-1103)
//   ww_queue_actions()  BotBehaviorNode (v2:468) + the Referee's record of each action (bt:204-225), through the wavefront
//                       work queue; ww_phase_branch() and ww_prepare_deal() run in the shadows of its two LDS round trips
//   ww_decide()         PhaseNode (v2:987): phase-0 guard, completion, first matching branch
//   ww_apply_effect()   RefereeNode (v2:619): the entry effect of the phase the room moves to
// WwBuild<NB, LOWOCC> (top of this file) says what differs between the lone-wavefront and the large-batch build.

// what a launch keeps constant for a lane's turns
struct WwCtx {
    const DevRow *rows;        // phase table in LDS
    CondCtx cc;                // the generic rows' literal image in LDS and its shape (GENERIC builds only)
    void *wave_lds;            // this wavefront's WaveLds / WaveLdsLow
    const uint8_t *nth8;       // LDS n-th-set-bit table (large-batch build)
    const uint32_t *ord8;      // LDS slot -> player table (N <= 8)
    bool valid;                // the lane holds a real room (lanes past the end of a segment shadow room 0 and never act)
    uint32_t n, nw, phase0_idx, rkey;
    uint32_t human;            // players the host drives (never acted for here)
    uint32_t term_mask;        // bit r = row r is terminal (no next_phase branch): all a single-turn launch needs of the row it moves to
};

// ---- who must act: target_players.condition AND alive, all players at once.
// The 12 base predicates live packed in s.W; the row carries byte-permute selectors that pull each term's mask out of
// the word pairs (0xFF where the term is elsewhere / absent), so the condition is 2-3 v_perm + AND, XOR with the
// negation mask, and a fold of the term bytes (ge_layout.h DevRow).
template <int NB, bool LOWOCC, int GENERIC>
__device__ __forceinline__ uint32_t ww_targets(const WWR<NB> &s, const DevRow &row, const WwCtx &c, uint32_t alive, uint32_t ALL) {
    using R = WWR<NB>;
    const uint32_t comp = row.r0 & 3u, nterms = (row.r0 >> 8) & 7u;
    uint32_t T = 0;
    const bool generic_row = GENERIC && (row.r0 & ROW_GENERIC);   // its condition is the literal image below, not the row's terms
    if (LOWOCC ? !GENERIC || !generic_row : (comp == COMP_ACTION && !generic_row)) {   // LOWOCC: always evaluated, masked below (no branch)
        uint32_t X;
        if (NB <= 8) {
            X = __builtin_amdgcn_perm(s.W[1], s.W[0], row.r4) & __builtin_amdgcn_perm(s.W[2], s.W[2], row.r5);
            X ^= row.r7;
            X &= X >> 16; X &= X >> 8;                         // 4 terms, one byte each
        } else {
            X = __builtin_amdgcn_perm(s.W[1], s.W[0], row.r4) & __builtin_amdgcn_perm(s.W[R::NW > 3 ? 3 : 0], s.W[2], row.r5) &
                __builtin_amdgcn_perm(s.W[R::NW - 1], s.W[R::NW - 2], row.r6);
            X ^= row.r7;
            X &= X >> 16;                                      // terms 0..1, one half-word each
            if (nterms > 2u) {                                 // no shipped phase has more than two terms
                auto term = [&](uint32_t j) -> uint32_t {
                    const uint32_t e = (row.r1 >> (8u * j)) & 255u;
                    const uint32_t wi = e >> 5;
                    uint32_t word = 0xFFFFFFFFu;               // wi == 7: no term
#pragma unroll
                    for (int k = 0; k < R::NW; k++) word = wi == (uint32_t)k ? s.W[k] : word;
                    const uint32_t m = word >> (e & 31u);
                    return m ^ (uint32_t)((int32_t)(row.r0 << (15u - j)) >> 31);   // term_neg bit j -> 0 / ~0
                };
                X &= term(2) & term(3);
            }
        }
        T = X & alive & (comp == COMP_ACTION ? ALL : 0u);
    }
    if (generic_row && comp == COMP_ACTION)                    // or / in [..] / numeric comparisons: the clause form
        T = ww_cond_generic<NB>(s, c.cc, row.r0, ALL) & alive;
    return T;
}

// ---- PhaseNode, the part that does not depend on this turn's actions (nobody dies before the Referee): phase-0 guard
// (v2:1025-1052), resolver bitset, first matching branch.
struct WwBranch {
    bool open;                 // not the guard turn
    uint32_t qe;               // next row index | its entry effect << 5 (the current row if no branch matches)
};
template <int NB>
__device__ __forceinline__ void ww_phase_branch(const WWR<NB> &s, const DevRow &row, uint32_t phase0_idx, uint32_t alive, uint32_t team_w, WwBranch &b) {
    const bool guard = s.phase == phase0_idx && !(s.flags & FLAG_PHASE0_DONE);
    b.open = !guard;
    const uint32_t w = popc(alive & team_w), g = popc(alive & s.template get<F_TEAM_V>());
    // (prev_eff == EFF_DAY_RESOLVE) << RES_FOLLOWS_DAY | (prev_eff == EFF_NIGHT_RESOLVE) << RES_FOLLOWS_NIGHT, prev_eff = flags bits 1..3,
    // as three simple instructions instead of a field extract, two compares and two selects (a compare or a select costs a SIMD as
    // much as two simple instructions: profiles/r05_encoding_probe.txt): 0xC00 >> 2 * prev_eff has bit 3 set for 4, bit 4 for 3
    static_assert(EFF_DAY_RESOLVE == 4 && RES_FOLLOWS_DAY == 3 && EFF_NIGHT_RESOLVE == 3 && RES_FOLLOWS_NIGHT == 4 && FLAG_PHASE0_DONE == 1,
                  "the constant below is made for these codes");
    const uint32_t follows = (0xC00u >> (s.flags & 0xEu)) & 0x18u;
    const uint32_t C = 1u | ((w == 0u) << RES_WOLVES_ZERO) | ((w >= g) << RES_WOLVES_GE_VILLAGERS) | follows | (1u << RES_OTHERWISE);
    // first branch (DSL order) whose resolver holds: row.r2 has one byte per branch with the bit of
    // its resolver set (0 for absent branches), so the lowest non-zero byte of r2 & (C in every byte) wins
    const uint32_t hit = row.r2 & __builtin_amdgcn_perm(C, C, 0u);   // C (< 256) in every byte
    const uint32_t sh = ctz(hit) & 24u;
    b.qe = hit != 0u ? ((row.r3 >> sh) & 255u) : s.phase;
}

// ---- `deal_now` (wave-uniform, every GE_DEAL_PERIOD-th turn): lanes without a prepared deal compute their next one; then the
// role-assignment values of the prepared deal (what an assignment writes) - except in the packed form, which only the
// assigning lanes expand (ww_apply_effect)
template <int NB, bool LOWOCC, bool SINGLE>
__device__ __forceinline__ void ww_prepare_deal(const WWR<NB> &s, const WwCtx &c, Deal &deal, bool deal_now, uint32_t ALL, WWR<NB> &dealt) {
    using B = WwBuild<NB, LOWOCC, SINGLE>;
    const uint32_t g = deal_next_game<NB>(s);
    if (deal_now && deal.gv != (g | DEAL_VALID)) {             // no deal, or (single-turn Werewolf x 12: an entry of the side plane) one for another game
        deal_roles<NB, B::TABLE, B::DEAL_FORM>(deal, deal_key(c.rkey, g), g, c.n, c.nw, c.nth8);
    }
    if (B::DEAL_FORM != DEAL_PACKED) deal_words<NB, B::DEAL_FORM>(deal, ALL, dealt);
}

// ---- BotBehaviorNode: every due bot acts with probability 3/4, one action per visit (POLICY.md §3).
// Lane = room leaves this step badly balanced, so the wavefront compacts all due (room, player) actions of its 64 rooms
// into one queue in LDS (WaveLds), every lane takes one slot per round, results return by LDS atomic OR.
// shadow1 / shadow2: work that does not depend on this turn's actions, placed behind the first slot read / the result
// read (WwBuild::SHADOW), else run after the queue.
struct WwActs { uint32_t newly, det_v, det_w; };   // who acted now; the Detective's new knowledge (villager / werewolf)

template <int NB, bool LOWOCC, typename S1, typename S2>
__device__ __forceinline__ void ww_queue_actions(WWR<NB> &s, const WwCtx &c, uint32_t T, uint32_t act, bool night, uint32_t alive, uint32_t team_w,
                                                 uint32_t r_det, uint32_t tk, WwActs &out, S1 &&shadow1, S2 &&shadow2, Stamps *stamps) {
    using nib_t = typename WWR<NB>::nib_t;
    using B = WwBuild<NB, LOWOCC>;
    auto *lw = static_cast<typename WaveLdsOf<LOWOCC>::type *>(c.wave_lds);
    const uint32_t todo = c.valid ? (T & ~s.acted & ~c.human) : 0u;
    const uint32_t known = s.det_v | s.det_w;
    const uint32_t kw_alive = s.det_w & alive;
    const uint32_t lo_kw = kw_alive & (0u - kw_alive);       // lowest known living werewolf
    const uint32_t lane = __lane_id();
    const uint32_t cnt = popc(todo);
    // N <= 8: the room's slot -> player map (nibble r = its r-th due bot) from the ord8 table; the read is
    // in flight during the scan, and a slot then needs a shift instead of an n-th-set-bit search
#if GE_QUEUE_PRIO
    if (!LOWOCC) __builtin_amdgcn_s_setprio(GE_QUEUE_PRIO);   // from the first table read and the scan on (Werewolf x 12 -0.7 % against raising it after the scan)
#endif
    uint32_t ord = 0;
    if (B::ORD) ord = c.ord8[todo & 0xFFu];
    uint32_t off, total;
    wave_excl_scan(cnt, off, total);
    out.newly = 0; out.det_v = 0; out.det_w = 0;
    if (!(LOWOCC || total != 0u)) {                         // wave-uniform; LOWOCC: some room almost always has a due bot
#if GE_QUEUE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        shadow1();
        shadow2();
        return;
    }
    // per-room context of an action; `ky`: what the acting role knows (the Detective's memory
    // at night, who the Detective is by day - ww_choose reads only one of the two per kind)
    const uint32_t ky = night ? known : (B::ONEHOT ? (lo_kw != 0u ? r_det : 0u) : r_det);
    const uint32_t kind = B::ONEHOT ? ((1u << act) >> 1) : act;          // one-hot: ACT_WOLF_TARGET = 1 -> bit 0 ...
    const uint4 ctx = B::ORD ? make_uint4(alive | (team_w << 16) | (kind << 28), ky | (lo_kw << 8) | (off << 16) | (lane << 26), ord, tk)
                             : make_uint4(alive | (team_w << 16) | (act << 28), ky | (lo_kw << 16), todo | (off << 16) | (lane << 26), tk);
    // N <= 8: only x, y come back - the results sit 8 bytes apart (GE_RES_PACKED; 16 bytes apart, a 64-bit access of 16 consecutive
    // lanes hits every bank twice: /opt/skills/guides/MI355X_MICROARCH.md "LDS", ds_write_b64 / ds_read_b64 banking)
    constexpr uint32_t RW = (NB <= 8 && GE_RES_PACKED) ? 2u : 4u;      // words per room in `res`
    uint32_t *const res_w = reinterpret_cast<uint32_t *>(lw->res);
    if (NB <= 8) *reinterpret_cast<uint2 *>(res_w + RW * lane) = make_uint2(0u, 0u);
    else lw->res[lane] = make_uint4(0u, 0u, 0u, 0u);
    // Queue slot -> owning room.  A room with cnt due bots owns slots [off, off + cnt); it writes
    // NB slots from `off` on, highest first (immediate offsets, no per-slot address or
    // predicate).  The surplus writes land in the ranges of the rooms after it and are
    // overwritten by their owners: an owner's write to its r-th slot is issued at step r,
    // any intruder's at a step > r, i.e. earlier.  Rooms without a due bot do not write
    // (they would tie with the next owner inside one instruction).
    if (LOWOCC) {
        auto *lo = reinterpret_cast<WaveLdsLow *>(lw);
        // no predicate here either: a room without a due bot writes to a scratch range behind the queue
        uint4 *qp = lo->slot + (cnt != 0u ? off : 64u * 13u + lane);
#pragma unroll
        for (int j = NB - 1; j >= 0; j--) {
            qp[j] = ctx;
            asm volatile("" ::: "memory");             // the stores must issue in this order
        }
    } else {
        auto *hi = reinterpret_cast<WaveLds *>(lw);
        hi->ctx[lane] = ctx;
        if (cnt != 0u) {
            uint8_t *qp = hi->queue + off;
#pragma unroll
            for (int j = NB - 1; j >= 0; j--) {
                qp[j] = (uint8_t)lane;
                asm volatile("" ::: "memory");
            }
        }
    }
    wave_sync();
    // slot k -> the owning room's context (slots past `total` hold stale entries: computed like the
    // others, result dropped).  The first round's read is issued BEFORE the shadow work below.
    auto fetch = [&](uint32_t k) -> uint4 {
        if (LOWOCC) return reinterpret_cast<WaveLdsLow *>(lw)->slot[k];
        auto *hi = reinterpret_cast<WaveLds *>(lw);
        return hi->ctx[hi->queue[k] & 63u];
    };
    uint4 c4 = fetch(lane);
    if (B::SHADOW) shadow1();
    // at least one round (total == 0: every slot is stale and dropped): the loop is left BEFORE the next
    // round's read is issued, so nothing of the queue is in flight behind it
    for (uint32_t base = 0;; base += 64u) {
        const uint32_t k = base + lane;
        if (GE_STAMPS == 1 && stamps && base == 0u) { asm volatile("" :: "v"(c4.x)); stamps->mark(1); }   // [.. first slot in registers]
        uint32_t L, i, know, lokw;
        if (B::ORD) {
            L = c4.y >> 26;
            const uint32_t rank = (k - ((c4.y >> 16) & 0x3FFu)) & 7u;       // this slot = the rank-th due bot of room L
            i = (c4.z >> (4u * rank)) & 7u;
            know = c4.y & 0xFFu; lokw = (c4.y >> 8) & 0xFFu;
        } else {
            L = c4.z >> 26;
            const uint32_t due = c4.z & 0xFFFFu, rank = (k - ((c4.z >> 16) & 0x3FFu)) & 15u;
            i = (LOWOCC ? nth_set_bit<NB>(due | (1u << 31), rank) : nth_set_bit_lds<NB>(c.nth8, due, rank)) & 15u;
            know = c4.y & 0xFFFFu; lokw = c4.y >> 16;
        }
        const uint32_t d = draw(c4.w, i);
        const bool go = k < total && (d & 3u) != 0u;
        if (LOWOCC) {
            // the choice is computed for every slot and only the result is predicated: a
            // conditional block would split the slot read in two dependent LDS round trips
            uint32_t ch = B::ONEHOT ? ww_choose_onehot8(c4.x, i, d, know, lokw, know)
                                    : ww_choose<NB, false>(c4.x >> 28, i, d, c4.x & 0xFFFFu, (c4.x >> 16) & 0xFFFu, know, lokw, know, c.nth8);
            if (B::PIN_CHOICE) asm volatile("" : "+v"(ch));   // stays outside the exec-masked block below
            if (go) {
                uint32_t *r = res_w + RW * L;
                if (!B::ONE_ATOMIC) atomicOr(r, 1u << i);
                atomicOr(r + 1 + (i >> 3), ch << (4u * (i & 7u)));
            }
        } else if (go) {
            const uint32_t ch = ww_choose<NB, true>(c4.x >> 28, i, d, c4.x & 0xFFFFu, (c4.x >> 16) & 0xFFFu, know, lokw, know, c.nth8);
            uint32_t *r = res_w + RW * L;
            if (NB <= 8 && GE_RES_PACKED && GE_RES_ATOMIC64) {
                // both result words of the room in ONE 64-bit LDS atomic (the pair is 8-byte aligned): who acted | the choice nibble
                atomicOr(reinterpret_cast<unsigned long long *>(r), (unsigned long long)(1u << i) | ((unsigned long long)(ch << (4u * i)) << 32));
            } else {
                atomicOr(r, 1u << i);
                atomicOr(r + 1 + (i >> 3), ch << (4u * (i & 7u)));
            }
        }
        if (base + 64u >= total) break;                // wave-uniform
        c4 = fetch(k + 64u);
    }
    wave_sync();
    const uint4 r = NB <= 8 ? make_uint4(reinterpret_cast<const uint2 *>(res_w + RW * lane)->x, reinterpret_cast<const uint2 *>(res_w + RW * lane)->y, 0u, 0u)
                            : lw->res[lane];
    if (B::SHADOW) {                                       // shadow of the result read
        shadow2();
        // keeps the slot registers reserved up to here: reusing them for the work above would make the
        // compiler wait for the queue's LDS traffic first (a read into them may be in flight)
        asm volatile("" :: "v"(c4.x), "v"(c4.y), "v"(c4.z), "v"(c4.w));
    }
    uint32_t newly = r.x;
#if GE_QUEUE_PRIO
    if (!LOWOCC) { asm volatile("" :: "v"(newly)); __builtin_amdgcn_s_setprio(0); }
#endif
    if (GE_STAMPS == 1 && stamps) { asm volatile("" :: "v"(newly)); stamps->mark(2); }                 // [.. results in registers]
    const nib_t got = NB > 8 ? (nib_t)(((uint64_t)r.z << 32) | r.y) : (nib_t)r.y;
    nib_t m15;                                       // nibble mask of the players who acted now
    if (B::TABLE && !B::ONE_ATOMIC) {                // large-batch builds: the go mask through the spread8 table (see plurality)
        const uint32_t *spread8 = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(c.rows) + IMG_SPREAD8);
        m15 = (nib_t)spread8[newly & 0xFFu];
        if (NB > 8) m15 |= (nib_t)((uint64_t)spread8[(newly >> 8) & 0xFFu] << 32);
    } else {
        m15 = nib_nonzero(got);                      // a choice is >= 1, so a nibble is set iff that player acted
    }
    if (B::ONE_ATOMIC) {                             // bit i = nibble i is non-zero
        uint32_t x = (uint32_t)m15 & 0x11111111u;
        x = (x | (x >> 3)) & 0x03030303u; x = (x | (x >> 6)) & 0x000F000Fu;
        newly = (x | (x >> 12)) & 0xFFu;
    }
    s.choice = (s.choice & ~m15) | got;
    // RefereeNode (A): record the action (bt:204-225 update_player_state)
    s.sel = night ? ((s.sel & ~m15) | got) : s.sel;
    {
        const uint32_t ch = (uint32_t)(got >> (4u * ctz(newly | 0x80000000u))) & 15u;
        const uint32_t tb = (act == ACT_DETECTIVE && newly) ? (1u << ((ch - 1u) & 15u)) : 0u;
        out.det_w = tb & team_w;
        out.det_v = tb & ~team_w;
    }
    out.newly = newly;
    if (!B::SHADOW) { shadow1(); shadow2(); }
}

// ---- PhaseNode: completion (every target player has acted in this visit) and the chosen branch
template <int NB, bool LOWOCC>
__device__ __forceinline__ uint32_t ww_decide(const WWR<NB> &s, uint32_t comp, uint32_t T, const WwBranch &b) {
    if (WwBuild<NB, LOWOCC>::SEL_OPEN) {
        // no short-circuit evaluation: && / || became three nested exec-mask regions here (see ww_choose)
        const uint32_t open = (uint32_t)b.open & ((uint32_t)(comp != COMP_ACTION) | (uint32_t)((T & ~s.acted) == 0u));
        return sel32(open != 0u, b.qe, s.phase);
    }
    const bool open = b.open && (comp != COMP_ACTION || (T & ~s.acted) == 0u);
    return open ? b.qe : s.phase;
}

// ---- RefereeNode (B): the effect of entering row q = qe & 31 (qe >> 5 = its entry effect), then the move itself
template <int NB, bool LOWOCC, bool SINGLE>
__device__ __forceinline__ void ww_apply_effect(WWR<NB> &s, const DevRow &row, const DevRow &qrow, const WwCtx &c, uint32_t qe, uint32_t alive, uint32_t ALL, uint32_t turn,
                                                Deal &deal, WWR<NB> &dealt) {
    using R = WWR<NB>;
    using nib_t = typename R::nib_t;
    using B = WwBuild<NB, LOWOCC, SINGLE>;
    const uint32_t q = qe & 31u, eff = qe >> 5, p_eff = (row.r0 >> 5) & 7u;
    // night / day resolution: the plurality victim dies unless the (highest-id living) Doctor guards it
    auto resolve = [&](bool on, bool day) {
        const uint32_t voters = day ? (alive & s.acted) : (alive & s.template get<F_WOLF>());
        const uint32_t victim = plurality<NB, nib_t, B::TABLE>(day ? s.choice : s.sel, voters, c.rows);
        const uint32_t docs = alive & s.template get<F_DOC>();
        const uint32_t guarded = (uint32_t)(s.sel >> (4u * (31u - (uint32_t)__clz((int)(docs | 1u))))) & 15u;
        const uint32_t protect = (!day && docs) ? guarded : 0u;
        const uint32_t bit = (on && victim != 0u && victim != protect) ? (1u << ((victim - 1u) & 15u)) : 0u;
        s.template clear<F_ALIVE>(bit); s.template clear<F_CAN_VOTE>(bit); s.template clear<F_ELIG>(bit);
        s.template set<F_REVEALED>(bit);
    };
    // role assignment: the deal of this game was normally prepared ahead (ww_prepare_deal, every GE_DEAL_PERIOD-th turn,
    // for all lanes of the wavefront at once); fall back to dealing here if it was not
    const bool is_assign = eff == EFF_ASSIGN_ROLES;
    if (is_assign && deal.gv != (s.games | DEAL_VALID)) {
        deal_roles<NB, B::TABLE, B::DEAL_FORM>(deal, deal_key(c.rkey, s.games), s.games, c.n, c.nw, c.nth8);
        if (B::DEAL_FORM != DEAL_PACKED) deal_words<NB, B::DEAL_FORM>(deal, ALL, dealt);
    }
    // the fields a deal replaces, as masks over the packed predicate words
    R dmask;
#pragma unroll
    for (int k = 0; k < R::NW; k++) dmask.W[k] = 0;
    dmask.template set<F_VIL>(R::FM); dmask.template set<F_WOLF>(R::FM); dmask.template set<F_DOC>(R::FM); dmask.template set<F_DET>(R::FM);
    dmask.template set<F_TEAM_W>(R::FM); dmask.template set<F_TEAM_V>(R::FM); dmask.template set<F_SECRET>(R::FM); dmask.template set<F_ELIG>(R::FM);
    if (LOWOCC) {
        // lone wavefront: every divergent block costs an exec-mask sequence and a branch bubble, and both
        // effects are entered by some room of the wavefront on most turns anyway - so both are evaluated
        // for every lane and applied by selects
        if (B::DEAL_FORM == DEAL_PACKED) deal_words<NB, B::DEAL_FORM>(deal, ALL, dealt);
#pragma unroll
        for (int k = 0; k < R::NW; k++) s.W[k] = is_assign ? ((s.W[k] & ~dmask.W[k]) | dealt.W[k]) : s.W[k];
        deal.gv = is_assign ? 0u : deal.gv;
        resolve(eff == EFF_NIGHT_RESOLVE || eff == EFF_DAY_RESOLVE, eff == EFF_DAY_RESOLVE);
    } else if (is_assign) {
        if (B::DEAL_FORM == DEAL_PACKED) deal_words<NB, B::DEAL_FORM>(deal, ALL, dealt);
#pragma unroll
        for (int k = 0; k < R::NW; k++) s.W[k] = (s.W[k] & ~dmask.W[k]) | dealt.W[k];
        deal.gv = 0u;
    } else if (eff == EFF_NIGHT_RESOLVE || eff == EFF_DAY_RESOLVE) {
#if GE_RESOLVE_PRIO
        __builtin_amdgcn_s_setprio(GE_RESOLVE_PRIO);
#endif
        resolve(true, eff == EFF_DAY_RESOLVE);
#if GE_RESOLVE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    const bool nbeg = eff == EFF_NIGHT_BEGIN;
    s.template clear<F_SUB>(nbeg ? R::FM : 0u);
    s.sel = nbeg ? nib_t(0) : s.sel;
    s.acted = 0; s.choice = 0;
    s.flags = (s.flags & FLAG_PHASE0_DONE) | (p_eff << 1);
    s.prev = s.phase;
    s.phase = q;
    const bool terminal = SINGLE ? ((c.term_mask >> q) & 1u) != 0u : ((qrow.r0 >> 11) & 7u) == 0u;
    s.end_turn = (terminal && s.end_turn == END_NONE) ? (turn < 0xFFFEu ? turn : 0xFFFEu) : s.end_turn;
}

// `row` is the table row of s.phase, kept in registers across turns: LDS is read only on a transition (a single-turn
// build leaves it as it is: nobody reads it after the turn).
// tk_io: in = turn_key(rkey, turn), out = the next turn's key (computed in an LDS wait shadow; not in a single-turn build).
// ev_*: this turn's logged actions (who acted, what they chose) for the optional event trace.
template <int NB, bool LOWOCC, int GENERIC = false, bool SINGLE = false>
__device__ __forceinline__ void ww_turn(WWR<NB> &s, DevRow &row, const WwCtx &c, uint32_t turn, uint32_t &tk_io, bool trace, Deal &deal, bool deal_now,
                                        uint32_t &ev_newly, uint64_t &ev_choice, Stamps *stamps = nullptr) {
    using R = WWR<NB>;
    using B = WwBuild<NB, LOWOCC, SINGLE>;
    const uint32_t ALL = (1u << c.n) - 1u;
    const uint32_t comp = row.r0 & 3u, act = (row.r0 >> 2) & 7u;
    const uint32_t alive = s.template get<F_ALIVE>(), team_w = s.template get<F_TEAM_W>(), r_det = s.template get<F_DET>();
    const bool night = act >= ACT_WOLF_TARGET && act <= ACT_DETECTIVE;

    const uint32_t T = ww_targets<NB, LOWOCC, GENERIC>(s, row, c, alive, ALL);
    if (GE_STAMPS == 1 && stamps) { asm volatile("" :: "v"(T)); stamps->mark(0); }        // [end of previous turn .. row in registers]

    WwBranch br;
    R dealt;
    if (B::DEAL_EARLY) ww_prepare_deal<NB, LOWOCC, SINGLE>(s, c, deal, deal_now, ALL, dealt);
    uint32_t tk_next = 0;
    WwActs acts;
    ww_queue_actions<NB, LOWOCC>(s, c, T, act, night, alive, team_w, r_det, tk_io, acts,
        [&]() {
            ww_phase_branch<NB>(s, row, c.phase0_idx, alive, team_w, br);
            if (B::SHADOW) asm volatile("" : "+v"(br.qe));     // stays in the shadow: not sunk below the queue loop
        },
        [&]() {
            if (!SINGLE) tk_next = turn_key(c.rkey, turn + 1u);
            if (!B::DEAL_EARLY) ww_prepare_deal<NB, LOWOCC, SINGLE>(s, c, deal, deal_now, ALL, dealt);
            if (B::SHADOW) {
                if (!SINGLE) asm volatile("" : "+v"(tk_next));
                if (B::DEAL_FORM != DEAL_PACKED && !B::DEAL_EARLY) {
#pragma unroll
                    for (int k = 0; k < R::NW; k++) asm volatile("" : "+v"(dealt.W[k]));
                }
            }
        }, stamps);
    s.acted |= acts.newly;
    s.template set<F_SUB>(night ? acts.newly : 0u);            // night_action_submitted
    ev_newly = acts.newly;
    if (trace) {                                              // wave-uniform
        uint64_t m = 0;                                       // nibble mask of the new actors
#pragma unroll
        for (int i = 0; i < NB; i++) m |= (uint64_t)((acts.newly >> i) & 1u) << (4 * i);
        ev_choice = (uint64_t)s.choice & nib_fill(m);
    }

    tk_io = tk_next;
    s.flags |= FLAG_PHASE0_DONE;                               // set by the guard turn; already set afterwards
    const uint32_t qe = ww_decide<NB, LOWOCC>(s, comp, T, br);
    {   // investigated_alignments[c] = team(c): an assignment, so a stale entry is replaced
        // (the guard turn has no actions: both masks are 0)
        const uint32_t seen = acts.det_v | acts.det_w;
        s.det_v = (s.det_v & ~seen) | acts.det_v;
        s.det_w = (s.det_w & ~seen) | acts.det_w;
    }
    if (SINGLE) {
        if ((qe & 31u) != s.phase) ww_apply_effect<NB, LOWOCC, true>(s, row, row, c, qe, alive, ALL, turn, deal, dealt);
    } else if ((qe & 31u) != s.phase) {
        const DevRow qrow = lds_row<!LOWOCC>(c.rows, qe & 31u);         // LDS read in flight during the effect: first used at its end
        ww_apply_effect<NB, LOWOCC, false>(s, row, qrow, c, qe, alive, ALL, turn, deal, dealt);
        row = qrow;
    }
}

// ------------------------------------------------------------------ two truths and a lie
// per-player mask of rounds_as_speaker >= R (kept beside the room in registers: "whose turn is next"
// and "all rounds done" are then one bit operation instead of a scan over the nibble array)
template <int NB> __device__ __forceinline__ uint32_t tt_done_mask(uint64_t rounds_nib, uint32_t R) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) m |= (((uint32_t)(rounds_nib >> (4 * i)) & 15u) >= R ? 1u : 0u) << i;
    return m;
}

// QUEUE: bot actions through the wavefront work queue (see ww_turn) - pays from 8 players on, where the
// first turn of a vote has 7-11 due bots in some room of every wavefront; TABLE: n-th-set-bit from LDS
// SINGLE: a one-turn launch - the row the room moves to is not fetched (only whether it is terminal: term_mask)
template <int NB, bool QUEUE, bool TABLE, int GENERIC = false, bool SINGLE = false>
__device__ __forceinline__ void tt_turn(TT<NB> &s, uint32_t &done, DevRow &row, const DevRow *rows, const CondCtx &cc, void *wave_lds, const uint8_t *nth8,
                                        bool valid, uint32_t n, uint32_t rounds,
                                        uint32_t phase0_idx, uint32_t rkey, uint32_t turn,
                                        bool trace, uint32_t human, uint32_t term_mask, uint32_t &ev_newly, uint64_t &ev_choice) {
    // done: tt_done_mask of s.rounds, maintained here (the caller derives it when it loads or replaces s)
    const uint32_t ALL = (1u << n) - 1u;
    const uint32_t comp = row.r0 & 3u, act = (row.r0 >> 2) & 7u, p_eff = (row.r0 >> 5) & 7u;
    const uint32_t nterms = (row.r0 >> 8) & 7u;
    const uint32_t tk = turn_key(rkey, turn);

    uint32_t T = 0;
    const bool generic_row = GENERIC && (row.r0 & ROW_GENERIC);
    if (comp == COMP_ACTION && !generic_row) {
        // the 5 base predicates, two per 32-bit word; terms 0..1 by byte permute (see ww_turn, ge_layout.h DevRow)
        const uint32_t W0 = s.speaker | (s.submitted << 16), W1 = s.revealed | (s.can_vote << 16), W2 = s.has_voted;
        uint32_t X = __builtin_amdgcn_perm(W1, W0, row.r4) & __builtin_amdgcn_perm(W2, W2, row.r5);
        X ^= row.r7;
        X &= X >> 16;
        if (nterms > 2u) {                                     // no shipped phase has more than two terms
            auto term = [&](uint32_t j) -> uint32_t {
                const uint32_t e = (row.r1 >> (8u * j)) & 255u;
                const uint32_t wi = e >> 5;
                uint32_t word = 0xFFFFFFFFu;
                word = wi == 0u ? W0 : word; word = wi == 1u ? W1 : word; word = wi == 2u ? W2 : word;
                const uint32_t m = word >> (e & 31u);
                return m ^ (uint32_t)((int32_t)(row.r0 << (15u - j)) >> 31);   // term_neg bit j -> 0 / ~0
            };
            X &= term(2) & term(3);
        }
        T = X & ALL;
    }
    if (generic_row && comp == COMP_ACTION)                    // the clause form (see ww_turn)
        T = tt_cond_generic<NB, GENERIC>(s, cc, row.r0, ALL);

    uint32_t newly = 0;
    {
        uint32_t todo = valid ? (T & ~s.acted & ~human) : 0u;
        const bool a_stm = act == ACT_TT_STATEMENTS, a_lie = act == ACT_TT_LIE, a_vote = act == ACT_TT_VOTE;
        if (!QUEUE) {
            while (todo) {
                const uint32_t i = ctz(todo);
                todo &= todo - 1u;
                const uint32_t d = draw(tk, i);
                const bool go = (d & 3u) != 0u;
                const uint32_t c = a_stm ? 1u : 1u + pick(d, 3u);
                const uint32_t sh = 2u * i;
                const uint32_t clr = ~(3u << sh), put = c << sh;
                s.choice = go ? ((s.choice & clr) | put) : s.choice;
                newly |= go ? (1u << i) : 0u;
                s.lie = (go && a_lie) ? ((s.lie & clr) | put) : s.lie;
                s.vote = (go && a_vote) ? ((s.vote & clr) | put) : s.vote;
            }
        } else {
            // same slot protocol as ww_turn: prefix sum of the due counts, lane ids written highest slot
            // first with immediate offsets, one slot per lane and round, results by LDS atomic OR
            auto *lw = static_cast<WaveLds *>(wave_lds);
            const uint32_t lane = __lane_id();
            const uint32_t cnt = popc(todo);
            uint32_t off, total;
            wave_excl_scan(cnt, off, total);
            if (total != 0u) {                                  // wave-uniform
#if GE_QUEUE_PRIO
                if (TABLE) __builtin_amdgcn_s_setprio(GE_QUEUE_PRIO);
#endif
                lw->ctx[lane] = make_uint4(a_stm ? 1u : 0u, 0u, todo | (off << 16) | (lane << 26), tk);
                lw->res[lane] = make_uint4(0u, 0u, 0u, 0u);
                if (cnt != 0u) {
                    uint8_t *qp = lw->queue + off;
#pragma unroll
                    for (int j = NB - 1; j >= 0; j--) {
                        qp[j] = (uint8_t)lane;
                        asm volatile("" ::: "memory");             // the stores must issue in this order
                    }
                }
                wave_sync();
                for (uint32_t base = 0; base < total; base += 64u) {
                    const uint32_t k = base + lane;
                    const uint4 c4 = lw->ctx[lw->queue[k] & 63u]; // slots past `total`: stale, result dropped
                    const uint32_t L = c4.z >> 26;
                    const uint32_t due = c4.z & 0xFFFFu, rank = (k - ((c4.z >> 16) & 0x3FFu)) & 15u;
                    const uint32_t i = (TABLE ? nth_set_bit_lds<NB>(nth8, due, rank) : nth_set_bit<NB>(due | (1u << 31), rank)) & 15u;
                    const uint32_t d = draw(c4.w, i);
                    if (k < total && (d & 3u) != 0u) {
                        const uint32_t c = c4.x ? 1u : 1u + pick(d, 3u);
                        uint32_t *r = reinterpret_cast<uint32_t *>(&lw->res[L]);
                        atomicOr(r, 1u << i);
                        atomicOr(r + 1, c << (2u * i));
                    }
                }
                wave_sync();
                const uint4 r = lw->res[lane];
                newly = r.x;
#if GE_QUEUE_PRIO
                if (TABLE) { asm volatile("" :: "v"(newly)); __builtin_amdgcn_s_setprio(0); }
#endif
                const uint32_t got = r.y;                        // 2 bits per player, c >= 1 for every actor
                const uint32_t t1 = (got | (got >> 1)) & 0x00555555u, m2 = t1 | (t1 << 1);
                s.choice = (s.choice & ~m2) | got;
                s.lie = a_lie ? ((s.lie & ~m2) | got) : s.lie;
                s.vote = a_vote ? ((s.vote & ~m2) | got) : s.vote;
            }
        }
    }
    s.acted |= newly;
    ev_newly = newly;
    if (trace) {
        uint64_t c4 = 0;                                      // 2-bit choices widened to the nibble form of the trace
#pragma unroll
        for (int i = 0; i < NB; i++)
            c4 |= (uint64_t)(((newly >> i) & 1u) ? ((s.choice >> (2 * i)) & 3u) : 0u) << (4 * i);
        ev_choice = c4;
    }
    if (act == ACT_TT_STATEMENTS) s.submitted |= newly;
    if (act == ACT_TT_VOTE) s.has_voted |= newly;

    if (s.phase == phase0_idx && !(s.flags & FLAG_PHASE0_DONE)) {
        s.flags |= FLAG_PHASE0_DONE;
        return;
    }
    uint32_t qe = s.phase;
    if (comp != COMP_ACTION || (T & ~s.acted) == 0u) {
        const uint32_t all_done = (done & ALL) == ALL;         // every player has spoken R rounds
        const uint32_t C = 1u | (all_done << RES_ALL_ROUNDS_DONE) | (1u << RES_OTHERWISE);
        const uint32_t hit = row.r2 & __builtin_amdgcn_perm(C, C, 0u);       // see ww_turn
        const uint32_t sh = ctz(hit) & 24u;
        qe = hit != 0u ? ((row.r3 >> sh) & 255u) : qe;
    }
    const uint32_t q = qe & 31u;
    if (q == s.phase) return;

    DevRow qrow = row;
    if (!SINGLE) qrow = lds_row<TABLE>(rows, q);                      // in flight during the effect (see ww_turn)
    const uint32_t eff = qe >> 5;
    if (eff == EFF_TT_ROUND_START) {
        const uint32_t cand = ALL & ~done;                     // lowest id that has not spoken R rounds yet
        const uint32_t speaker = cand & (0u - cand);
        s.speaker = speaker; s.can_vote = ALL & ~speaker;
        s.submitted = 0; s.lie = 0; s.revealed = 0; s.vote = 0; s.has_voted = 0;
    } else if (eff == EFF_TT_REVEAL) {
        s.revealed |= s.speaker;
    } else if (eff == EFF_TT_SCORE) {
        // all voters at once: a voter scores if its 2-bit vote equals the speaker's lie index, else the
        // speaker does (tt:5-6); scores are bytes, 4 players per word
        const bool on = s.speaker != 0u;
        const uint32_t sp = ctz(s.speaker | 0x80000000u) & 15u;
        const uint32_t lie = (s.lie >> (2u * sp)) & 3u;
        const uint32_t x = s.vote ^ (lie * 0x00555555u);       // lie index replicated into every field
        const uint32_t eq = even_bits(~(x | (x >> 1)));        // field == 0  <=>  vote == lie
        const uint32_t spbit = s.speaker & (0u - s.speaker);    // written states may flag several: the lowest one counts
        const uint32_t voters = on ? (s.has_voted & ~spbit) : 0u;
        const uint32_t right = eq & voters, fooled = popc(~eq & voters);
#pragma unroll
        for (int w = 0; w < (NB + 3) / 4; w++) {
            const uint32_t nib = (right >> (4 * w)) & 15u;       // 4 players -> bit 0 of 4 bytes
            s.score[w] += ((nib * 0x00204081u) & 0x01010101u) + ((sp >> 2) == (uint32_t)w ? fooled << (8u * (sp & 3u)) : 0u);
        }
        const uint32_t had = (uint32_t)(s.rounds >> (4u * sp)) & 15u;
        s.rounds += (uint64_t)(on ? 1u : 0u) << (4u * sp);
        done |= (on && had + 1u >= rounds) ? spbit : 0u;
    }
    s.acted = 0; s.choice = 0;
    s.flags = (s.flags & FLAG_PHASE0_DONE) | (p_eff << 1);
    s.prev = s.phase;
    s.phase = q;
    const bool terminal = SINGLE ? ((term_mask >> q) & 1u) != 0u : ((qrow.r0 >> 11) & 7u) == 0u;
    if (!SINGLE) row = qrow;
    if (terminal && s.end_turn == END_NONE) s.end_turn = turn < 0xFFFEu ? turn : 0xFFFEu;
}
#endif  // __HIPCC__

}  // namespace ge
