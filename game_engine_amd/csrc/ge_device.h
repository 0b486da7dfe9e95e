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
#ifndef GE_DPP_SCAN
#define GE_DPP_SCAN 1
#endif
// A/B switches (tools/ab.sh): GE_SHADOW = action-independent work placed in the shadows of the action queue's two
// LDS round trips; GE_ORD = queue slots find their player through the ord8 table (werewolf N <= 8)
#ifndef GE_SHADOW
#define GE_SHADOW 1
#endif
#ifndef GE_ORD
#define GE_ORD 1
#endif
// the shadows (and the pins that keep their values live across the queue) in the large-batch build too.  Measured
// (profiles/r02_ab_occupancy.txt): Werewolf x 8 gains 1.7 % from them at 1 M rooms (68 -> 71 VGPRs, still 7 wavefronts per
// SIMD); Werewolf x 12 is better off without them and held to 80 VGPRs = 6 wavefronts per SIMD (2 spilled registers):
// 21.96 -> 21.53 us/turn at 2 M rooms
#ifndef GE_SHADOW_HI
#define GE_SHADOW_HI 1
#endif
// diagnostic build (-DGE_STAMPS=1, tools/stamps.py): s_memtime stamps at points of the werewolf turn where no LDS
// operation is outstanding anyway, accumulated per wavefront; never in the product build
#ifndef GE_STAMPS
#define GE_STAMPS 0
#endif
// branch diet of the lone-wavefront build, A/B on MI355X at 65 536 rooms (gpurun_out/abn_branch.txt, us/turn at fuse 64):
// none 1.442; GE_TPL_TRACE (turn loop compiled per trace setting: two always-taken wave-uniform branches less) 1.420;
// GE_GO_BRANCHLESS (every queue slot ORs a result, zeros if it does not act) 1.468 - worse, the extra LDS atomics cost
// more than the branch; GE_UNLIKELY (fallback role deal hinted out of line) 1.446; GE_UNROLL2 (two turns per trip) 1.439
#ifndef GE_TPL_TRACE
#define GE_TPL_TRACE 1
#endif
#ifndef GE_GO_BRANCHLESS
#define GE_GO_BRANCHLESS 0
#endif
#ifndef GE_UNLIKELY
#define GE_UNLIKELY 0
#endif
#ifndef GE_UNROLL2
#define GE_UNROLL2 0
#endif
// role deals are prepared ahead every GE_DEAL_PERIOD-th turn (a power of two; a game is longer, and a room whose deal
// is not ready when it needs one deals on the spot)
#ifndef GE_DEAL_PERIOD
#define GE_DEAL_PERIOD 16
#endif
// werewolf N <= 8: a queue slot returns its result with ONE LDS atomic (the choice into the actor's nibble); the room derives
// who acted from the non-zero nibbles instead of receiving a second, go-mask atomic
#ifndef GE_ONE_ATOMIC
#define GE_ONE_ATOMIC 1
#endif
// more of the lone-wavefront build's exec-mask regions turned into data flow (profiles/r02_ab_sel_pin.txt): the victim /
// protection selects of a resolution (GE_SEL_RESOLVE: no effect, off); the choice computed outside the `acts this turn`
// region of a queue slot (GE_PIN_CHOICE: Werewolf x 12 1.787 -> 1.751 us/turn at 65 536 rooms, x 8 1.313 -> 1.327: on for N > 8)
#ifndef GE_SEL_RESOLVE
#define GE_SEL_RESOLVE 0
#endif
#ifndef GE_PIN_CHOICE
#define GE_PIN_CHOICE 1
#endif
// GE_SEL_OPEN: the completion test without short-circuit evaluation (&& / || had become three nested exec-mask regions):
// C2 1.308 -> 1.282 us/turn (profiles/r02_ab_open_tpldeal.txt); GE_SEL_NEED: the same for the fallback-deal test, no effect
#ifndef GE_SEL_OPEN
#define GE_SEL_OPEN 1
#endif
#ifndef GE_SEL_NEED
#define GE_SEL_NEED 0
#endif

// Round 3 of the lone-wavefront diet (werewolf N <= 8 only; every instruction of a lone wavefront is a >= 4-cycle issue slot,
// and gfx950 needs two wait states between a VALU write of VCC / an SGPR and a VALU read of it - a dependent
// v_cmp -> v_cndmask pair costs a third slot for the s_nop the compiler has to put between them):
// GE_ACT_ONEHOT: a queue slot carries the action kind one-hot; selects by v_bfe_i32 masks + v_bfi instead of compares
// GE_NTH_SWAR:   n-th set bit of the candidate mask from nibble prefix counts (one multiply) instead of a binary search
// GE_PK_KEYS:    the plurality's max over (count << 4 | 15 - id) keys with packed 16-bit max
// profiles/r02_ab_onehot_swar_pk.txt, us/turn at 64 fused turns, 65 536 / 1 048 576 rooms: none 1.280 / 8.50; one-hot 1.257 / 8.49;
// SWAR n-th bit 1.266 / 8.48; packed keys 1.288 / 8.39; all three 1.222 / 8.33 (alone the packed keys lose 0.6 % at C2, together with the others they gain 1 %)
#ifndef GE_ACT_ONEHOT
#define GE_ACT_ONEHOT 1
#endif
#ifndef GE_NTH_SWAR
#define GE_NTH_SWAR 1
#endif
#ifndef GE_PK_KEYS
#define GE_PK_KEYS 1
#endif

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
template <int NB, typename nib_t, bool PK = false>
__device__ __forceinline__ uint32_t plurality(nib_t votes, uint32_t voters) {
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
        if (PK && NB == 8) {
            // two keys per register, 16 bits each: counters 1 / 5 sit at bits 4..7 of the two halves already (count << 4),
            // 2 / 6, 3 / 7 and 4 after a shift; player 8's count comes from bit 3 of the vote nibbles
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            auto pk = [](uint32_t a) { return __builtin_bit_cast(u16x2, a); };
            const uint32_t c8 = popc(v32 & 0x88888888u);
            const uint32_t a = (tally & 0x00F000F0u) | 0x000A000Eu;                 // ids 1, 5: 15 - id = 14, 10
            const uint32_t b = ((tally >> 4) & 0x00F000F0u) | 0x0009000Du;          // ids 2, 6
            const uint32_t c = ((tally >> 8) & 0x00F000F0u) | 0x0008000Cu;          // ids 3, 7
            const uint32_t d = ((tally >> 12) & 0x000000F0u) | (c8 << 20) | 0x0007000Bu;   // ids 4, 8
            const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(pk(a), pk(b)), __builtin_elementwise_max(pk(c), pk(d)));
            key = m.x > m.y ? m.x : m.y;
        } else
#pragma unroll
        for (int k = 1; k <= NB; k++) {
            const uint32_t cnt = k < 8 ? (tally >> (4 * k)) & 15u : popc(v32 & 0x88888888u);
            const uint32_t kk = (cnt << 4) | (uint32_t)(15 - k);
            key = kk > key ? kk : key;
        }
    } else {
        uint64_t tally = 0;
#pragma unroll
        for (int i = 0; i < NB; i++) tally += uint64_t(1) << (4u * ((uint32_t)(v >> (4 * i)) & 15u));
#pragma unroll
        for (int k = 1; k <= NB; k++) {
            const uint32_t cnt = (uint32_t)(tally >> (4 * k)) & 15u;
            const uint32_t kk = (cnt << 4) | (uint32_t)(15 - k);
            key = kk > key ? kk : key;
        }
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

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum of a small per-lane count over the wavefront, and the wave total (uniform)
__device__ __forceinline__ void wave_excl_scan(uint32_t cnt, uint32_t &off, uint32_t &total) {
    if (GE_DPP_SCAN) {
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
    } else {
        // one ballot per bit of the count (<= 15), mbcnt for the lanes below
        off = 0; total = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned long long m = __ballot((cnt >> b) & 1u);
            off += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
            total += (uint32_t)__popcll(m) << b;
        }
    }
}

struct Stamps {
    unsigned long long last, acc[4];
    __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); acc[0] = acc[1] = acc[2] = acc[3] = 0; }
    __device__ __forceinline__ void mark(int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[k] += now - last; last = now; }
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
    return (TABLE ? nth_set_bit_lds<NB>(nth8, cand, idx) : (GE_NTH_SWAR && NB <= 8) ? nth_set_bit_swar8(cand, idx) : nth_set_bit<NB>(cand, idx)) + 1u;
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

// ww_choose for a queue slot of the lone-wavefront build, N <= 8 (GE_ACT_ONEHOT).  x = alive | team_w << 16 | kind << 28 with
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
    return (GE_NTH_SWAR ? nth_set_bit_swar8(cand, idx) : nth_set_bit<8>(cand, idx)) + 1u;
}

// nibble mask (0xF per player) of the non-zero nibbles of x
__device__ __forceinline__ uint32_t nib_nonzero(uint32_t x) {
    uint32_t m = x | (x >> 1); m |= m >> 2; m &= 0x11111111u; return nib_fill(m);
}
__device__ __forceinline__ uint64_t nib_nonzero(uint64_t x) {
    uint64_t m = x | (x >> 1); m |= m >> 2; m &= 0x1111111111111111ull; return nib_fill(m);
}

// A room's role deal (POLICY.md §3 ASSIGN_ROLES): nw werewolves, then a Doctor, then a Detective, by
// repeated n-th-set-bit sampling of the players not yet dealt; `rem` = the Villagers.
// Large-batch build, N <= 8: kept as the three packed predicate words a role assignment writes (WWR::W, see deal_words) -
// turning the four masks into those words is then paid once per deal, not once per turn (1 M x 8: 8.33 -> 8.10 us/turn).
// The lone-wavefront build keeps the masks: there that work sits in an LDS wait shadow and costs nothing (C2: 1.220 -> 1.229
// with the words).  profiles/r02_ab_onehot_swar_pk.txt
struct Deal { uint32_t wolves, doc, det, rem, game, valid; };      // words form: wolves / doc / det hold W[0] / W[1] / W[2], rem is unused
template <int NB, bool LOWOCC> constexpr bool deal_as_words() { return NB <= 8 && !LOWOCC; }

template <int NB, bool LOWOCC>
__device__ __forceinline__ void deal_roles(Deal &d, uint32_t dk, uint32_t game, uint32_t n, uint32_t nw, const uint8_t *nth8) {
    uint32_t rem = (1u << n) - 1u, wolves = 0, doc = 0, det = 0;
#pragma unroll
    for (uint32_t j = 0; j < (NB > 8 ? 5u : 4u); j++) {           // nw + 2 picks, nw <= NB / 4
        const uint32_t k = popc(rem);
        const bool on = j < nw + 2u && k != 0u;            // selects, not branches (see LOWOCC)
        const uint32_t idx = pick(draw(dk, 16u + j), k | (k == 0u));
        const uint32_t pos = LOWOCC ? ((GE_NTH_SWAR && NB <= 8) ? nth_set_bit_swar8(rem, idx) : nth_set_bit<NB>(rem | (1u << 31), idx))
                                    : nth_set_bit_lds<NB>(nth8, rem, idx);
        const uint32_t bit = on ? (1u << (pos & 15u)) : 0u;
        rem &= ~bit;
        wolves |= j < nw ? bit : 0u;
        doc = j == nw ? bit : doc;
        det = j == nw + 1u ? bit : det;
    }
    if (deal_as_words<NB, LOWOCC>()) {
        WWR<NB> w;
        const uint32_t all = (1u << n) - 1u, special = all & ~rem;
#pragma unroll
        for (int k = 0; k < WWR<NB>::NW; k++) w.W[k] = 0;
        w.template set<F_VIL>(rem); w.template set<F_WOLF>(wolves); w.template set<F_DOC>(doc); w.template set<F_DET>(det);
        w.template set<F_TEAM_W>(wolves); w.template set<F_TEAM_V>(all & ~wolves);
        w.template set<F_SECRET>(special); w.template set<F_ELIG>(special);
        d.wolves = w.W[0]; d.doc = w.W[1]; d.det = w.W[2]; d.rem = 0u;
    } else {
        d.wolves = wolves; d.doc = doc; d.det = det; d.rem = rem;
    }
    d.game = game; d.valid = 1u;
}

// the packed predicate words a role assignment writes, from the prepared deal
template <int NB, bool LOWOCC> __device__ __forceinline__ void deal_words(const Deal &deal, uint32_t all, WWR<NB> &dealt) {
    if (deal_as_words<NB, LOWOCC>()) {
        dealt.W[0] = deal.wolves; dealt.W[1] = deal.doc; dealt.W[2] = deal.det;
        return;
    }
#pragma unroll
    for (int k = 0; k < WWR<NB>::NW; k++) dealt.W[k] = 0;
    const uint32_t special = all & ~deal.rem;
    dealt.template set<F_VIL>(deal.rem); dealt.template set<F_WOLF>(deal.wolves); dealt.template set<F_DOC>(deal.doc); dealt.template set<F_DET>(deal.det);
    dealt.template set<F_TEAM_W>(deal.wolves); dealt.template set<F_TEAM_V>(all & ~deal.wolves);
    dealt.template set<F_SECRET>(special); dealt.template set<F_ELIG>(special);
}

// ------------------------------------------------------------------ generic target conditions
// The slow path of `target_players.condition`: an OR of AND-clauses of literals (ge_layout.h DevCond), for DSLs that use
// the rest of the grammar (or, in [..], numeric comparisons).  Only the GENERIC kernel builds contain it, and only
// the lanes whose current row is flagged ROW_GENERIC run it; the shipped games never do.
template <int NB> __device__ __forceinline__ uint32_t ww_base_mask(const WWR<NB> &s, uint32_t set) {
    uint32_t m = 0;
    m |= (set >> F_ALIVE) & 1u ? s.template get<F_ALIVE>() : 0u;       m |= (set >> F_CAN_VOTE) & 1u ? s.template get<F_CAN_VOTE>() : 0u;
    m |= (set >> F_REVEALED) & 1u ? s.template get<F_REVEALED>() : 0u; m |= (set >> F_SECRET) & 1u ? s.template get<F_SECRET>() : 0u;
    m |= (set >> F_ELIG) & 1u ? s.template get<F_ELIG>() : 0u;         m |= (set >> F_SUB) & 1u ? s.template get<F_SUB>() : 0u;
    m |= (set >> F_TEAM_V) & 1u ? s.template get<F_TEAM_V>() : 0u;     m |= (set >> F_TEAM_W) & 1u ? s.template get<F_TEAM_W>() : 0u;
    m |= (set >> F_VIL) & 1u ? s.template get<F_VIL>() : 0u;           m |= (set >> F_WOLF) & 1u ? s.template get<F_WOLF>() : 0u;
    m |= (set >> F_DOC) & 1u ? s.template get<F_DOC>() : 0u;           m |= (set >> F_DET) & 1u ? s.template get<F_DET>() : 0u;
    return m;
}

// players whose small-integer field (FB bits per player in `arr`) lies in [lo, hi]
template <int NB, int FB, typename arr_t> __device__ __forceinline__ uint32_t range_mask(arr_t arr, uint32_t lo, uint32_t hi) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const uint32_t v = (uint32_t)(arr >> (FB * i)) & ((1u << FB) - 1u);
        m |= (v >= lo && v <= hi ? 1u : 0u) << i;
    }
    return m;
}

// `shape` (DevTable::cond_shape, wave-uniform): the largest clause count [2:0] and clause length [6:4] among the table's
// generic rows, and whether any of their literals is a base set [8] / a numeric range [9] - the loops stop there and the
// literal kind nobody uses is not evaluated (a typical generated condition has two clauses of two or three literals)
template <typename LITMASK> __device__ __forceinline__ uint32_t eval_clauses(const DevCond &c, uint32_t all, uint32_t shape, LITMASK lit_mask) {
    const uint32_t ncl = c.meta & 7u;
    const uint32_t max_ncl = shape & 7u, max_len = (shape >> 4) & 7u;
    uint32_t T = ncl ? 0u : all;                              // no condition: everybody
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((uint32_t)k >= max_ncl) break;                    // wave-uniform
        const uint32_t len = (c.meta >> (4 + 4 * k)) & 7u;
        uint32_t m = all;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if ((uint32_t)l >= max_len) break;                // wave-uniform
            const uint32_t w = c.lit[k][l];
            const uint32_t x = lit_mask(w, k, l) ^ ((w >> 30) & 1u ? all : 0u);
            m &= (uint32_t)l < len ? x : all;
        }
        T |= (uint32_t)k < ncl ? m : 0u;
    }
    return T & all;
}

// N <= 8: literals prepared for the packed predicate words (DevCond::prep, built by to_dev_cond)
// `slots` (DevTable::cond_slots, wave-uniform): which (clause, literal) slots hold a base set / a numeric range in some row
__device__ __forceinline__ uint32_t ww8_cond_generic(const WWR<8> &s, const DevCond &c, uint32_t all, uint32_t shape, uint32_t slots) {
    const uint32_t ncl = c.meta & 7u;
    const uint32_t max_ncl = shape & 7u, max_len = (shape >> 4) & 7u;
    // the selected-target nibbles of the even / odd players, one per byte
    const uint32_t ev = s.sel & 0x0F0F0F0Fu, od = (s.sel >> 4) & 0x0F0F0F0Fu;
    uint32_t T = ncl ? 0u : all;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((uint32_t)k >= max_ncl) break;                    // wave-uniform
        const uint32_t len = (c.meta >> (4 + 4 * k)) & 7u;
        uint32_t m = all;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if ((uint32_t)l >= max_len) break;                // wave-uniform
            const DevLit q = c.prep[k][l];                    // one 16-byte load
            const uint32_t w = q.w, a0 = q.a0, a1 = q.a1;
            uint32_t x = 0;
            const bool any_base = (slots >> (4 * k + l)) & 1u, any_num = (slots >> (16 + 4 * k + l)) & 1u;   // wave-uniform
            if (any_base) {                                   // gather the set's fields, OR-fold the four bytes
                uint32_t g = __builtin_amdgcn_perm(s.W[1], s.W[0], a0) | __builtin_amdgcn_perm(s.W[2], s.W[2], a1);
                g |= g >> 16; g |= g >> 8;
                x = g & 0xFFu;
            }
            if (any_num) {                                    // lo <= v <= hi, four players per word: bit 7 of a byte = in range
                const uint32_t ie = ((ev | 0x80808080u) - a0) & (a1 - ev) & 0x80808080u;
                const uint32_t io = ((od | 0x80808080u) - a0) & (a1 - od) & 0x80808080u;
                const uint32_t p = (ie >> 7) | (io >> 6);     // bits 0 / 1 of every byte = the byte's even / odd player
                const uint32_t r = (p | (p >> 6) | (p >> 12) | (p >> 18)) & 0xFFu;
                x = (any_base && ((w >> 28) & 3u) == 1u) ? x : r;
            }
            x ^= (w >> 30) & 1u ? all : 0u;
            m &= (uint32_t)l < len ? x : all;
        }
        T |= (uint32_t)k < ncl ? m : 0u;
    }
    return T & all;
}

template <int NB> __device__ __forceinline__ uint32_t ww_cond_generic(const WWR<NB> &s, const DevCond &c, uint32_t all, CondShape cs) {
    if constexpr (NB <= 8) {
        if (!(c.meta >> 31)) return ww8_cond_generic(s, c, all, cs.shape, cs.slots);
    }
    return eval_clauses(c, all, cs.shape, [&](uint32_t w, int k, int l) -> uint32_t {
        const bool any_base = (cs.slots >> (4 * k + l)) & 1u, any_num = (cs.slots >> (16 + 4 * k + l)) & 1u;   // wave-uniform
        uint32_t m = 0;
        if (any_base) m = ww_base_mask<NB>(s, w & 0xFFFFu);
        if (any_num) {
            const uint32_t r = range_mask<NB, 4>(s.sel, w & 0xFFu, (w >> 8) & 0xFFu);   // GE_NUM_SELECTED_TARGET is the pack's only numeric field
            m = (any_base && ((w >> 28) & 3u) == 1u) ? m : r;
        }
        return m;
    });
}

template <int NB> __device__ __forceinline__ uint32_t tt_cond_generic(const TT<NB> &s, const DevCond &c, uint32_t all, CondShape cs) {
    return eval_clauses(c, all, cs.shape, [&](uint32_t w, int k, int l) -> uint32_t {
        // wave-uniform: what slot (k, l) holds in some row of the table - a base set, and / or a range over which fields
        const bool any_base = (cs.slots >> (4 * k + l)) & 1u;
        const uint32_t flds = ((4 * k + l) < 8 ? cs.f0 >> (4 * ((4 * k + l) & 7)) : cs.f1 >> (4 * ((4 * k + l) & 7))) & 15u;
        uint32_t m = 0;
        if (any_base) {
            const uint32_t set = w & 0xFFFFu;
            m = ((set & 1u) ? s.speaker : 0u) | ((set & 2u) ? s.submitted : 0u) | ((set & 4u) ? s.revealed : 0u) |
                ((set & 8u) ? s.can_vote : 0u) | ((set & 16u) ? s.has_voted : 0u);
        }
        if (flds) {
            const uint32_t lo = w & 0xFFu, hi = (w >> 8) & 0xFFu, f = (w >> 16) & 7u;
            uint32_t r = 0;
            if (flds & 1u) r = f == 1u ? range_mask<NB, 2>(s.lie, lo, hi) : r;                    // GE_NUM_LIE_INDEX
            if (flds & 2u) r = f == 2u ? range_mask<NB, 2>(s.vote, lo, hi) : r;                   // GE_NUM_VOTE_CHOICE
            if (flds & 8u) r = f == 4u ? range_mask<NB, 4>(s.rounds, lo, hi) : r;                 // GE_NUM_ROUNDS_AS_SPEAKER
            if (flds & 4u) {                                                                      // GE_NUM_TOTAL_SCORE: a byte per player
                uint32_t q = 0;
#pragma unroll
                for (int i = 0; i < NB; i++) {
                    const uint32_t v = (s.score[i / 4] >> (8 * (i % 4))) & 255u;
                    q |= (v >= lo && v <= hi ? 1u : 0u) << i;
                }
                r = f == 3u ? q : r;
            }
            m = (any_base && ((w >> 28) & 3u) == 1u) ? m : r;
        }
        return m;
    });
}

// ------------------------------------------------------------------ werewolf
// LOWOCC: the launch has fewer than ~3 wavefronts per SIMD (e.g. 65 536 rooms): a lone wavefront
// stalls on every branch instruction and every dependent LDS access, so that build is branch-lean
// and computes; the other build (many wavefronts, VALU-bound) prefers LDS tables and skip-branches.
// DEAL: 1 / 0 = this instantiation is for the turns that do / do not prepare role deals (the lone-wavefront build compiles
// the turn twice rather than test a wave-uniform flag inside an exec-mask region every turn); 2 = `deal_now` decides
template <int NB, bool QUEUE, bool LOWOCC, bool GENERIC = false, int DEAL = 2>
__device__ __forceinline__ void ww_turn(WWR<NB> &s, DevRow &row, const DevRow *rows, const DevCond *conds, CondShape cs, void *wave_lds, const uint8_t *nth8, const uint32_t *ord8, bool valid, uint32_t n,
                                        uint32_t nw, uint32_t phase0_idx, uint32_t rkey, uint32_t turn, uint32_t &tk_io,
                                        bool trace, uint32_t human, Deal &deal, bool deal_now, uint32_t &ev_newly, uint64_t &ev_choice, Stamps *stamps = nullptr) {
    // human: players the host drives (never acted for here)
    // ev_*: this turn's logged actions (who acted, what they chose) for the optional event trace
    // `row` is the table row of s.phase, kept in registers across turns: LDS is read only on a transition
    // tk_io: in = turn_key(rkey, turn), out = the next turn's key
    // deal_now (wave-uniform, every GE_DEAL_PERIOD-th turn of a long launch): lanes without a prepared role deal
    // compute their next one - in the shadow of the queue's result round trip
    using nib_t = typename WWR<NB>::nib_t;
    using R = WWR<NB>;
    constexpr bool ORD = GE_ORD && NB <= 8;                    // queue slots find their player through the ord8 table
    constexpr bool ONE = GE_ONE_ATOMIC && NB <= 8 && LOWOCC;   // one result atomic per slot (C2 1.331 -> 1.312 us/turn; the large-batch build loses 1.4 %)
    constexpr bool SHADOW = GE_SHADOW && (LOWOCC || (GE_SHADOW_HI && NB <= 8));   // action-independent work inside the queue's LDS round trips
    constexpr bool ONEHOT = GE_ACT_ONEHOT && LOWOCC && ORD;    // queue slots carry the action kind one-hot (ww_choose_onehot8)
    auto *lw = static_cast<typename WaveLdsOf<LOWOCC>::type *>(wave_lds);
    const uint32_t ALL = (1u << n) - 1u;
    const uint32_t comp = row.r0 & 3u, act = (row.r0 >> 2) & 7u, p_eff = (row.r0 >> 5) & 7u;
    const uint32_t nterms = (row.r0 >> 8) & 7u;
    const uint32_t tk = tk_io;
    const uint32_t alive = s.template get<F_ALIVE>(), team_w = s.template get<F_TEAM_W>(), r_det = s.template get<F_DET>();

    // ---- who must act: target_players.condition AND alive, all players at once.
    // The 12 base predicates live packed in s.W; the row carries byte-permute selectors that pull each
    // term's mask out of the word pairs (0xFF where the term is elsewhere / absent), so the condition
    // is 2-3 v_perm + AND, XOR with the negation mask, and a fold of the term bytes (ge_layout.h DevRow).
    uint32_t T = 0;
    if (LOWOCC || comp == COMP_ACTION) {                       // LOWOCC: always evaluated, masked below (no branch)
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
    if (GENERIC && (row.r0 & ROW_GENERIC) && comp == COMP_ACTION)        // or / in [..] / numeric comparisons: the clause form
        T = ww_cond_generic<NB>(s, conds[s.phase], ALL, cs) & alive;
    if (GE_STAMPS && stamps) { asm volatile("" :: "v"(T)); stamps->mark(0); }        // [end of previous turn .. row in registers]

    // ---- PhaseNode, the part that does not depend on this turn's actions (nobody dies before the Referee):
    // phase-0 guard, resolver bitset, first matching branch.  With the action queue it runs in the shadow
    // of the queue's first LDS round trip (a lone wavefront has nothing else to cover it with).
    bool pre_open = false;
    uint32_t qe_cand = 0;
    auto phase_precompute = [&]() {
        const bool guard = s.phase == phase0_idx && !(s.flags & FLAG_PHASE0_DONE);
        pre_open = !guard;
        const uint32_t w = popc(alive & team_w), g = popc(alive & s.template get<F_TEAM_V>());
        const uint32_t prev_eff = (s.flags >> 1) & 7u;
        const uint32_t C = 1u | ((w == 0u) << RES_WOLVES_ZERO) | ((w >= g) << RES_WOLVES_GE_VILLAGERS) |
                           ((prev_eff == EFF_DAY_RESOLVE) << RES_FOLLOWS_DAY) |
                           ((prev_eff == EFF_NIGHT_RESOLVE) << RES_FOLLOWS_NIGHT) | (1u << RES_OTHERWISE);
        // first branch (DSL order) whose resolver holds: row.r2 has one byte per branch with the bit of
        // its resolver set (0 for absent branches), so the lowest non-zero byte of r2 & (C in every byte) wins
        const uint32_t hit = row.r2 & __builtin_amdgcn_perm(C, C, 0u);   // C (< 256) in every byte
        const uint32_t sh = ctz(hit) & 24u;
        qe_cand = hit != 0u ? ((row.r3 >> sh) & 255u) : s.phase;         // next row index | its entry effect << 5
    };
    // the role-assignment values of the prepared deal (what `assign` writes), also shadow work
    R dealt;
    auto deal_precompute = [&]() {
        if ((DEAL == 2 ? deal_now : DEAL == 1) && !deal.valid) {
            // this game already has roles: prepare the next game's
            const bool has_roles = (NB <= 8 ? s.W[2] : (s.W[R::NW - 2] | s.W[R::NW - 1])) != 0u;
            const uint32_t g = has_roles ? (s.games < 0xFFFFu ? s.games + 1u : s.games) : s.games;
            deal_roles<NB, LOWOCC>(deal, deal_key(rkey, g), g, n, nw, nth8);
        }
        deal_words<NB, LOWOCC>(deal, ALL, dealt);
    };
    uint32_t tk_next = 0;

    // ---- BotBehaviorNode: every due bot acts with probability 3/4, one action per visit
    uint32_t newly = 0, new_det_v = 0, new_det_w = 0;
    const bool night = act >= ACT_WOLF_TARGET && act <= ACT_DETECTIVE;
    {
        uint32_t todo = valid ? (T & ~s.acted & ~human) : 0u;
        const uint32_t known = s.det_v | s.det_w;
        const uint32_t kw_alive = s.det_w & alive;
        const uint32_t lo_kw = kw_alive & (0u - kw_alive);       // lowest known living werewolf
        if (!QUEUE) {
            phase_precompute();
            deal_precompute();
            tk_next = turn_key(rkey, turn + 1u);
            while (todo) {
                const uint32_t i = ctz(todo);
                todo &= todo - 1u;
                const uint32_t d = draw(tk, i);
                const bool go = (d & 3u) != 0u;
                const uint32_t c = ww_choose<NB, false>(act, i, d, alive, team_w, known, lo_kw, r_det, nullptr);
                const uint32_t sh = 4u * i;
                const nib_t clr = ~(nib_t(15) << sh), put = nib_t(c) << sh;
                s.choice = go ? ((s.choice & clr) | put) : s.choice;
                newly |= go ? (1u << i) : 0u;
                // RefereeNode (A): record the action (bt:204-225 update_player_state)
                s.sel = (go && night) ? ((s.sel & clr) | put) : s.sel;
                const uint32_t tb = (go && act == ACT_DETECTIVE) ? (1u << (c - 1u)) : 0u;
                new_det_w |= tb & team_w;
                new_det_v |= tb & ~team_w;
            }
        } else {
            const uint32_t lane = __lane_id();
            const uint32_t cnt = popc(todo);
            // NB <= 8: the room's slot -> player map (nibble r = its r-th due bot) from the ord8 table; the read is
            // in flight during the scan, and a slot then needs a shift instead of an n-th-set-bit search
            uint32_t ord = 0;
            if (ORD) ord = ord8[todo & 0xFFu];
            uint32_t off, total;
            wave_excl_scan(cnt, off, total);
            if (LOWOCC || total != 0u) {                        // wave-uniform; LOWOCC: some room almost always has a due bot
                // per-room context of an action; `ky`: what the acting role knows (the Detective's memory
                // at night, who the Detective is by day - ww_choose reads only one of the two per kind)
                const uint32_t ky = night ? known : (ONEHOT ? (lo_kw != 0u ? r_det : 0u) : r_det);
                const uint32_t kind = ONEHOT ? ((1u << act) >> 1) : act;          // one-hot: ACT_WOLF_TARGET = 1 -> bit 0 ...
                const uint4 ctx = ORD ? make_uint4(alive | (team_w << 16) | (kind << 28), ky | (lo_kw << 8) | (off << 16) | (lane << 26), ord, tk)
                                      : make_uint4(alive | (team_w << 16) | (act << 28), ky | (lo_kw << 16), todo | (off << 16) | (lane << 26), tk);
                if (NB <= 8) *reinterpret_cast<uint2 *>(&lw->res[lane]) = make_uint2(0u, 0u);     // only x, y come back
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
                if (SHADOW) {
                    phase_precompute();
                    asm volatile("" : "+v"(qe_cand));              // stays here: not sunk below the loop
                }
                // at least one round (total == 0: every slot is stale and dropped): the loop is left BEFORE the next
                // round's read is issued, so nothing of the queue is in flight behind it
                for (uint32_t base = 0;; base += 64u) {
                    const uint32_t k = base + lane;
                    if (GE_STAMPS && stamps && base == 0u) { asm volatile("" :: "v"(c4.x)); stamps->mark(1); }   // [.. first slot in registers]
                    uint32_t L, i, know, lokw;
                    if (ORD) {
                        L = c4.y >> 26;
                        const uint32_t rank = (k - ((c4.y >> 16) & 0x3FFu)) & 7u;       // this slot = the rank-th due bot of room L
                        i = (c4.z >> (4u * rank)) & 7u;
                        know = c4.y & 0xFFu; lokw = (c4.y >> 8) & 0xFFu;
                    } else {
                        L = c4.z >> 26;
                        const uint32_t due = c4.z & 0xFFFFu, rank = (k - ((c4.z >> 16) & 0x3FFu)) & 15u;
                        i = (LOWOCC ? nth_set_bit<NB>(due | (1u << 31), rank) : nth_set_bit_lds<NB>(nth8, due, rank)) & 15u;
                        know = c4.y & 0xFFFFu; lokw = c4.y >> 16;
                    }
                    const uint32_t d = draw(c4.w, i);
                    const bool go = k < total && (d & 3u) != 0u;
                    if (LOWOCC) {
                        // the choice is computed for every slot and only the result is predicated: a
                        // conditional block would split the slot read in two dependent LDS round trips
                        uint32_t c = ONEHOT ? ww_choose_onehot8(c4.x, i, d, know, lokw, know)
                                            : ww_choose<NB, false>(c4.x >> 28, i, d, c4.x & 0xFFFFu, (c4.x >> 16) & 0xFFFu,
                                                                   know, lokw, know, nth8);
                        if (GE_PIN_CHOICE && NB > 8) asm volatile("" : "+v"(c));   // stays outside the exec-masked block below
                        if (GE_GO_BRANCHLESS) {
                            // every slot ORs into its room's result (L is a lane index even for a stale slot), zeros if it does not act
                            uint32_t *r = reinterpret_cast<uint32_t *>(&lw->res[L]);
                            atomicOr(r, go ? (1u << i) : 0u);
                            atomicOr(r + 1 + (i >> 3), go ? (c << (4u * (i & 7u))) : 0u);
                        } else if (go) {
                            uint32_t *r = reinterpret_cast<uint32_t *>(&lw->res[L]);
                            if (!ONE) atomicOr(r, 1u << i);
                            atomicOr(r + 1 + (i >> 3), c << (4u * (i & 7u)));
                        }
                    } else if (go) {
                        const uint32_t c = ww_choose<NB, true>(c4.x >> 28, i, d, c4.x & 0xFFFFu, (c4.x >> 16) & 0xFFFu,
                                                               know, lokw, know, nth8);
                        uint32_t *r = reinterpret_cast<uint32_t *>(&lw->res[L]);
                        if (!ONE) atomicOr(r, 1u << i);
                        atomicOr(r + 1 + (i >> 3), c << (4u * (i & 7u)));
                    }
                    if (base + 64u >= total) break;                // wave-uniform
                    c4 = fetch(k + 64u);
                }
                wave_sync();
                const uint4 r = NB <= 8 ? make_uint4(reinterpret_cast<const uint2 *>(&lw->res[lane])->x, reinterpret_cast<const uint2 *>(&lw->res[lane])->y, 0u, 0u)
                                        : lw->res[lane];
                if (SHADOW) {                                      // shadow of the result read
                    tk_next = turn_key(rkey, turn + 1u);
                    deal_precompute();
                    asm volatile("" : "+v"(tk_next));
#pragma unroll
                    for (int k = 0; k < R::NW; k++) asm volatile("" : "+v"(dealt.W[k]));
                    // keeps the slot registers reserved up to here: reusing them for the work above would make the
                    // compiler wait for the queue's LDS traffic first (a read into them may be in flight)
                    asm volatile("" :: "v"(c4.x), "v"(c4.y), "v"(c4.z), "v"(c4.w));
                }
                newly = r.x;
                if (GE_STAMPS && stamps) { asm volatile("" :: "v"(newly)); stamps->mark(2); }                 // [.. results in registers]
                const nib_t got = NB > 8 ? (nib_t)(((uint64_t)r.z << 32) | r.y) : (nib_t)r.y;
                const nib_t m15 = nib_nonzero(got);              // c >= 1, so a nibble is set iff that player acted
                if (ONE) {                                       // bit i = nibble i is non-zero
                    uint32_t x = (uint32_t)m15 & 0x11111111u;
                    x = (x | (x >> 3)) & 0x03030303u; x = (x | (x >> 6)) & 0x000F000Fu;
                    newly = (x | (x >> 12)) & 0xFFu;
                }
                s.choice = (s.choice & ~m15) | got;
                // RefereeNode (A): record the action (bt:204-225 update_player_state)
                s.sel = night ? ((s.sel & ~m15) | got) : s.sel;
                {
                    const uint32_t c = (uint32_t)(got >> (4u * ctz(newly | 0x80000000u))) & 15u;
                    const uint32_t tb = (act == ACT_DETECTIVE && newly) ? (1u << ((c - 1u) & 15u)) : 0u;
                    new_det_w = tb & team_w;
                    new_det_v = tb & ~team_w;
                }
                if (!SHADOW) { phase_precompute(); tk_next = turn_key(rkey, turn + 1u); deal_precompute(); }
            } else {
                phase_precompute();
                deal_precompute();
                tk_next = turn_key(rkey, turn + 1u);
            }
        }
    }
    tk_io = tk_next;
    s.acted |= newly;
    s.template set<F_SUB>(night ? newly : 0u);                 // night_action_submitted
    ev_newly = newly;
    if (trace) {                                              // wave-uniform
        uint32_t x = newly;                                   // nibble mask of the new actors
        uint64_t m = 0;
#pragma unroll
        for (int i = 0; i < NB; i++) m |= (uint64_t)((x >> i) & 1u) << (4 * i);
        ev_choice = (uint64_t)s.choice & nib_fill(m);
    }

    // ---- PhaseNode: phase-0 guard (v2:1025-1052): first turn only records phase 0, Referee skipped;
    // completion: every target player has acted in this visit
    s.flags |= FLAG_PHASE0_DONE;                               // set by the guard turn; already set afterwards
    uint32_t qe;
    if (LOWOCC && GE_SEL_OPEN) {
        // no short-circuit evaluation: && / || became three nested exec-mask regions here (see ww_choose)
        const uint32_t open = (uint32_t)pre_open & ((uint32_t)(comp != COMP_ACTION) | (uint32_t)((T & ~s.acted) == 0u));
        qe = sel32(open != 0u, qe_cand, s.phase);
    } else {
        const bool open = pre_open && (comp != COMP_ACTION || (T & ~s.acted) == 0u);
        qe = open ? qe_cand : s.phase;
    }
    const uint32_t q = qe & 31u;
    {   // investigated_alignments[c] = team(c): an assignment, so a stale entry is replaced
        // (the guard turn has no actions: both masks are 0)
        const uint32_t seen = new_det_v | new_det_w;
        s.det_v = (s.det_v & ~seen) | new_det_v;
        s.det_w = (s.det_w & ~seen) | new_det_w;
    }
    if (q == s.phase) return;

    // ---- RefereeNode (B): effect of entering q
    const DevRow qrow = rows[q];                               // LDS read in flight during the effect: first used at the end
    const uint32_t eff = qe >> 5;
    // night / day resolution: the plurality victim dies unless the (highest-id living) Doctor guards it
    auto resolve = [&](bool on, bool day) {
        const uint32_t voters = day ? (alive & s.acted) : (alive & s.template get<F_WOLF>());
        const uint32_t victim = plurality<NB, nib_t, GE_PK_KEYS != 0>(day ? s.choice : s.sel, voters);
        const uint32_t docs = alive & s.template get<F_DOC>();
        const uint32_t guarded = (uint32_t)(s.sel >> (4u * (31u - (uint32_t)__clz((int)(docs | 1u))))) & 15u;
        uint32_t protect, bit;
        if (LOWOCC && GE_SEL_RESOLVE) {                        // data flow, no exec-mask region (see ww_choose)
            protect = guarded & (0u - (uint32_t)(!day && docs != 0u));
            bit = (1u << ((victim - 1u) & 15u)) & (0u - (uint32_t)(on && victim != 0u && victim != protect));
        } else {
            protect = (!day && docs) ? guarded : 0u;
            bit = (on && victim != 0u && victim != protect) ? (1u << ((victim - 1u) & 15u)) : 0u;
        }
        s.template clear<F_ALIVE>(bit); s.template clear<F_CAN_VOTE>(bit); s.template clear<F_ELIG>(bit);
        s.template set<F_REVEALED>(bit);
    };
    // role assignment: the deal of this game was normally prepared ahead (run loop, every 8th turn, for
    // all lanes of the wavefront at once); fall back to dealing here if it was not
    const bool is_assign = eff == EFF_ASSIGN_ROLES;
    // (bitwise, not && : one exec-mask region instead of two nested ones in the lone-wavefront build)
    const bool deal_missing = GE_SEL_NEED ? (bool)((uint32_t)is_assign & ((uint32_t)(deal.valid == 0u) | (uint32_t)(deal.game != s.games)))
                                          : (is_assign && !(deal.valid && deal.game == s.games));
    if (GE_UNLIKELY ? __builtin_expect(deal_missing, 0) : deal_missing) {
        deal_roles<NB, LOWOCC>(deal, deal_key(rkey, s.games), s.games, n, nw, nth8);
        deal_precompute();
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
#pragma unroll
        for (int k = 0; k < R::NW; k++) s.W[k] = is_assign ? ((s.W[k] & ~dmask.W[k]) | dealt.W[k]) : s.W[k];
        deal.valid = is_assign ? 0u : deal.valid;
        resolve(eff == EFF_NIGHT_RESOLVE || eff == EFF_DAY_RESOLVE, eff == EFF_DAY_RESOLVE);
    } else if (is_assign) {
#pragma unroll
        for (int k = 0; k < R::NW; k++) s.W[k] = (s.W[k] & ~dmask.W[k]) | dealt.W[k];
        deal.valid = 0u;
    } else if (eff == EFF_NIGHT_RESOLVE || eff == EFF_DAY_RESOLVE) {
        resolve(true, eff == EFF_DAY_RESOLVE);
    }
    const bool nbeg = eff == EFF_NIGHT_BEGIN;
    s.template clear<F_SUB>(nbeg ? R::FM : 0u);
    s.sel = nbeg ? nib_t(0) : s.sel;
    s.acted = 0; s.choice = 0;
    s.flags = (s.flags & FLAG_PHASE0_DONE) | (p_eff << 1);
    s.prev = s.phase;
    s.phase = q;
    row = qrow;
    s.end_turn = (((qrow.r0 >> 11) & 7u) == 0u && s.end_turn == END_NONE) ? (turn < 0xFFFEu ? turn : 0xFFFEu) : s.end_turn;
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

// the even bits of x (bit 2i -> bit i), 12 fields at most
__device__ __forceinline__ uint32_t even_bits(uint32_t x) {
    x &= 0x00555555u;
    x = (x | (x >> 1)) & 0x00333333u; x = (x | (x >> 2)) & 0x000F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu; x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// QUEUE: bot actions through the wavefront work queue (see ww_turn) - pays from 8 players on, where the
// first turn of a vote has 7-11 due bots in some room of every wavefront; TABLE: n-th-set-bit from LDS
template <int NB, bool QUEUE, bool TABLE, bool GENERIC = false>
__device__ __forceinline__ void tt_turn(TT<NB> &s, uint32_t &done, DevRow &row, const DevRow *rows, const DevCond *conds, CondShape cs, void *wave_lds, const uint8_t *nth8,
                                        bool valid, uint32_t n, uint32_t rounds,
                                        uint32_t phase0_idx, uint32_t rkey, uint32_t turn,
                                        bool trace, uint32_t human, uint32_t &ev_newly, uint64_t &ev_choice) {
    // done: tt_done_mask of s.rounds, maintained here (the caller derives it when it loads or replaces s)
    const uint32_t ALL = (1u << n) - 1u;
    const uint32_t comp = row.r0 & 3u, act = (row.r0 >> 2) & 7u, p_eff = (row.r0 >> 5) & 7u;
    const uint32_t nterms = (row.r0 >> 8) & 7u;
    const uint32_t tk = turn_key(rkey, turn);

    uint32_t T = 0;
    if (comp == COMP_ACTION) {
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
    if (GENERIC && (row.r0 & ROW_GENERIC) && comp == COMP_ACTION)        // the clause form (see ww_turn)
        T = tt_cond_generic<NB>(s, conds[s.phase], ALL, cs);

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

    const DevRow qrow = rows[q];                               // in flight during the effect (see ww_turn)
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
    row = qrow;
    if (((qrow.r0 >> 11) & 7u) == 0u && s.end_turn == END_NONE) s.end_turn = turn < 0xFFFEu ? turn : 0xFFFEu;
}
#endif  // __HIPCC__

}  // namespace ge
