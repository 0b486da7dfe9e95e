// ge_layout.h — packed room records in HBM and their in-register form.
//
// The reference keeps a room as Python dicts (AgentState, agent/game_agent_v2.py:97-117;
// per-player field dicts from games/*.yaml player_states_template).  Here a room is a
// fixed-size little-endian record of 32-bit words, FIELD-MAJOR ("bitboards"): every boolean
// player field is one N-bit mask over the players, small integers are nibble / 2-bit / byte
// arrays.  A condition such as `player.role == 'Werewolf' and player.is_alive == true`
// (ww:247) is then one AND of two masks for all players at once.
//
// HBM layout ("plane SoA"): a segment of R rooms with W words per record is stored as
// ceil(W/4) planes; plane j holds words 4j..4j+3 of every room as one dense array of
// 16-byte elements (the last plane 8-byte elements when W % 4 == 2).  A wavefront whose
// lanes are 64 consecutive rooms therefore loads/stores 1 KiB (or 512 B) contiguous per
// instruction: dwordx4 / dwordx2 per lane, fully coalesced.
//
// Record sizes = the algorithmic bytes B/2 of DESIGN.md (read once + written once per turn):
//   werewolf  N<=8 : 8 words  = 32 B      werewolf  N<=12: 10 words = 40 B
//   two-truths N<=4: 6 words  = 24 B      two-truths N<=8:  8 words = 32 B     N<=12: 12 words = 48 B
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GE_HD __host__ __device__ __forceinline__
#else
#define GE_HD inline
#endif

namespace ge {

template <int NB> struct Nib;                       // nibble-array carrier: 4 bits per player
template <> struct Nib<8> { using type = uint32_t; };
template <> struct Nib<12> { using type = uint64_t; };

enum Kind { K_WW8 = 0, K_WW12, K_TT4, K_TT8, K_TT12, K_COUNT };   // the record layouts (one kernel instantiation each)

constexpr int FLAG_PHASE0_DONE = 1;                 // flags bit 0; bits 1..3 = effect of prev phase
constexpr uint32_t END_NONE = 0xFFFFu;              // end_turn: not finished

// ---------------------------------------------------------------- werewolf pack
// masks 0..7 are, in this order, the base predicates 0..7 a DSL condition may test.
template <int NB> struct WW {
    using nib_t = typename Nib<NB>::type;
    uint32_t alive, can_vote, revealed, secret, elig, sub, team_v, team_w;   // bases 0..7
    uint32_t acted, rb0, rb1, rb2, det_v, det_w;    // rb*: bit-planes of the role index 0..4
    nib_t sel, choice;                              // selected_target_id / latest action choice
    uint32_t phase, prev, flags, end_turn;
    uint32_t games;                                 // games this slot has completed (restart mode), 16 bits
    uint32_t deal_cache;                            // N <= 8: the prepared role deal of the room's next assignment (DealPk), 0 = none
};

template <int NB> struct WWLayout;

// N <= 8: the record IS the kernels' in-register form (WWR<8> below) - words 0..2 are the three packed predicate words
// (roles one-hot per class), so a launch neither converts roles on load nor on store:
//   w0 alive | can_vote | revealed | secret        w1 elig | sub | team_v | team_w       w2 r_vil | r_wolf | r_doc | r_det
//   w3 det_v | det_w | phase | prev                w4 selected_target nibbles            w5 choice nibbles
//   w6 end_turn (16) | flags (8) | acted (8)       w7 games (16) | prepared deal (16)
// The prepared deal (ge_device.h DealPk) is a cache of a pure function of (seed, room, games): it lets un-fused launches
// (one turn each) prepare role deals ahead like a fused launch does in registers.  It is not part of the room's state:
// ge_room_view does not show it, ge_batch_write_rooms clears it, the summary checksum skips it.
template <> struct WWLayout<8> {
    static constexpr int WORDS = 8;
    static constexpr uint32_t CHECKSUM_MASK7 = 0x0000FFFFu;   // word 7 without the deal cache
    static GE_HD void unpack(const uint32_t *w, WW<8> &s) {
        s.alive = w[0] & 0xFF; s.can_vote = (w[0] >> 8) & 0xFF; s.revealed = (w[0] >> 16) & 0xFF; s.secret = w[0] >> 24;
        s.elig = w[1] & 0xFF; s.sub = (w[1] >> 8) & 0xFF; s.team_v = (w[1] >> 16) & 0xFF; s.team_w = w[1] >> 24;
        const uint32_t vil = w[2] & 0xFF, wolf = (w[2] >> 8) & 0xFF, doc = (w[2] >> 16) & 0xFF, det = w[2] >> 24;
        s.rb0 = vil | doc; s.rb1 = wolf | doc; s.rb2 = det;             // role class 1 / 2 / 3 / 4 as bit-planes
        s.det_v = w[3] & 0xFF; s.det_w = (w[3] >> 8) & 0xFF; s.phase = (w[3] >> 16) & 0xFF; s.prev = w[3] >> 24;
        s.sel = w[4]; s.choice = w[5];
        s.end_turn = w[6] & 0xFFFF; s.flags = (w[6] >> 16) & 0xFF; s.acted = w[6] >> 24;
        s.games = w[7] & 0xFFFF; s.deal_cache = w[7] >> 16;
    }
    static GE_HD void pack(const WW<8> &s, uint32_t *w) {
        w[0] = s.alive | (s.can_vote << 8) | (s.revealed << 16) | (s.secret << 24);
        w[1] = s.elig | (s.sub << 8) | (s.team_v << 16) | (s.team_w << 24);
        const uint32_t n2 = ~s.rb2 & 0xFFu;
        w[2] = (s.rb0 & ~s.rb1 & n2) | ((~s.rb0 & s.rb1 & n2) << 8) | ((s.rb0 & s.rb1 & n2) << 16) | ((s.rb2 & ~s.rb1 & ~s.rb0 & 0xFFu) << 24);
        w[3] = s.det_v | (s.det_w << 8) | (s.phase << 16) | (s.prev << 24);
        w[4] = s.sel; w[5] = s.choice;
        w[6] = s.end_turn | (s.flags << 16) | (s.acted << 24);
        w[7] = (s.games & 0xFFFF) | (s.deal_cache << 16);
    }
};

template <> struct WWLayout<12> {
    static constexpr int WORDS = 10;
    static constexpr uint32_t CHECKSUM_MASK7 = 0xFFFFFFFFu;
    // words 0..6: mask(12) | mask(12)<<12 | byte<<24 ; bytes: phase prev flags end_lo end_hi games_lo games_hi
    static GE_HD void unpack(const uint32_t *w, WW<12> &s) {
        s.alive = w[0] & 0xFFF; s.can_vote = (w[0] >> 12) & 0xFFF; s.phase = w[0] >> 24;
        s.revealed = w[1] & 0xFFF; s.secret = (w[1] >> 12) & 0xFFF; s.prev = w[1] >> 24;
        s.elig = w[2] & 0xFFF; s.sub = (w[2] >> 12) & 0xFFF; s.flags = w[2] >> 24;
        s.team_v = w[3] & 0xFFF; s.team_w = (w[3] >> 12) & 0xFFF;
        s.acted = w[4] & 0xFFF; s.rb0 = (w[4] >> 12) & 0xFFF;
        s.end_turn = (w[3] >> 24) | ((w[4] >> 24) << 8);
        s.rb1 = w[5] & 0xFFF; s.rb2 = (w[5] >> 12) & 0xFFF;
        s.det_v = w[6] & 0xFFF; s.det_w = (w[6] >> 12) & 0xFFF;
        s.games = (w[5] >> 24) | ((w[6] >> 24) << 8);
        s.sel = (uint64_t)w[7] | ((uint64_t)(w[9] & 0xFFFF) << 32);
        s.choice = (uint64_t)w[8] | ((uint64_t)(w[9] >> 16) << 32);
        s.deal_cache = 0;                               // all 320 bits of the record are state: no room for a prepared deal
    }
    static GE_HD void pack(const WW<12> &s, uint32_t *w) {
        w[0] = s.alive | (s.can_vote << 12) | (s.phase << 24);
        w[1] = s.revealed | (s.secret << 12) | (s.prev << 24);
        w[2] = s.elig | (s.sub << 12) | (s.flags << 24);
        w[3] = s.team_v | (s.team_w << 12) | ((s.end_turn & 0xFF) << 24);
        w[4] = s.acted | (s.rb0 << 12) | ((s.end_turn >> 8) << 24);
        w[5] = s.rb1 | (s.rb2 << 12) | ((s.games & 0xFF) << 24);
        w[6] = s.det_v | (s.det_w << 12) | (((s.games >> 8) & 0xFF) << 24);
        w[7] = (uint32_t)s.sel; w[8] = (uint32_t)s.choice;
        w[9] = (uint32_t)((s.sel >> 32) & 0xFFFF) | ((uint32_t)((s.choice >> 32) & 0xFFFF) << 16);
    }
};

// ---------------------------------------------------------------- werewolf room in registers (device)
// The kernels do not keep the 12 base predicates as 12 separate masks: they stay packed, 4 byte
// fields (N<=8) or 2 half-word fields (N<=12) per 32-bit word, in predicate order
//   0 alive 1 can_vote 2 revealed 3 secret | 4 elig 5 sub 6 team_v 7 team_w | 8 r_vil 9 r_wolf 10 r_doc 11 r_det
// (roles one-hot per class instead of the record's three bit-planes).  A phase's target condition is
// then a byte permute of these words by a selector precomputed in the table row (ge_device.h), and
// single fields are byte / half-word operand selects.
enum { F_ALIVE = 0, F_CAN_VOTE, F_REVEALED, F_SECRET, F_ELIG, F_SUB, F_TEAM_V, F_TEAM_W, F_VIL, F_WOLF, F_DOC, F_DET };

template <int NB> struct WWR {
    using nib_t = typename Nib<NB>::type;
    static constexpr int FPW = NB <= 8 ? 4 : 2;     // fields per word
    static constexpr int FB = 32 / FPW;             // bits per field
    static constexpr int NW = 12 / FPW;
    static constexpr uint32_t FM = (1u << FB) - 1u;
    uint32_t W[NW];
    uint32_t acted, det_v, det_w;
    nib_t sel, choice;
    uint32_t phase, prev, flags, end_turn, games;

    template <int F> GE_HD uint32_t get() const { return (W[F / FPW] >> (FB * (F % FPW))) & FM; }
    template <int F> GE_HD void set(uint32_t bits) { W[F / FPW] |= bits << (FB * (F % FPW)); }
    template <int F> GE_HD void clear(uint32_t bits) { W[F / FPW] &= ~(bits << (FB * (F % FPW))); }
    template <int F> GE_HD void put(uint32_t v) { W[F / FPW] = (W[F / FPW] & ~(FM << (FB * (F % FPW)))) | (v << (FB * (F % FPW))); }

    // flat form (the restart template travels as SegDev::init_regs and is read with scalar loads)
    static constexpr int NREGS = NW + 3 + 2 * (int)(sizeof(nib_t) / 4) + 5;
    GE_HD void to_regs(uint32_t *r) const {
        int k = 0;
        for (int j = 0; j < NW; j++) r[k++] = W[j];
        r[k++] = acted; r[k++] = det_v; r[k++] = det_w;
        r[k++] = (uint32_t)sel; if (sizeof(nib_t) > 4) r[k++] = (uint32_t)((uint64_t)sel >> 32);
        r[k++] = (uint32_t)choice; if (sizeof(nib_t) > 4) r[k++] = (uint32_t)((uint64_t)choice >> 32);
        r[k++] = phase; r[k++] = prev; r[k++] = flags; r[k++] = end_turn; r[k++] = games;
    }
    GE_HD void from_regs(const uint32_t *r) {
        int k = 0;
        for (int j = 0; j < NW; j++) W[j] = r[k++];
        acted = r[k++]; det_v = r[k++]; det_w = r[k++];
        if (sizeof(nib_t) > 4) { sel = (nib_t)((uint64_t)r[k] | ((uint64_t)r[k + 1] << 32)); k += 2; choice = (nib_t)((uint64_t)r[k] | ((uint64_t)r[k + 1] << 32)); k += 2; }
        else { sel = (nib_t)r[k++]; choice = (nib_t)r[k++]; }
        phase = r[k++]; prev = r[k++]; flags = r[k++]; end_turn = r[k++]; games = r[k++];
    }

    GE_HD void from(const WW<NB> &u) {
        for (int k = 0; k < NW; k++) W[k] = 0;
        set<F_ALIVE>(u.alive); set<F_CAN_VOTE>(u.can_vote); set<F_REVEALED>(u.revealed); set<F_SECRET>(u.secret);
        set<F_ELIG>(u.elig); set<F_SUB>(u.sub); set<F_TEAM_V>(u.team_v); set<F_TEAM_W>(u.team_w);
        const uint32_t n2 = ~u.rb2 & FM;
        set<F_VIL>(u.rb0 & ~u.rb1 & n2); set<F_WOLF>(~u.rb0 & u.rb1 & n2);
        set<F_DOC>(u.rb0 & u.rb1 & n2); set<F_DET>(u.rb2 & ~u.rb1 & ~u.rb0 & FM);
        acted = u.acted; det_v = u.det_v; det_w = u.det_w; sel = u.sel; choice = u.choice;
        phase = u.phase; prev = u.prev; flags = u.flags; end_turn = u.end_turn; games = u.games;
    }
    GE_HD void to(WW<NB> &u) const {
        u.alive = get<F_ALIVE>(); u.can_vote = get<F_CAN_VOTE>(); u.revealed = get<F_REVEALED>(); u.secret = get<F_SECRET>();
        u.elig = get<F_ELIG>(); u.sub = get<F_SUB>(); u.team_v = get<F_TEAM_V>(); u.team_w = get<F_TEAM_W>();
        const uint32_t doc = get<F_DOC>();
        u.rb0 = get<F_VIL>() | doc; u.rb1 = get<F_WOLF>() | doc; u.rb2 = get<F_DET>();
        u.acted = acted; u.det_v = det_v; u.det_w = det_w; u.sel = sel; u.choice = choice;
        u.phase = phase; u.prev = prev; u.flags = flags; u.end_turn = end_turn; u.games = games;
    }
};

// record words <-> registers.  N <= 8: the record is the register form; N <= 12: through the 12-bit field layout.
template <int NB> GE_HD void ww_load_regs(const uint32_t *w, WWR<NB> &s, uint32_t &deal_cache) {
    if (NB <= 8) {
        s.W[0] = w[0]; s.W[1] = w[1]; s.W[2] = w[2];
        s.det_v = w[3] & 0xFF; s.det_w = (w[3] >> 8) & 0xFF; s.phase = (w[3] >> 16) & 0xFF; s.prev = w[3] >> 24;
        s.sel = (typename WWR<NB>::nib_t)w[4]; s.choice = (typename WWR<NB>::nib_t)w[5];
        s.end_turn = w[6] & 0xFFFF; s.flags = (w[6] >> 16) & 0xFF; s.acted = w[6] >> 24;
        s.games = w[7] & 0xFFFF; deal_cache = w[7] >> 16;
    } else {
        WW<NB> u;
        WWLayout<NB>::unpack(w, u);
        s.from(u);
        deal_cache = 0;
    }
}
template <int NB> GE_HD void ww_store_regs(const WWR<NB> &s, uint32_t deal_cache, uint32_t *w) {
    if (NB <= 8) {
        w[0] = s.W[0]; w[1] = s.W[1]; w[2] = s.W[2];
        w[3] = s.det_v | (s.det_w << 8) | (s.phase << 16) | (s.prev << 24);
        w[4] = (uint32_t)s.sel; w[5] = (uint32_t)s.choice;
        w[6] = s.end_turn | (s.flags << 16) | (s.acted << 24);
        w[7] = s.games | (deal_cache << 16);
    } else {
        WW<NB> u;
        s.to(u);
        u.deal_cache = 0;
        WWLayout<NB>::pack(u, w);
    }
}

// ---------------------------------------------------------------- two-truths pack
// masks 0..4 are the base predicates 0..4 (is_speaker, statements_submitted, lie_revealed,
// can_vote, has_voted).  lie / vote / choice: 2 bits per player.  score: a byte per player,
// rounds_as_speaker: a nibble per player.
template <int NB> struct TT {
    uint32_t speaker, submitted, revealed, can_vote, has_voted, acted;
    uint32_t lie, vote, choice;                     // 2 bits x N
    uint32_t score[(NB + 3) / 4];                   // 4 players per word
    uint64_t rounds;                                // 4 bits x N
    uint32_t phase, prev, flags, end_turn;
    uint32_t games;

    // flat form of the restart template (SegDev::init_regs)
    static constexpr int NS = (NB + 3) / 4;
    static constexpr int NREGS = 9 + NS + 2 + 5;
    GE_HD void to_regs(uint32_t *r) const {
        int k = 0;
        r[k++] = speaker; r[k++] = submitted; r[k++] = revealed; r[k++] = can_vote; r[k++] = has_voted; r[k++] = acted;
        r[k++] = lie; r[k++] = vote; r[k++] = choice;
        for (int j = 0; j < NS; j++) r[k++] = score[j];
        r[k++] = (uint32_t)rounds; r[k++] = (uint32_t)(rounds >> 32);
        r[k++] = phase; r[k++] = prev; r[k++] = flags; r[k++] = end_turn; r[k++] = games;
    }
    GE_HD void from_regs(const uint32_t *r) {
        int k = 0;
        speaker = r[k++]; submitted = r[k++]; revealed = r[k++]; can_vote = r[k++]; has_voted = r[k++]; acted = r[k++];
        lie = r[k++]; vote = r[k++]; choice = r[k++];
        for (int j = 0; j < NS; j++) score[j] = r[k++];
        rounds = (uint64_t)r[k] | ((uint64_t)r[k + 1] << 32); k += 2;
        phase = r[k++]; prev = r[k++]; flags = r[k++]; end_turn = r[k++]; games = r[k++];
    }
};

template <int NB> struct TTLayout;

template <> struct TTLayout<4> {
    static constexpr int WORDS = 6;
    static GE_HD void unpack(const uint32_t *w, TT<4> &s) {
        s.speaker = w[0] & 0xF; s.submitted = (w[0] >> 4) & 0xF; s.revealed = (w[0] >> 8) & 0xF;
        s.can_vote = (w[0] >> 12) & 0xF; s.has_voted = (w[0] >> 16) & 0xF; s.acted = (w[0] >> 20) & 0xF;
        s.phase = w[0] >> 24;
        s.lie = w[1] & 0xFF; s.vote = (w[1] >> 8) & 0xFF; s.choice = (w[1] >> 16) & 0xFF; s.prev = w[1] >> 24;
        s.score[0] = w[2];
        s.rounds = w[3] & 0xFFFF; s.end_turn = w[3] >> 16;
        s.flags = w[4] & 0xFF; s.games = w[4] >> 16;
    }
    static GE_HD void pack(const TT<4> &s, uint32_t *w) {
        w[0] = s.speaker | (s.submitted << 4) | (s.revealed << 8) | (s.can_vote << 12) | (s.has_voted << 16) |
               (s.acted << 20) | (s.phase << 24);
        w[1] = s.lie | (s.vote << 8) | (s.choice << 16) | (s.prev << 24);
        w[2] = s.score[0];
        w[3] = (uint32_t)s.rounds | (s.end_turn << 16);
        w[4] = s.flags | ((s.games & 0xFFFF) << 16); w[5] = 0;
    }
};

template <> struct TTLayout<8> {
    static constexpr int WORDS = 8;
    static GE_HD void unpack(const uint32_t *w, TT<8> &s) {
        s.speaker = w[0] & 0xFF; s.submitted = (w[0] >> 8) & 0xFF; s.revealed = (w[0] >> 16) & 0xFF; s.can_vote = w[0] >> 24;
        s.has_voted = w[1] & 0xFF; s.acted = (w[1] >> 8) & 0xFF; s.phase = (w[1] >> 16) & 0xFF; s.prev = w[1] >> 24;
        s.lie = w[2] & 0xFFFF; s.vote = w[2] >> 16;
        s.choice = w[3] & 0xFFFF; s.end_turn = w[3] >> 16;
        s.score[0] = w[4]; s.score[1] = w[5];
        s.rounds = w[6];
        s.flags = w[7] & 0xFF; s.games = w[7] >> 16;
    }
    static GE_HD void pack(const TT<8> &s, uint32_t *w) {
        w[0] = s.speaker | (s.submitted << 8) | (s.revealed << 16) | (s.can_vote << 24);
        w[1] = s.has_voted | (s.acted << 8) | (s.phase << 16) | (s.prev << 24);
        w[2] = s.lie | (s.vote << 16);
        w[3] = s.choice | (s.end_turn << 16);
        w[4] = s.score[0]; w[5] = s.score[1];
        w[6] = (uint32_t)s.rounds;
        w[7] = s.flags | ((s.games & 0xFFFF) << 16);
    }
};

template <> struct TTLayout<12> {
    static constexpr int WORDS = 12;
    static GE_HD void unpack(const uint32_t *w, TT<12> &s) {
        s.speaker = w[0] & 0xFFF; s.submitted = (w[0] >> 12) & 0xFFF; s.phase = w[0] >> 24;
        s.revealed = w[1] & 0xFFF; s.can_vote = (w[1] >> 12) & 0xFFF; s.prev = w[1] >> 24;
        s.has_voted = w[2] & 0xFFF; s.acted = (w[2] >> 12) & 0xFFF; s.flags = w[2] >> 24;
        s.lie = w[3] & 0xFFFFFF; s.vote = w[4] & 0xFFFFFF; s.choice = w[5] & 0xFFFFFF;
        s.end_turn = (w[3] >> 24) | ((w[4] >> 24) << 8);
        s.score[0] = w[6]; s.score[1] = w[7]; s.score[2] = w[8];
        s.rounds = (uint64_t)w[9] | ((uint64_t)(w[10] & 0xFFFF) << 32);
        s.games = w[10] >> 16;
    }
    static GE_HD void pack(const TT<12> &s, uint32_t *w) {
        w[0] = s.speaker | (s.submitted << 12) | (s.phase << 24);
        w[1] = s.revealed | (s.can_vote << 12) | (s.prev << 24);
        w[2] = s.has_voted | (s.acted << 12) | (s.flags << 24);
        w[3] = s.lie | ((s.end_turn & 0xFF) << 24);
        w[4] = s.vote | ((s.end_turn >> 8) << 24);
        w[5] = s.choice;
        w[6] = s.score[0]; w[7] = s.score[1]; w[8] = s.score[2];
        w[9] = (uint32_t)s.rounds; w[10] = ((uint32_t)(s.rounds >> 32) & 0xFFFF) | ((s.games & 0xFFFF) << 16); w[11] = 0;
    }
};

// ---------------------------------------------------------------- phase table on the device
// One row = 8 words (two ds_read_b128):
//   r0: completion[1:0] act[4:2] effect[7:5] n_terms[10:8] n_br[13:11] term_neg[19:16]
//   r0 bit 20: some branch asks "all rounds done?" (two-truths)
//   r1: per term a byte {word index [7:5] (7 = no term), shift [4:0]} into the packed predicate words
//       (two-truths, and werewolf N<=12 terms 2..3)
//   r2: 4 x 8 bits, byte b = 1 << resolver of branch b (0: no such branch)
//   r3: 4 x 8 bits, byte b = target row index [4:0] | the target's entry effect [7:5]
//   r4..r6: werewolf: v_perm_b32 selectors that gather each term's mask out of the packed predicate
//       word pairs (W1:W0), (W3:W2), (W5:W4) - a term's bytes in the pair that holds its predicate,
//       0xFF bytes (selector 0x0D) in the others, so the AND of the three permutes is the term.
//       N<=8: 4 terms x 1 byte, pair (W2:W2) in r5;  N<=12: terms 0..1 x 2 bytes.
//   r7: XOR mask of the negated terms (same byte positions)
//   r0 bit 21: the target condition is generic (or / in [..] / numeric): the row's DevCond describes it, r1/r4..r7 do not;
//   r0 bits 22..26: its slot in the table's literal image (DevTable::cond_img)
struct DevRow { uint32_t r0, r1, r2, r3, r4, r5, r6, r7; };
constexpr uint32_t ROW_GENERIC = 1u << 21;

// A target condition in clause form (include/ge_step.h ge_literal), for the rows whose condition is not a plain
// conjunction of base predicates.  lit[c][l]: bits 0..15 = base-predicate bit set, or lo | hi << 8 of a numeric
// range; 16..18 = numeric field (GE_NUM_*); 28..29 = kind (1 base set, 2 numeric); 30 = negated.
// meta: n_clauses [2:0], length of clause c [4 + 4c +: 3].  Read by ge_inject_kernel (one player of one room at a time).
struct DevCond { uint32_t lit[4][4]; uint32_t meta, pad[3]; };

// The same conditions as the step kernels evaluate them (GENERIC builds): an IMAGE that a block copies into its LDS.
// Every generic row of a table is padded to the table's common shape - NCL clauses of LEN literals, the largest any of its
// rows has - with neutral literals (TRUE inside a clause the row uses, FALSE in a clause it does not have), so every lane of
// a wavefront walks the same NCL x LEN slots whatever row it is in, and no lane tests a clause count or length.  A row
// finds its literals through its slot number (DevRow r0 bits 22..26): image + slot * NCL * LEN * stride.
// One literal = 4 words (Werewolf N <= 8, Two-Truths) or 8 (Werewolf N <= 12):
//   w    bit 0: numeric range; bits 1..3: the numeric field (GE_NUM_*), bits 4..7: the same one-hot (bit 3 + field, Two-Truths);
//        bit 8: conj; neither: a base set; bits 16..31: 0xFFFF if negated (base set / numeric range)
//   conj        an AND of up to 4 (Werewolf N <= 8) / 2 single base predicates, each possibly negated - what most of a generated
//               condition consists of: m[0..] = v_perm_b32 selectors over the predicate word pairs + the XOR mask of the negated
//               terms, exactly a shipped row's own terms (DevRow r4..r7): {sA, sB, xor} / Werewolf N <= 12 {sA, sB, sC, xor}
//   base set    m[k] = AND-mask over packed predicate word k (0xFF / 0xFFFF in the fields the set names): the literal is
//               the OR-fold of (W[k] & m[k]) - any number of fields, three AND / OR per word (Two-Truths packs its five
//               masks as speaker | submitted << 16, revealed | can_vote << 16, has_voted)
//   numeric     m[0], m[1] = the bounds prepared for the field's SWAR compare (ge_device.h range_*): lo in every byte and
//               hi | 0x80 in every byte (nibble arrays), the same in half-words (scores), or the allowed-value masks of a
//               2-bit field
// An empty range and the neutral literals are base sets with all-zero masks (FALSE), negated for TRUE.
struct CondLit { uint32_t w, m[3]; };
struct CondLit12 { uint32_t w, m[6], pad; };
constexpr uint32_t COND_IMG_BYTES = 16384;      // 32 rows x 16 literals x 32 B at the very most
constexpr uint32_t ROW_COND_SLOT_SHIFT = 22;    // DevRow r0 bits 22..26: the generic row's slot in the image

// the generic rows' common shape, read once per launch into scalar registers (wave-uniform loop bounds and skips).
// shape: NCL [2:0], LEN [6:4].  g / f: a nibble per slot in the order the evaluator visits them (slot i = clause i / LEN,
// literal i % LEN; the evaluator shifts them out) - g: bit 0 some row has a base set in this slot, bit 1 a numeric range,
// bit 2 the slot ends a clause, bit 3 some row has a conj; f: which numeric fields the slot compares in some row
// (bit = GE_NUM_* - 1, Two-Truths)
struct CondShape { uint32_t shape, g_lo, g_hi, f_lo, f_hi; };

// rows | ord8 | nth8 | spread8 | tally64 lead the structure, contiguous and in the order of a step block's LDS: one linear
// copy of the first IMG_* bytes fills it (the lone-wavefront builds take rows (+ ord8), Two-Truths up to nth8)
constexpr uint32_t IMG_ROWS = 0, IMG_ORD8 = 1024, IMG_NTH8 = 2048, IMG_SPREAD8 = 4096, IMG_TALLY = 5120, IMG_END = 7168;
struct DevTable {
    DevRow rows[32];
    uint32_t ord8[256];      // ord8[mask] = the positions of the set bits of an 8-bit mask, ascending, one nibble each
    uint8_t nth8[2048];      // n-th-set-bit table (ge_device.h), copied to LDS by the large-batch build
    uint32_t spread8[256];   // spread8[mask] = 0xF in nibble i for every set bit i of the 8-bit mask (voters -> vote nibbles)
    uint64_t tally64[256];   // tally64[a | b << 4] = (1 << 4a) + (1 << 4b): two votes' worth of nibble counters per lookup
    int32_t n_phases, rounds, n_players;
    uint32_t cond_shape;     // generic rows: largest clause count [2:0] and clause length [6:4] (CondShape::shape)
    uint32_t cond_g[2];      // generic rows: CondShape::g
    uint32_t cond_fields[2]; // generic rows: CondShape::f
    uint32_t cond_n16;       // generic rows: 16-byte elements of cond_img in use (what a block copies into its LDS)
    uint32_t pad_[2];
    DevCond conds[32];       // clause form of the generic rows, per row (ge_inject_kernel)
    alignas(16) unsigned char cond_img[COND_IMG_BYTES];   // the same for the step kernels: padded literals, slot-major (CondLit / CondLit12)
};

// plane geometry of a segment
GE_HD int planes_of(int words) { return (words + 3) / 4; }
GE_HD int plane_words(int words, int j) { return (words - 4 * j) >= 4 ? 4 : (words - 4 * j); }
// byte offset of plane j inside a segment of `rooms` rooms (rooms padded by the caller to 64)
GE_HD uint64_t plane_offset(uint64_t rooms_padded, int j) { return (uint64_t)j * 16u * rooms_padded; }

}  // namespace ge
