// ge_step.hip — host side and C ABI of libge_step.so (gfx950 only); the kernels are ge_kernels.inl.
//
// Replaces, for a batch of rooms, the reference's per-room turn loop
// (agent/game_agent_v2.py:1571-1587 graph; nodes :198/:468/:987/:619) and its dict plumbing
// (agent/tools/backend_tools.py:204-225, 285-344).  See include/ge_step.h for the boundary,
// ge_layout.h for the HBM layout, ge_device.h for the turn itself.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ge_step.h"
#include "ge_device.h"
#include "ge_layout.h"
#include "ge_host.h"

using namespace ge;

namespace {

#include "ge_kernels.inl"          // all device code: step / summary / fill / injection kernels

thread_local int g_last_hip = 0;
thread_local uint64_t g_last_rejected_room = ~0ull;   // ge_batch_write_rooms: the first view (index in the batch) that did not fit its segment

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) { g_last_hip = (int)e__; return (int)GE_ERR_HIP; } \
    } while (0)

struct Segment {
    ge_game_table table;
    SegDev dev;
    uint64_t local_first;      // index of the segment's room 0 inside the batch
};

}  // namespace

struct ge_batch {
    int device = 0;
    uint64_t seed = 0, first_room = 0, turn = 0, n_rooms = 0;
    uint32_t max_fuse = 64, block_threads = 256, n_blocks = 0, flags = 0;
    std::vector<Segment> segs;
    void *state = nullptr;            // one allocation, segments back to back
    void *trace = nullptr;            // GE_FLAG_TRACE: per segment [max_fuse][rooms_padded] x 16 B
    void *deal_side = nullptr;        // Werewolf x 12 segments: prepared role deals of the single-turn launches, [rooms_padded] x 8 B each (ge_kernels.inl run_ww); allocated by the first single-turn launch (ensure_deal_side)
    bool deal_side_failed = false;    // ... or never (no Werewolf x 12 segment / no memory for a cache)
    uint32_t last_step_turns = 0;     // turns of the most recent ge_batch_step (what the trace holds)
    size_t state_bytes = 0;
    bool stream_loads = false;  // a single-game batch's large single-turn launches load the record with streaming loads (create_impl)
    DevTable *tables = nullptr;
    SegDev *segs_dev = nullptr;
    unsigned long long *sum_dev = nullptr;
    hipStream_t last_stream = nullptr;
    bool generic = false;             // some phase has a generic target condition: the GENERIC kernel builds are launched
    uint32_t gshape = 0;              // ... a single Two-Truths table whose conditions all fit 1 x 1 / 1 x 2 literal slots: 2 / 3 = the shape-specialised fused builds
    uint32_t cond_bytes = 0;          // ... and their blocks keep this much of literal image in LDS (the largest table's)
    bool pending = false;             // work was queued on last_stream since the last synchronisation
    hipEvent_t order_ev = nullptr;    // orders a step on a new stream behind the previous stream's work
    unsigned long long *stamps_dev = nullptr;   // GE_STAMPS diagnostic build only
    void *inj_buf = nullptr;          // device scratch of ge_batch_inject_actions
    size_t inj_cap = 0;
    void *io_buf = nullptr;           // pinned staging of ge_batch_read_rooms / write_rooms (packed planes of the range)
    size_t io_cap = 0;
    // hipGraph replay of launch-bound step sequences (many short launches per ge_batch_step call)
    uint32_t *turn_dev = nullptr;     // turn base the captured launches read
    uint64_t turn_dev_value = ~0ull;  // what *turn_dev holds (host mirror)
    hipStream_t cap_stream = nullptr; // capture needs a non-default stream
    std::vector<std::pair<uint32_t, hipGraphExec_t>> graphs;   // per n_turns
    bool graphs_ok = true;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double timed_ms = 0.0;
    uint64_t launches = 0;
};

static int fill_args(const ge_batch *b, StepArgs &a, uint32_t turn0, uint32_t n_turns) {
    memset(&a, 0, sizeof a);
    a.n_seg = (uint32_t)b->segs.size();
    for (uint32_t k = 0; k < a.n_seg; k++) a.block_begin[k] = b->segs[k].dev.block_begin;
    a.turn0 = turn0; a.n_turns = n_turns;
    a.seed_key = seed_key((uint32_t)b->seed, (uint32_t)(b->seed >> 32));
    a.block_threads = b->block_threads;
    a.restart = (b->flags & GE_FLAG_RESTART) ? 1u : 0u;
    a.trace = (b->flags & GE_FLAG_TRACE) ? 1u : 0u;
    // up to one wavefront per SIMD (1 024 SIMDs x 64 rooms) -> the branch-lean lone-wavefront build (ge_device.h LOWOCC):
    // 65 536 rooms 1.283 vs 1.372 us/turn; from the first SIMD with two wavefronts on the large-batch build is as fast or
    // faster (69 632 rooms 1.680 vs 1.668, 131 072: 1.791 vs 1.738; tools/threshold_probe.sh, profiles/r02_threshold.txt).
    // GE_LOWOCC_ROOMS overrides the threshold (tuning / A-B runs)
    static const uint64_t low_rooms = [] {
        const char *e = getenv("GE_LOWOCC_ROOMS");
        return e ? strtoull(e, nullptr, 10) : (uint64_t)(1024u * 64u + 1u);
    }();
    a.lowocc = b->n_rooms < low_rooms ? 1u : 0u;
    a.stamps = b->stamps_dev;
    return GE_OK;
}

// every entry point leaves the caller's current device as it found it
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { hipError_t e = hipSetDevice(dev); if (e != hipSuccess) { g_last_hip = (int)e; ok = false; } } else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define GE_ON_DEVICE(b) DeviceGuard dg__((b)->device); if (!dg__.ok) return (int)GE_ERR_HIP

// nothing may throw across the C ABI (std::vector / new inside the entry points)
template <class F> static int guarded(F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return GE_ERR_NOMEM; }
    catch (...) { return GE_ERR_ARG; }
}

// a step on a stream other than the previous one is ordered behind it
static int order_after_previous(ge_batch *b, hipStream_t st) {
    if (b->pending && st != b->last_stream) {
        if (!b->order_ev) HIP_TRY(hipEventCreateWithFlags(&b->order_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->order_ev, b->last_stream));
        HIP_TRY(hipStreamWaitEvent(st, b->order_ev, 0));
    }
    b->last_stream = st;
    b->pending = true;
    return GE_OK;
}

extern "C" {

int ge_abi_version(void) { return GE_ABI_VERSION; }
int ge_last_hip_error(void) { return g_last_hip; }
uint64_t ge_last_rejected_room(void) { return g_last_rejected_room; }

int ge_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ge_strerror(int status) {
    switch (status) {
    case GE_OK: return "ok";
    case GE_ERR_ARG: return "invalid argument";
    case GE_ERR_DSL: return "game DSL cannot be compiled";
    case GE_ERR_NO_DEVICE: return "no HIP device (the stepper has no CPU path)";
    case GE_ERR_HIP: return "HIP runtime error";
    case GE_ERR_NOMEM: return "out of memory";
    case GE_ERR_RANGE: return "out of range";
    case GE_ERR_UNSUPPORTED: return "unsupported";
    case GE_ERR_COMM: return "collective library (RCCL) error";
    default: return "unknown status";
    }
}

}  // extern "C"

// seg_first (may be null): global index of each segment's room 0.  Null = the segments follow each other from
// desc->first_room on; a device group (ge_group.inl) passes the indices its shard has in the whole job.
static int create_impl(const ge_batch_desc *desc, ge_batch **out, const uint64_t *seg_first = nullptr) {
    if (!desc || !out || desc->n_segments == 0 || desc->n_segments > GE_MAX_SEGMENTS) return GE_ERR_ARG;
    *out = nullptr;
    if (ge_device_count() <= 0) return GE_ERR_NO_DEVICE;
    if (desc->device < 0 || desc->device >= ge_device_count()) return GE_ERR_ARG;
    ge_batch *b = new (std::nothrow) ge_batch();
    if (!b) return GE_ERR_NOMEM;
    b->device = desc->device; b->seed = desc->seed; b->first_room = desc->first_room;
    b->max_fuse = desc->max_fuse ? desc->max_fuse : 64u;
    b->flags = desc->flags;
    uint64_t total = 0;
    for (uint32_t k = 0; k < desc->n_segments; k++) total += desc->seg[k].n_rooms;
    // Rooms per block.  Up to four wavefronts per SIMD (262 144 rooms) 64-room blocks are as fast as any and balance best; beyond, a block's
    // LDS (the 7 KB table image + its wavefronts' queues) caps a CU at 5 wavefronts per SIMD when every wavefront brings its own image, and
    // the launch takes a second round: fused us per turn at 393 216 / 524 000 rooms, 64- against 256-room blocks: Werewolf x 8 3.50 / 4.66 ->
    // 2.54 / 3.52, Two-Truths x 4 at 524 000 3.31 -> 2.35, Werewolf x 12 6.46 -> 4.87 (-24 .. -28 %; level at 262 144 and below;
    // profiles/r05_fused_midsize.txt).  (Until round 5 the switch was at 524 288 rooms.)
    b->block_threads = total > (1u << 18) ? 256u : 64u;
    // a batch that only ever runs single-turn launches (max_fuse = 1) on the large-batch kernels: 256-room blocks at once (launch_geometry does
    // the same per launch for a single-game batch; a mixed batch's block size is fixed here, its segments are padded to it)
    if (b->max_fuse == 1u && total > 65536u) b->block_threads = 256u;
    if (const char *e = getenv("GE_BLOCK_THREADS")) {          // tuning / A-B runs: 64, 128 or 256
        const unsigned long v = strtoul(e, nullptr, 10);
        if (v == 64 || v == 128 || v == 256) b->block_threads = (uint32_t)v;
    }
    uint64_t local = 0, global = desc->first_room;
    size_t bytes = 0;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < desc->n_segments; k++) {
        const ge_segment_desc &sd = desc->seg[k];
        if (!sd.table || sd.n_rooms == 0 || sd.table->abi_version != GE_ABI_VERSION ||
            sd.table->n_phases <= 0 || sd.table->n_phases > GE_MAX_PHASES) { delete b; return GE_ERR_ARG; }
        Segment s;
        s.table = *sd.table;
        const uint32_t n = sd.n_players;
        if (n > GE_MAX_PLAYERS || (int)n < s.table.min_players) { delete b; return GE_ERR_ARG; }
        SegDev &d = s.dev;
        memset(&d, 0, sizeof d);
        if (s.table.pack == GE_PACK_WEREWOLF) {
            if (n < 4) { delete b; return GE_ERR_ARG; }
            d.kind = n <= 8 ? K_WW8 : K_WW12;
        } else if (s.table.pack == GE_PACK_TWO_TRUTHS) {
            if (n < 2 || s.table.rounds < 1 || s.table.rounds > 15 ||
                s.table.rounds * 2 * ((int)n - 1) > 255) { delete b; return GE_ERR_ARG; }
            d.kind = n <= 4 ? K_TT4 : (n <= 8 ? K_TT8 : K_TT12);
        } else { delete b; return GE_ERR_ARG; }
        d.words = (uint32_t)words_of(d.kind);
        d.rooms = sd.n_rooms;
        d.rooms_padded = (sd.n_rooms + 255u) & ~uint64_t(255);
        d.first_global = seg_first ? seg_first[k] : global;
        d.local_first = local;
        d.n_players = n; d.nw = n / 4 > 1 ? n / 4 : 1; d.rounds = (uint32_t)s.table.rounds;
        d.human_mask = sd.human_mask & ((1u << n) - 1u);
        d.phase0_idx = 255;
        for (int r = 0; r < s.table.n_phases; r++)
            if (s.table.rows[r].phase_id == 0) d.phase0_idx = (uint32_t)r;
        if (d.phase0_idx == 255) { delete b; return GE_ERR_ARG; }   // AgentState starts at phase id 0 (v2:103)
        d.block_begin = blocks; d.table_idx = k;
        blocks += (uint32_t)((sd.n_rooms + b->block_threads - 1) / b->block_threads);
        s.local_first = local;
        d.base = reinterpret_cast<uint32_t *>(bytes);     // offset for now, rebased after hipMalloc
        bytes += (size_t)planes_of((int)d.words) * 16u * d.rooms_padded;
        local += sd.n_rooms; global += sd.n_rooms;
        b->segs.push_back(s);
    }
    b->n_rooms = local; b->n_blocks = blocks; b->state_bytes = bytes;
    // record_loads - the record loads of a single-game batch's large single-turn launches, plain or streaming (non-temporal), by layout and by where the state
    // lives (ge_kernels.inl load_words; profiles/r05_ab_ww8_nt_loads.txt, r05_ab_nt_others.txt): beyond the 256 MiB Infinity Cache of MI355X
    // every layout streams; while the cache holds the state plain loads are served from it (+4 ... +18 % at 100 - 250 MiB), except that the
    // smallest states measured - Two-Truths at 24 MiB, Werewolf x 12 at 80 MiB - are level or 2 % better streaming.  GE_NT_LOADS=0 / 1 forces
    // plain / streaming (A/B, knob runs)
    {
        static const int nt_env = [] { const char *e = getenv("GE_NT_LOADS"); return e ? atoi(e) : -1; }();
        const uint32_t kind0 = b->segs[0].dev.kind;
        const size_t mib = (size_t)1 << 20, small = kind0 == K_WW8 ? 0 : kind0 == K_WW12 ? 96 * mib : 48 * mib;
        size_t touched = 0;                                     // what a launch reads: the records' own words (a layout's last plane is allocated 16 bytes wide)
        for (const Segment &sg : b->segs) touched += (size_t)sg.dev.words * 4u * sg.dev.rooms_padded;
        b->stream_loads = nt_env >= 0 ? nt_env != 0 : (touched > 288 * mib || touched <= small);
    }
    int st = GE_OK;
    {
        DeviceGuard dg(b->device);
        do {
            if (!dg.ok) { st = GE_ERR_HIP; break; }
            hipError_t e = hipMalloc(&b->state, bytes);
            if (e != hipSuccess) { g_last_hip = (int)e; st = e == hipErrorOutOfMemory ? GE_ERR_NOMEM : GE_ERR_HIP; break; }
            if (hipMalloc(reinterpret_cast<void **>(&b->tables), sizeof(DevTable) * b->segs.size()) != hipSuccess ||
                hipMalloc(reinterpret_cast<void **>(&b->segs_dev), sizeof(SegDev) * GE_MAX_SEGMENTS) != hipSuccess ||
                hipMalloc(reinterpret_cast<void **>(&b->sum_dev), sizeof(unsigned long long) * 64) != hipSuccess) { st = GE_ERR_HIP; break; }
            if (b->flags & GE_FLAG_TRACE) {
                size_t tb = 0;
                for (Segment &s : b->segs) tb += (size_t)b->max_fuse * s.dev.rooms_padded * 16u;
                if (hipMalloc(&b->trace, tb) != hipSuccess) { st = GE_ERR_NOMEM; break; }
                size_t off = 0;
                for (Segment &s : b->segs) {
                    s.dev.trace = reinterpret_cast<uint32_t *>(static_cast<char *>(b->trace) + off);
                    off += (size_t)b->max_fuse * s.dev.rooms_padded * 16u;
                }
            }
            std::vector<DevTable> host_tables(b->segs.size());
            for (size_t k = 0; k < b->segs.size(); k++) {
                Segment &s = b->segs[k];
                s.dev.base = reinterpret_cast<uint32_t *>(static_cast<char *>(b->state) + reinterpret_cast<size_t>(s.dev.base));
                DevTable &dt = host_tables[k];
                memset(&dt, 0, sizeof dt);
                for (int r = 0; r < s.table.n_phases; r++) {
                    dt.rows[r] = to_dev_row(s.table, s.table.rows[r], s.dev.kind);
                    dt.conds[r] = to_dev_cond(s.table.rows[r]);
                    if (s.table.rows[r].generic) b->generic = true;
                }
                build_cond_image(s.table, s.dev.kind, dt);
                b->cond_bytes = std::max<uint32_t>(b->cond_bytes, dt.cond_n16 * 16u);
                if (b->segs.size() == 1 && s.table.pack == GE_PACK_TWO_TRUTHS && dt.cond_n16) {
                    static const bool no_shapes = getenv("GE_NO_GENERIC_SHAPES") != nullptr;       // A/B runs: the rolled walk for every table
                    const uint32_t ncl = dt.cond_shape & 7u, len = (dt.cond_shape >> 4) & 7u;
                    b->gshape = no_shapes ? 0u : (ncl == 1u && len == 1u) ? 2u : (ncl == 1u && len == 2u) ? 3u : 0u;
                }
                dt.n_phases = s.table.n_phases; dt.rounds = s.table.rounds; dt.n_players = (int32_t)s.dev.n_players;
                fill_nth8_host(dt.nth8);
                fill_ord8_host(dt.ord8);
                fill_vote_luts_host(dt.spread8, dt.tally64);
                // the initial record: player_states_template for every player, phase id 0 (utils.py:642-647)
                ge_room_view v;
                memset(&v, 0, sizeof v);
                v.phase_id = 0; v.prev_phase_id = 0; v.end_turn = -1; v.n_players = (uint8_t)s.dev.n_players;
                v.pack = (uint8_t)s.table.pack;
                for (uint32_t i = 0; i < s.dev.n_players; i++) memcpy(v.players[i], s.table.init_fields, 12);
                uint32_t w[12] = {0};
                view_to_words(s.dev.kind, v, s.table, w);
                memcpy(s.dev.init_words, w, sizeof w);
                init_regs_of(s.dev.kind, w, s.dev.init_regs);
                s.dev.done0 = 0;                                          // tt_done_mask of the initial record: rounds_as_speaker >= agreed rounds
                if (s.table.pack == GE_PACK_TWO_TRUTHS)
                    for (uint32_t i = 0; i < s.dev.n_players; i++)
                        if ((int)(v.players[i][8] & 15) >= s.table.rounds) s.dev.done0 |= 1u << i;
                for (int r = 0; r < s.table.n_phases; r++)
                    if (s.table.rows[r].n_branches == 0) s.dev.term_mask |= 1u << r;
            }
            if (hipMemcpy(b->tables, host_tables.data(), sizeof(DevTable) * host_tables.size(), hipMemcpyHostToDevice) != hipSuccess) { st = GE_ERR_HIP; break; }
            SegDev host[GE_MAX_SEGMENTS];
            memset(host, 0, sizeof host);
            for (size_t k = 0; k < b->segs.size(); k++) host[k] = b->segs[k].dev;
            if (hipMemcpy(b->segs_dev, host, sizeof host, hipMemcpyHostToDevice) != hipSuccess) { st = GE_ERR_HIP; break; }
        } while (0);
    }
    if (st != GE_OK) { ge_batch_destroy(b); return st; }
#if GE_STAMPS
    if (getenv("GE_STAMPS_OUT")) {
        DeviceGuard dg(b->device);
        const size_t sb = 64 + (GE_STAMPS == 2 ? 32 * (size_t)((b->n_rooms + 63) / 64 + 64) : 0);    // GE_STAMPS = 2: + a {start, end, where, cycles} record per wavefront
        if (hipMalloc(reinterpret_cast<void **>(&b->stamps_dev), sb) == hipSuccess) (void)hipMemset(b->stamps_dev, 0, sb);
    }
#endif
    st = ge_batch_reset(b);
    if (st != GE_OK) { ge_batch_destroy(b); return st; }
    *out = b;
    return GE_OK;
}

static int sync_impl(ge_batch *b) {
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    b->pending = false;
    return GE_OK;
}

static int reset_impl(ge_batch *b) {
    GE_ON_DEVICE(b);
    int st = sync_impl(b);
    if (st != GE_OK) return st;
    b->turn = 0;
    // device-side fill: nothing the size of the state crosses PCIe
    uint64_t most = 0;
    for (const Segment &s : b->segs) most = s.dev.rooms_padded > most ? s.dev.rooms_padded : most;
    const uint32_t blocks = (uint32_t)((most / 256u) < 2048u ? (most / 256u) : 2048u);
    hipLaunchKernelGGL(ge_fill_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, b->last_stream, b->segs_dev, (uint32_t)b->segs.size());
    HIP_TRY(hipGetLastError());
    return sync_impl(b);
}

// ---- one launch of the step kernel.  What is launched is chosen from four facts, each a template argument of the kernels so that
// every build gets its own register allocation (ge_kernels.inl): the record layout (or "mixed": several segments), LOWOCC (up to
// one wavefront per SIMD), GENERIC (some table has a generic target condition) and SINGLE (the launch is exactly one turn).
namespace {
struct LaunchShape { dim3 grid, block; hipStream_t st; const ge_batch *b; StepArgs a; };

template <class K> inline void launch_one(K kernel, const LaunchShape &L, bool queue, bool lds_low) {
    StepArgs a = L.a;
    const uint32_t base = step_lds_bytes(queue, lds_low, L.block.x);
    a.cond_off = base;                                        // GENERIC builds: the literal image follows everything else in LDS
    hipLaunchKernelGGL(kernel, L.grid, L.block, base + (L.b->generic ? L.b->cond_bytes : 0u), L.st, L.b->segs_dev, L.b->tables, a);
}

template <bool LOW, int GEN, bool SINGLE> inline void launch_kind(uint32_t kind, const LaunchShape &L) {
    constexpr bool HAS_ALT = SINGLE && !LOW && GEN == 0;
    const bool alt = HAS_ALT && L.b->stream_loads != (kind != K_WW8);      // the layouts' default: Werewolf x 8 plain, the others streaming
    switch (kind) {
    // (the large-batch single-turn kernels of the shipped games exist with plain and with streaming record loads: ALT = the form that is not the
    // layout's default, launched when the batch's stream_loads (create_impl: record_loads) differs from that default)
    case K_WW8:
        if (alt) launch_one(ge_step_kernel<K_WW8, LOW, GEN, SINGLE, HAS_ALT ? 2 : 0>, L, true, LOW);
        else launch_one(ge_step_kernel<K_WW8, LOW, GEN, SINGLE>, L, true, LOW);
        break;
    case K_WW12:
        if (alt) launch_one(ge_step_kernel<K_WW12, LOW, GEN, SINGLE, HAS_ALT ? 1 : 0>, L, true, LOW);
        else launch_one(ge_step_kernel<K_WW12, LOW, GEN, SINGLE>, L, true, LOW);
        break;
    case K_TT4:
        if (alt) launch_one(ge_step_kernel<K_TT4, LOW, GEN, SINGLE, HAS_ALT ? 1 : 0>, L, tt_uses_queue(4, LOW), LOW);
        else launch_one(ge_step_kernel<K_TT4, LOW, GEN, SINGLE>, L, tt_uses_queue(4, LOW), LOW);
        break;
    case K_TT8:
        if (alt) launch_one(ge_step_kernel<K_TT8, LOW, GEN, SINGLE, HAS_ALT ? 1 : 0>, L, tt_uses_queue(8, LOW), LOW);
        else launch_one(ge_step_kernel<K_TT8, LOW, GEN, SINGLE>, L, tt_uses_queue(8, LOW), LOW);
        break;
    default:
        if (alt) launch_one(ge_step_kernel<K_TT12, LOW, GEN, SINGLE, HAS_ALT ? 1 : 0>, L, tt_uses_queue(12, LOW), LOW);
        else launch_one(ge_step_kernel<K_TT12, LOW, GEN, SINGLE>, L, tt_uses_queue(12, LOW), LOW);
        break;
    }
}
template <bool LOW, int GEN> inline void launch_kind(uint32_t kind, bool single, const LaunchShape &L) {
    if (single) launch_kind<LOW, GEN, true>(kind, L); else launch_kind<LOW, GEN, false>(kind, L);
}
// Two-Truths fused launches on a table whose generic conditions all have one of the common shapes: the shape-specialised builds
template <bool LOW, int GEN> inline void launch_tt_shaped(uint32_t kind, const LaunchShape &L) {
    switch (kind) {
    case K_TT4: launch_one(ge_step_kernel<K_TT4, LOW, GEN, false>, L, tt_uses_queue(4, LOW), LOW); break;
    case K_TT8: launch_one(ge_step_kernel<K_TT8, LOW, GEN, false>, L, tt_uses_queue(8, LOW), LOW); break;
    default: launch_one(ge_step_kernel<K_TT12, LOW, GEN, false>, L, tt_uses_queue(12, LOW), LOW); break;
    }
}
template <bool LOW> inline void launch_tt_shaped(uint32_t gshape, uint32_t kind, const LaunchShape &L) {
    if (gshape == 2u) launch_tt_shaped<LOW, 2>(kind, L); else launch_tt_shaped<LOW, 3>(kind, L);
}
template <bool LOW, int GEN> inline void launch_mixed(bool single, const LaunchShape &L) {
    if (single) launch_one(ge_step_kernel_mixed<LOW, GEN, true>, L, true, LOW); else launch_one(ge_step_kernel_mixed<LOW, GEN, false>, L, true, LOW);
}
}  // namespace

// rooms per block and number of blocks of a launch of n_turns turns
static void launch_geometry(const ge_batch *b, bool low, bool single, uint32_t &bt, uint32_t &blocks, uint32_t &rpb) {
    bt = b->block_threads; blocks = b->n_blocks;
    rpb = bt;                                                   // rooms per block
    // One turn per launch (max_fuse = 1, or the tail of a step): a block's first act is to fill its 7 KB of LDS tables behind a
    // barrier - the fewer blocks, the less of that per launch - so a large single-game Werewolf x 8 batch runs these launches in
    // blocks of 512 rooms (GE_SINGLE_BLOCK = 256 / 512 / 1024 overrides, for A/B runs); a single game's rooms start at block 0,
    // so the grid is just the rooms over the block size
    // (profiles/r04_ab_single_block.txt, sustained us per launch at 256 / 512 / 1 024 rooms per block: Werewolf x 8 14.23 / 14.01 /
    // 14.22 at 1 M rooms and 386 / 373 / 377 at 33 M; Werewolf x 12 34.4 / 34.6 / 37.9 at 2 M; Two-Truths x 4 10.2 / 10.2 / 10.4)
    static const uint32_t single_block_env = [] {
        const char *e = getenv("GE_SINGLE_BLOCK");
        const unsigned long v = e ? strtoul(e, nullptr, 10) : 0ul;
        return (v == 256 || v == 512 || v == 1024) ? (uint32_t)v : 0u;
    }();
    if (single && b->segs.size() == 1 && !b->generic && !low && bt == 256u) {
        // (not below 393 216 rooms: 300 000 rooms are 586 blocks of 512 - 2.3 per CU - and take 6.69 us against 6.10 in 256-room blocks)
        bt = single_block_env ? single_block_env : (b->segs[0].dev.kind == K_WW8 && b->segs[0].dev.rooms >= 393216u ? 512u : 256u);
        blocks = (uint32_t)((b->segs[0].dev.rooms + bt - 1u) / bt);
        rpb = bt;
    }
    // A single-game batch of up to 262 144 rooms is cut into 64-room blocks (create_impl: more, shorter blocks balance a fused launch
    // better).  Its single-turn launches on the large-batch kernels run 256-room blocks all the same - a quarter of the table fills:
    // sustained us per launch, 64- against 256-room blocks (profiles/r05_k1_midsize.txt): Werewolf x 8 at 131 072 / 262 144 rooms
    // 4.96 / 6.35 -> 4.21 / 5.40, Two-Truths x 4 at 262 144 5.14 -> 4.58, Werewolf x 12 at 131 072 / 262 144 5.80 / 7.53 -> 5.10 / 6.90
    // (-8 .. -15 %; -29 % at 524 000 rooms, which create_impl now cuts into 256-room blocks anyway; the lone-wavefront kernels, which fill
    // nothing, are level at 131 072 rooms and slower beyond).  A mixed batch's segments are padded to its block size, so it keeps it.
    if (single && b->segs.size() == 1 && !low && bt < 256u && single_block_env == 0u) {
        bt = 256u;
        blocks = (uint32_t)((b->segs[0].dev.rooms + bt - 1u) / bt);
        rpb = bt;
    }
    // Half-filled lone wavefronts: a fused launch over a single-game batch of at most 32 768 rooms runs 32 rooms per wavefront (still
    // at most one wavefront per SIMD).  A lone wavefront's time is its instruction stream, whatever its lanes hold, and a wavefront of
    // 32 rooms almost never needs the second round of its action queue that 43 % (Werewolf x 8) to 86 % (x 12) of the 64-room
    // wave-turns take: at 32 768 rooms Werewolf x 8 1.09 -> 0.97 us per turn (-11 %), x 12 -16 %, Two-Truths x 4 / 8 / 12 -9 / -16 /
    // -19 % (profiles/r05_ab_half_waves.txt).  Not beyond 32 768: two such wavefronts on a SIMD take 1.39 us against 1.06 for one
    // full one - at this occupancy a SIMD issues about one instruction per 4 cycles in all.  Not for single-turn launches (twice the
    // blocks to fill tables for: up to +9 %).  GE_HALF_WAVES=0 / 1 forces it off / on (A/B, and a knob run of the test suite)
    static const int half_env = [] { const char *e = getenv("GE_HALF_WAVES"); return e ? atoi(e) : -1; }();
    const bool half_fits = b->segs.size() == 1 && low && bt == 64u && (!single || half_env == 1);
    if (half_fits && (half_env == 1 || (half_env != 0 && b->segs[0].dev.rooms <= 32768u))) {
        rpb = 32u;
        blocks = (uint32_t)((b->segs[0].dev.rooms + 31u) / 32u);
    } else if (half_fits && half_env != 0 && b->segs[0].dev.rooms < 65536u) {
        // between 32 768 and 65 536 rooms: as few rooms per wavefront as still give every SIMD at most one (49 152 rooms = 1 024 wavefronts of 48):
        // us per turn against 64 rooms per wavefront (profiles/r05_rooms_per_wavefront.txt) Werewolf x 8 at 40 000 / 49 152 / 57 000 rooms
        // 1.08 / 1.06 / 1.05 -> 0.98 / 0.96 / 1.02, x 12 1.51 / 1.49 / 1.49 -> 1.29 / 1.35 / 1.43, Two-Truths x 7 1.15 -> 1.01 / 1.01 / 1.09, x 4 -4 %
        rpb = (uint32_t)((b->segs[0].dev.rooms + 1023u) / 1024u);
        blocks = (uint32_t)((b->segs[0].dev.rooms + rpb - 1u) / rpb);
    }
}

static bool launch_low(const ge_batch *b, const StepArgs &a) {
    return a.lowocc != 0u && !(b->generic && b->segs.size() > 1);   // mixed batches with generic tables: the large-batch build serves every size
}

static hipError_t launch_step(const ge_batch *b, const StepArgs &a_in, hipStream_t st) {
    const bool single = a_in.n_turns == 1u, mixed = b->segs.size() > 1, low = launch_low(b, a_in);
    uint32_t bt, blocks, rpb;
    launch_geometry(b, low, single, bt, blocks, rpb);
    LaunchShape L{dim3(blocks), dim3(bt), st, b, a_in};
    L.a.block_threads = bt;
    L.a.rooms_per_block = rpb;
    const uint32_t kind = b->segs[0].dev.kind;
    if (mixed) {
        if (b->generic) launch_mixed<false, 1>(single, L);
        else if (low) launch_mixed<true, 0>(single, L);
        else launch_mixed<false, 0>(single, L);
    } else if (b->generic) {
        if (b->gshape && !single) { if (low) launch_tt_shaped<true>(b->gshape, kind, L); else launch_tt_shaped<false>(b->gshape, kind, L); }
        else if (low) launch_kind<true, 1>(kind, single, L); else launch_kind<false, 1>(kind, single, L);
    } else {
        if (low) launch_kind<true, 0>(kind, single, L); else launch_kind<false, 0>(kind, single, L);
    }
    return hipGetLastError();
}

// The Werewolf x 12 side plane of prepared role deals (ge_kernels.inl run_ww): read by single-turn launches only, so it is
// allocated when the first such launch is due - a batch that only ever runs fused launches (max_fuse > 1 and turn counts that are
// multiples of it) never pays its 8 B per room (+20 % on a 40-byte record; 128 MiB for the whole of C4 on one GPU).
static int ensure_deal_side(ge_batch *b) {
    if (b->deal_side || b->deal_side_failed) return GE_OK;
    size_t sb = 0;
    for (Segment &s : b->segs) if (s.dev.kind == K_WW12) sb += (size_t)s.dev.rooms_padded * 8u;
    if (!sb) { b->deal_side_failed = true; return GE_OK; }       // nothing to allocate, ever
    int st = sync_impl(b);                                       // segs_dev is rewritten: nothing of this batch may be in flight
    if (st != GE_OK) return st;
    if (hipMalloc(&b->deal_side, sb) != hipSuccess) {            // a cache: without it every launch deals on the spot, as before round 4
        (void)hipGetLastError();
        b->deal_side = nullptr; b->deal_side_failed = true;
        return GE_OK;
    }
    HIP_TRY(hipMemset(b->deal_side, 0, sb));                     // tag 0 = no deal
    size_t off = 0;
    SegDev host[GE_MAX_SEGMENTS];
    memset(host, 0, sizeof host);
    for (size_t k = 0; k < b->segs.size(); k++) {
        Segment &s = b->segs[k];
        if (s.dev.kind == K_WW12) {
            s.dev.deal_side = reinterpret_cast<uint32_t *>(static_cast<char *>(b->deal_side) + off);
            off += (size_t)s.dev.rooms_padded * 8u;
        }
        host[k] = s.dev;
    }
    HIP_TRY(hipMemcpy(b->segs_dev, host, sizeof host, hipMemcpyHostToDevice));
    return GE_OK;
}

// A ge_batch_step call that needs many launches (small max_fuse: interactive or traced stepping) is
// launch-bound for small batches; the sequence is captured once per n_turns into a hipGraph whose
// launches take their first turn relative to a device word, and replayed.
constexpr uint32_t GRAPH_MIN_LAUNCHES = 4;

static hipGraphExec_t graph_for(ge_batch *b, uint32_t n_turns) {
    for (auto &g : b->graphs)
        if (g.first == n_turns) return g.second;
    if (!b->turn_dev && hipMalloc(reinterpret_cast<void **>(&b->turn_dev), sizeof(uint32_t)) != hipSuccess) return nullptr;
    if (!b->cap_stream && hipStreamCreateWithFlags(&b->cap_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipStreamBeginCapture(b->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
    bool ok = true;
    // (Round 5 also captured the launches as S independent chains over block ranges - parallel graph branches, rooms never
    // interact - to overlap one range's drain with another's steady state: the chains' launches ran on separate hardware queues
    // but never together, and every shorter launch paid its own ramp and drain; 0-65 % slower.  profiles/r05_ab_launch_chains.txt)
    for (uint32_t done = 0; done < n_turns && ok; ) {
        const uint32_t k = (n_turns - done) < b->max_fuse ? (n_turns - done) : b->max_fuse;
        StepArgs a;
        fill_args(b, a, done, k);
        a.turn_dev = b->turn_dev;
        ok = launch_step(b, a, b->cap_stream) == hipSuccess;
        done += k;
    }
    if (ok) {
        hipLaunchKernelGGL(ge_turn_bump, dim3(1), dim3(1), 0, b->cap_stream, b->turn_dev, n_turns);
        ok = hipGetLastError() == hipSuccess;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(b->cap_stream, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); return nullptr; }   // always ends the capture
    hipGraphExec_t exec = nullptr;
    if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
    (void)hipGraphDestroy(graph);
    if (!ok || !exec) return nullptr;
    if (b->graphs.size() >= 8) {
        // the oldest executable may still be running on the last stream: wait before releasing it
        (void)hipStreamSynchronize(b->last_stream);
        (void)hipGraphExecDestroy(b->graphs.front().second);
        b->graphs.erase(b->graphs.begin());
    }
    b->graphs.emplace_back(n_turns, exec);
    return exec;
}

static int step_impl(ge_batch *b, uint32_t n_turns, void *hip_stream) {
    if (b->turn + n_turns > 0xFFFFFFFFull) return GE_ERR_RANGE;
    if ((b->flags & GE_FLAG_TRACE) && n_turns > b->max_fuse) return GE_ERR_RANGE;   // the trace holds one launch
    GE_ON_DEVICE(b);
    if (n_turns % b->max_fuse == 1u || b->max_fuse == 1u) {   // a single-turn launch is due: its Werewolf x 12 segments take deals from the side plane
        int ds = ensure_deal_side(b);
        if (ds != GE_OK) return ds;
    }
    b->last_step_turns = n_turns;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t n_launch = (n_turns + b->max_fuse - 1) / b->max_fuse;
    static const bool no_graph = getenv("GE_NO_GRAPH") != nullptr;           // A/B runs
    hipGraphExec_t exec = nullptr;
    if (n_launch >= GRAPH_MIN_LAUNCHES && !b->timing && b->graphs_ok && !no_graph) {
        exec = graph_for(b, n_turns);                       // may synchronise the previous stream (eviction)
        if (!exec) { b->graphs_ok = false; (void)hipGetLastError(); }   // capture unsupported here: plain launches from now on
    }
    int ord = order_after_previous(b, st);
    if (ord != GE_OK) return ord;
    if (exec) {
        if (b->turn_dev_value != b->turn)
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->turn_dev), (int)(uint32_t)b->turn, 1, st));
        HIP_TRY(hipGraphLaunch(exec, st));
        b->turn += n_turns;
        b->turn_dev_value = b->turn;
        b->launches += n_launch;
        return GE_OK;
    }
    uint32_t left = n_turns;
    while (left) {
        const uint32_t k = left < b->max_fuse ? left : b->max_fuse;
        StepArgs a;
        fill_args(b, a, (uint32_t)b->turn, k);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (b->timing) {
            if (b->events_used == b->events.size()) {
                HIP_TRY(hipEventCreate(&e0));
                HIP_TRY(hipEventCreate(&e1));
                b->events.emplace_back(e0, e1);
            }
            e0 = b->events[b->events_used].first; e1 = b->events[b->events_used].second;
            b->events_used++;
            HIP_TRY(hipEventRecord(e0, st));
        }
        HIP_TRY(launch_step(b, a, st));
        if (b->timing) HIP_TRY(hipEventRecord(e1, st));
        b->launches++;
        b->turn += k;
        left -= k;
    }
    return GE_OK;
}

static int kernel_time_impl(ge_batch *b, int reset, double *total_ms, uint64_t *launches) {
    GE_ON_DEVICE(b);
    int st = sync_impl(b);
    if (st != GE_OK) return st;
    for (size_t i = 0; i < b->events_used; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->events[i].first, b->events[i].second));
        b->timed_ms += ms;
    }
    b->events_used = 0;
    if (total_ms) *total_ms = b->timed_ms;
    if (launches) *launches = b->launches;
    if (reset) { b->timed_ms = 0.0; b->launches = 0; }
    return GE_OK;
}

// Host side of ge_batch_read_rooms / write_rooms.  What crosses PCIe is the packed record (32 - 48 B per room, plane by plane,
// through a pinned staging buffer); the canonical view (228 B per room) is built from it / folded into it by the host cores, rooms
// split over a few threads - a single thread converts ~20 M rooms/s, which made a 1 M-room read 53 ms against 3 ms of copies.
static unsigned io_workers(uint64_t rooms) {
    unsigned n = 16;
    if (const char *e = getenv("GE_IO_THREADS")) n = (unsigned)atoi(e);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    const uint64_t by_size = rooms / 16384u + 1u;                 // a thread is worth starting for >= 16 K rooms
    return (unsigned)std::max<uint64_t>(1u, std::min<uint64_t>(n, by_size));
}

template <class F> static void for_room_ranges(uint64_t rooms, F &&f) {   // f(lo, hi) over disjoint ranges, on the calling thread too
    const unsigned n = io_workers(rooms);
    if (n <= 1) { f((uint64_t)0, rooms); return; }
    const uint64_t per = (rooms + n - 1) / n;
    std::vector<std::thread> th;
    th.reserve(n - 1);
    for (unsigned t = 1; t < n; t++) {
        const uint64_t lo = std::min<uint64_t>(rooms, t * per), hi = std::min<uint64_t>(rooms, lo + per);
        if (lo >= hi) continue;
        try {
            th.emplace_back([&f, lo, hi] { f(lo, hi); });
        } catch (...) {                                          // no thread to be had: this range on the calling thread
            f(lo, hi);
        }
    }
    f((uint64_t)0, std::min<uint64_t>(rooms, per));
    for (std::thread &x : th) x.join();
}

static int io_stage(ge_batch *b, size_t bytes, uint32_t **out) {          // pinned, grown on demand, freed with the batch
    if (bytes > b->io_cap) {
        if (b->io_buf) (void)hipHostFree(b->io_buf);
        b->io_buf = nullptr; b->io_cap = 0;
        HIP_TRY(hipHostMalloc(&b->io_buf, bytes, hipHostMallocDefault));
        b->io_cap = bytes;
    }
    *out = static_cast<uint32_t *>(b->io_buf);
    return GE_OK;
}

static int rooms_io(ge_batch *b, uint64_t first, uint64_t count, ge_room_view *dst, const ge_room_view *src) {
    if (first + count > b->n_rooms || first + count < first) return GE_ERR_RANGE;
    GE_ON_DEVICE(b);
    int st = sync_impl(b);
    if (st != GE_OK) return st;
    struct Part { Segment *s; uint64_t lo, r0, nr; };
    std::vector<Part> parts;
    for (Segment &s : b->segs) {
        const uint64_t lo = first > s.local_first ? first : s.local_first;
        const uint64_t hi = (first + count) < (s.local_first + s.dev.rooms) ? (first + count) : (s.local_first + s.dev.rooms);
        if (lo < hi) parts.push_back({&s, lo, lo - s.local_first, hi - lo});
    }
    if (src) {                                                   // nothing is written unless every view fits its segment
        for (const Part &p : parts) {
            std::atomic<uint64_t> bad{~0ull};                    // the lowest offending room of the part
            const ge_room_view *v = src + (p.lo - first);
            const uint32_t n = p.s->dev.n_players;
            const ge_game_table &tb = p.s->table;
            for_room_ranges(p.nr, [&](uint64_t a, uint64_t z) {
                for (uint64_t r = a; r < z; r++) {
                    if (view_fits(v[r], tb, n)) continue;
                    uint64_t cur = bad.load(std::memory_order_relaxed);
                    while (r < cur && !bad.compare_exchange_weak(cur, r, std::memory_order_relaxed)) {}
                    return;
                }
            });
            if (bad.load() != ~0ull) { g_last_rejected_room = p.lo + bad.load(); return GE_ERR_ARG; }
        }
    }
    for (const Part &p : parts) {
        Segment &s = *p.s;
        const int W = (int)s.dev.words, np = planes_of(W);
        // plane j of the range at word offset nr * (words of planes < j); a few rooms (the single-room latency path) go through
        // the stack, large ranges through the pinned buffer
        uint32_t small[1024];
        uint32_t *stage = small;
        const size_t stage_bytes = (size_t)p.nr * (size_t)W * 4u;
        if (stage_bytes > sizeof small && (st = io_stage(b, stage_bytes, &stage)) != GE_OK) return st;
        size_t off[4] = {0, 0, 0, 0};
        for (int j = 0, acc = 0; j < np; j++) { off[j] = (size_t)p.nr * acc; acc += plane_words(W, j); }
        auto dev_of = [&](int j) { return reinterpret_cast<char *>(s.dev.base) + plane_offset(s.dev.rooms_padded, j) + p.r0 * (uint64_t)plane_words(W, j) * 4u; };
        if (src) {
            const ge_room_view *v = src + (p.lo - first);
            for_room_ranges(p.nr, [&](uint64_t a, uint64_t z) {
                uint32_t w[12];
                for (uint64_t r = a; r < z; r++) {
                    view_to_words(s.dev.kind, v[r], s.table, w);
                    for (int j = 0; j < np; j++)
                        for (int x = 0, pw = plane_words(W, j); x < pw; x++) stage[off[j] + r * pw + x] = w[4 * j + x];
                }
            });
            for (int j = 0; j < np; j++)
                HIP_TRY(hipMemcpy(dev_of(j), stage + off[j], (size_t)p.nr * plane_words(W, j) * 4u, hipMemcpyHostToDevice));
        } else {
            for (int j = 0; j < np; j++)     // synchronous: the conversion below needs every plane, and the batch was drained above
                HIP_TRY(hipMemcpy(stage + off[j], dev_of(j), (size_t)p.nr * plane_words(W, j) * 4u, hipMemcpyDeviceToHost));
            ge_room_view *v = dst + (p.lo - first);
            for_room_ranges(p.nr, [&](uint64_t a, uint64_t z) {
                uint32_t w[12] = {0};
                for (uint64_t r = a; r < z; r++) {
                    for (int j = 0; j < np; j++)
                        for (int x = 0, pw = plane_words(W, j); x < pw; x++) w[4 * j + x] = stage[off[j] + r * pw + x];
                    words_to_view(s.dev.kind, w, s.table, (int)s.dev.n_players, v[r]);
                }
            });
        }
    }
    return GE_OK;
}

// n logged actions of host-driven players: sorted by room on the host (stable, so a room's actions keep
// their input order), applied by one device thread per distinct room
static int inject_impl(ge_batch *b, uint64_t n, const uint64_t *rooms, const uint32_t *players, const uint32_t *choices, int32_t *status) {
    if (n == 0) return GE_OK;
    if (!rooms || !players || !choices || n > 0x7FFFFFFFull) return GE_ERR_ARG;
    GE_ON_DEVICE(b);
    int st = sync_impl(b);
    if (st != GE_OK) return st;
    std::vector<uint32_t> order((size_t)n);
    for (size_t k = 0; k < n; k++) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return rooms[x] < rooms[y]; });
    // one upload: [rooms u64 x n][players u32 x n][choices u32 x n][group u32 x (n+1)][status i32 x n]
    const size_t off_pl = 8 * (size_t)n, off_ch = off_pl + 4 * (size_t)n, off_gr = off_ch + 4 * (size_t)n;
    const size_t off_st = off_gr + 4 * ((size_t)n + 1), total = off_st + 4 * (size_t)n;
    std::vector<unsigned char> host(total);
    uint64_t *h_rooms = reinterpret_cast<uint64_t *>(host.data());
    uint32_t *h_pl = reinterpret_cast<uint32_t *>(host.data() + off_pl), *h_ch = reinterpret_cast<uint32_t *>(host.data() + off_ch);
    uint32_t *h_gr = reinterpret_cast<uint32_t *>(host.data() + off_gr);
    uint32_t groups = 0;
    for (size_t k = 0; k < n; k++) {
        h_rooms[k] = rooms[order[k]]; h_pl[k] = players[order[k]]; h_ch[k] = choices[order[k]];
        if (k == 0 || h_rooms[k] != h_rooms[k - 1]) h_gr[groups++] = (uint32_t)k;
    }
    h_gr[groups] = (uint32_t)n;
    if (b->inj_cap < total) {
        if (b->inj_buf) (void)hipFree(b->inj_buf);     // inj_buf only: io_buf belongs to io_stage and ge_batch_destroy
        b->inj_buf = nullptr; b->inj_cap = 0;
        const size_t cap = total < 4096 ? 4096 : total * 2;
        if (hipMalloc(&b->inj_buf, cap) != hipSuccess) return GE_ERR_NOMEM;
        b->inj_cap = cap;
    }
    char *dev = static_cast<char *>(b->inj_buf);
    HIP_TRY(hipMemcpyAsync(dev, host.data(), off_st, hipMemcpyHostToDevice, b->last_stream));
    InjectArgs a;
    a.rooms = reinterpret_cast<const uint64_t *>(dev); a.players = reinterpret_cast<const uint32_t *>(dev + off_pl);
    a.choices = reinterpret_cast<const uint32_t *>(dev + off_ch); a.group = reinterpret_cast<const uint32_t *>(dev + off_gr);
    a.status = reinterpret_cast<int32_t *>(dev + off_st);
    a.n_groups = groups; a.n_seg = (uint32_t)b->segs.size();
    hipLaunchKernelGGL(ge_inject_kernel, dim3((groups + 63u) / 64u), dim3(64), 0, b->last_stream, a, b->segs_dev, b->tables);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h_st((size_t)n);
    HIP_TRY(hipMemcpyAsync(h_st.data(), dev + off_st, 4 * (size_t)n, hipMemcpyDeviceToHost, b->last_stream));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    int first_bad = GE_OK;
    size_t first_bad_at = (size_t)n;
    for (size_t k = 0; k < n; k++) {
        if (status) status[order[k]] = h_st[k];
        if (h_st[k] != GE_OK && order[k] < first_bad_at) { first_bad_at = order[k]; first_bad = h_st[k]; }
    }
    return first_bad;
}

static int read_events_impl(ge_batch *b, uint64_t first, uint64_t count, uint32_t *n_turns, ge_turn_event *dst, size_t cap_bytes) {
    if (!(b->flags & GE_FLAG_TRACE)) return GE_ERR_UNSUPPORTED;
    if (first + count > b->n_rooms || first + count < first) return GE_ERR_RANGE;
    const uint32_t T = b->last_step_turns;
    *n_turns = T;
    if (cap_bytes / sizeof(ge_turn_event) < count * (uint64_t)T) return GE_ERR_ARG;
    GE_ON_DEVICE(b);
    int st = sync_impl(b);
    if (st != GE_OK) return st;
    std::vector<uint32_t> buf;
    for (Segment &s : b->segs) {
        const uint64_t lo = first > s.local_first ? first : s.local_first;
        const uint64_t hi = (first + count) < (s.local_first + s.dev.rooms) ? (first + count) : (s.local_first + s.dev.rooms);
        if (lo >= hi) continue;
        const uint64_t r0 = lo - s.local_first, nr = hi - lo;
        buf.resize((size_t)nr * 4);
        for (uint32_t t = 0; t < T; t++) {
            const char *dev = reinterpret_cast<const char *>(s.dev.trace) + ((uint64_t)t * s.dev.rooms_padded + r0) * 16u;
            HIP_TRY(hipMemcpy(buf.data(), dev, (size_t)nr * 16u, hipMemcpyDeviceToHost));
            for (uint64_t r = 0; r < nr; r++) {
                const uint32_t *w = &buf[r * 4];
                ge_turn_event &e = dst[(lo - first + r) * T + t];
                memset(&e, 0, sizeof e);
                e.turn = w[0];
                e.from_phase_id = s.table.rows[w[1] & 255u].phase_id;
                e.to_phase_id = s.table.rows[(w[1] >> 8) & 255u].phase_id;
                e.restarted = (w[1] >> 16) & 1u;
                e.acted_now = (uint16_t)(w[1] >> 20);
                const uint64_t ch = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
                for (int i = 0; i < 16; i++) e.choice[i] = (uint8_t)((ch >> (4 * i)) & 15u);
            }
        }
    }
    return GE_OK;
}

static int summary_impl(ge_batch *b, ge_summary *out) {
    GE_ON_DEVICE(b);
    hipStream_t st = b->last_stream;
    HIP_TRY(hipMemsetAsync(b->sum_dev, 0, sizeof(unsigned long long) * 64, st));
    StepArgs a;
    fill_args(b, a, (uint32_t)b->turn, 0);
    // the summary kernel uses 256-thread blocks of its own
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < a.n_seg; k++) {
        a.block_begin[k] = blocks;
        blocks += (uint32_t)((b->segs[k].dev.rooms + 256u * SUM_CHUNKS - 1u) / (256u * SUM_CHUNKS));
    }
    hipLaunchKernelGGL(ge_summary_kernel, dim3(blocks), dim3(256), 0, st, a, b->segs_dev, b->tables, b->sum_dev);
    HIP_TRY(hipGetLastError());
    unsigned long long h[64];
    HIP_TRY(hipMemcpyAsync(h, b->sum_dev, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    b->pending = false;
    memset(out, 0, sizeof *out);
    out->rooms = b->n_rooms;
    out->finished = h[0]; out->village_wins = h[1]; out->wolf_wins = h[2]; out->alive_players = h[3];
    out->sum_end_turn = h[4];
    for (int i = 0; i < 16; i++) { out->end_turn_hist[i] = h[5 + i]; out->score_hist[i] = h[21 + i]; }
    out->checksum = h[37];
    out->games_recycled = h[38];
    out->turn = b->turn;
    return GE_OK;
}

extern "C" {

int ge_batch_create(const ge_batch_desc *desc, ge_batch **out) {
    return guarded([&] { return create_impl(desc, out); });
}

int ge_batch_reset(ge_batch *b) {
    if (!b) return GE_ERR_ARG;
    return guarded([&] { return reset_impl(b); });
}

int ge_batch_set_turn(ge_batch *b, uint64_t turn) {
    if (!b) return GE_ERR_ARG;
    if (turn > 0xFFFFFFFFull) return GE_ERR_RANGE;
    return guarded([&] {
        GE_ON_DEVICE(b);
        int st = sync_impl(b);
        if (st != GE_OK) return st;
        b->turn = turn;                                    // the graph path re-seeds its device word when it differs
        uint64_t most = 0;
        for (const Segment &sg : b->segs) most = ((sg.dev.kind == K_WW8 || sg.dev.deal_side) && sg.dev.rooms_padded > most) ? sg.dev.rooms_padded : most;
        if (most) {
            const uint32_t blocks = (uint32_t)((most / 256u) < 2048u ? (most / 256u) : 2048u);
            hipLaunchKernelGGL(ge_clear_deal_cache, dim3(blocks ? blocks : 1u), dim3(256), 0, b->last_stream, b->segs_dev, (uint32_t)b->segs.size());
            HIP_TRY(hipGetLastError());
            return sync_impl(b);
        }
        return (int)GE_OK;
    });
}

int ge_batch_set_timing(ge_batch *b, int on) {
    if (!b) return GE_ERR_ARG;
    b->timing = on != 0;
    return GE_OK;
}

int ge_batch_step(ge_batch *b, uint32_t n_turns, void *hip_stream) {
    if (!b) return GE_ERR_ARG;
    return guarded([&] { return step_impl(b, n_turns, hip_stream); });
}

int ge_batch_sync(ge_batch *b) {
    if (!b) return GE_ERR_ARG;
    return guarded([&] { GE_ON_DEVICE(b); return sync_impl(b); });
}

int ge_batch_kernel_time(ge_batch *b, int reset, double *total_ms, uint64_t *launches) {
    if (!b) return GE_ERR_ARG;
    return guarded([&] { return kernel_time_impl(b, reset, total_ms, launches); });
}

int ge_batch_turn(const ge_batch *b, uint64_t *turn) {
    if (!b || !turn) return GE_ERR_ARG;
    *turn = b->turn;
    return GE_OK;
}

int ge_batch_n_rooms(const ge_batch *b, uint64_t *n) {
    if (!b || !n) return GE_ERR_ARG;
    *n = b->n_rooms;
    return GE_OK;
}

int ge_batch_read_rooms(ge_batch *b, uint64_t first, uint64_t count, ge_room_view *dst, size_t cap_bytes) {
    if (!b || (!dst && count)) return GE_ERR_ARG;
    if (cap_bytes / sizeof(ge_room_view) < count) return GE_ERR_ARG;
    return guarded([&] { return rooms_io(b, first, count, dst, nullptr); });
}

int ge_batch_write_rooms(ge_batch *b, uint64_t first, uint64_t count, const ge_room_view *src) {
    if (!b || (!src && count)) return GE_ERR_ARG;
    return guarded([&] { return rooms_io(b, first, count, nullptr, src); });
}

int ge_batch_inject_actions(ge_batch *b, uint64_t n, const uint64_t *rooms, const uint32_t *player_ids,
                            const uint32_t *choices, int32_t *status) {
    if (!b) return GE_ERR_ARG;
    return guarded([&] { return inject_impl(b, n, rooms, player_ids, choices, status); });
}

int ge_batch_inject_action(ge_batch *b, uint64_t room, uint32_t player_id, uint32_t choice) {
    if (!b || room >= b->n_rooms) return GE_ERR_ARG;
    return ge_batch_inject_actions(b, 1, &room, &player_id, &choice, nullptr);
}

int ge_batch_read_events(ge_batch *b, uint64_t first, uint64_t count, uint32_t *n_turns, ge_turn_event *dst, size_t cap_bytes) {
    if (!b || !n_turns || (!dst && count)) return GE_ERR_ARG;
    return guarded([&] { return read_events_impl(b, first, count, n_turns, dst, cap_bytes); });
}

int ge_batch_summary(ge_batch *b, ge_summary *out) {
    if (!b || !out) return GE_ERR_ARG;
    return guarded([&] { return summary_impl(b, out); });
}

int ge_batch_state(ge_batch *b, uint32_t segment, void **dev_ptr, size_t *bytes, uint32_t *bytes_per_room) {
    if (!b || segment >= b->segs.size()) return GE_ERR_ARG;
    const SegDev &d = b->segs[segment].dev;
    if (dev_ptr) *dev_ptr = d.base;
    if (bytes) *bytes = (size_t)planes_of((int)d.words) * 16u * d.rooms_padded;
    if (bytes_per_room) *bytes_per_room = d.words * 4u;
    return GE_OK;
}

void ge_batch_destroy(ge_batch *b) {
    if (!b) return;
    {
        DeviceGuard dg(b->device);
        (void)hipStreamSynchronize(b->last_stream);               // nothing of this batch may still be running
#if GE_STAMPS
        if (b->stamps_dev) {
            unsigned long long h[8] = {0};
            (void)hipMemcpy(h, b->stamps_dev, 64, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(getenv("GE_STAMPS_OUT"), "a")) {
                fprintf(f, "{\"rooms\": %llu, \"wave_turns\": %llu, \"seg\": [%llu, %llu, %llu, %llu], \"wave_shader_cycles\": %llu, \"wave_realtime_ticks_100MHz\": %llu}\n",
                        (unsigned long long)b->n_rooms, h[4], h[0], h[1], h[2], h[3], h[5], h[6]);
                fclose(f);
            }
#if GE_STAMPS == 2
            if (b->segs.size() == 1) {                                       // the last long launch's timeline, one record per wavefront
                const size_t nw = (size_t)((b->n_rooms + 63) / 64);
                std::vector<unsigned long long> log(4 * nw);
                (void)hipMemcpy(log.data(), b->stamps_dev + 8, 32 * nw, hipMemcpyDeviceToHost);
                std::string path = std::string(getenv("GE_STAMPS_OUT")) + ".waves.bin";
                if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(log.data(), 32, nw, f); fclose(f); }
            }
#endif
            (void)hipFree(b->stamps_dev);
        }
#endif
        for (auto &ev : b->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        if (b->order_ev) (void)hipEventDestroy(b->order_ev);
        for (auto &g : b->graphs) (void)hipGraphExecDestroy(g.second);
        if (b->cap_stream) (void)hipStreamDestroy(b->cap_stream);
        if (b->turn_dev) (void)hipFree(b->turn_dev);
        if (b->inj_buf) (void)hipFree(b->inj_buf);
        if (b->io_buf) (void)hipHostFree(b->io_buf);
        if (b->state) (void)hipFree(b->state);
        if (b->trace) (void)hipFree(b->trace);
        if (b->deal_side) (void)hipFree(b->deal_side);
        if (b->tables) (void)hipFree(b->tables);
        if (b->segs_dev) (void)hipFree(b->segs_dev);
        if (b->sum_dev) (void)hipFree(b->sum_dev);
    }
    delete b;
}

}  // extern "C"

#include "ge_group.inl"
