// ge_group.inl — one host process, N devices: rooms sharded over the GPUs of a node, one RCCL all-gather of the per-GPU
// summaries (included at the end of ge_step.hip: same translation unit, it needs ge_batch's internals).
//
// The reference runs one LangGraph thread per room and rooms never interact (src/app/api/copilotkit/route.ts:24-37;
// agent/requirements.txt:1-11 has no collective library at all), so there is nothing to exchange on the step path.  The
// single exchange of the whole job is the fixed-size ge_summary per device.  Process model of SURVEY.md 8(e): a single
// process (what an N-API addon inside one Node process is), one stream per device, ncclCommInitAll + one ncclAllGather.
//
// RCCL is bound at run time (dlopen of librccl.so.1 when the first group is created), not linked: hosts that step one GPU
// never load it, and a process that already has an RCCL (torch ships its own copy) keeps exactly one.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.handle, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.AllGather && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}

thread_local int g_last_comm = 0;          // ncclResult_t of the last GE_ERR_COMM on this thread

#define NCCL_TRY(expr)                                   \
    do {                                                 \
        ncclResult_t r__ = (expr);                       \
        if (r__ != ncclSuccess) { g_last_comm = (int)r__; return (int)GE_ERR_COMM; } \
    } while (0)

constexpr int SUMMARY_WORDS = (int)(sizeof(ge_summary) / sizeof(uint64_t));
static_assert(sizeof(ge_summary) == SUMMARY_WORDS * sizeof(uint64_t), "ge_summary is all 64-bit words");

// the raw accumulators of ge_summary_kernel -> the words of a ge_summary, on the device (what the all-gather carries)
__global__ void ge_summary_pack(const unsigned long long *__restrict__ acc, unsigned long long rooms, unsigned long long turn,
                                unsigned long long *__restrict__ out) {
    const uint32_t i = threadIdx.x;
    if (i >= (uint32_t)SUMMARY_WORDS) return;
    // ge_summary: rooms finished village wolf alive sum_end | end_hist[16] | score_hist[16] | checksum turn games_recycled
    unsigned long long v;
    if (i == 0) v = rooms;
    else if (i <= 5) v = acc[i - 1];
    else if (i < 22) v = acc[5 + (i - 6)];
    else if (i < 38) v = acc[21 + (i - 22)];
    else if (i == 38) v = acc[37];
    else if (i == 39) v = turn;
    else v = acc[38];
    out[i] = v;
}

}  // namespace

struct ge_group {
    std::vector<ge_batch *> shards;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<unsigned long long *> send, recv;   // per device: its summary words; the gathered [n][SUMMARY_WORDS]
    bool comms_up = false;
};

// this device's ge_summary words into `words_dev`, on `st` (ordered behind the batch's earlier work; asynchronous)
static int summary_words_async(ge_batch *b, hipStream_t st, unsigned long long *words_dev) {
    GE_ON_DEVICE(b);
    int ord = order_after_previous(b, st);
    if (ord != GE_OK) return ord;
    HIP_TRY(hipMemsetAsync(b->sum_dev, 0, sizeof(unsigned long long) * 64, st));
    StepArgs a;
    fill_args(b, a, (uint32_t)b->turn, 0);
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < a.n_seg; k++) {
        a.block_begin[k] = blocks;
        blocks += (uint32_t)((b->segs[k].dev.rooms + 256u * SUM_CHUNKS - 1u) / (256u * SUM_CHUNKS));
    }
    hipLaunchKernelGGL(ge_summary_kernel, dim3(blocks), dim3(256), 0, st, a, b->segs_dev, b->tables, b->sum_dev);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ge_summary_pack, dim3(1), dim3(64), 0, st, b->sum_dev, (unsigned long long)b->n_rooms, (unsigned long long)b->turn, words_dev);
    HIP_TRY(hipGetLastError());
    return GE_OK;
}

static void group_free(ge_group *g) {
    if (!g) return;
    for (size_t i = 0; i < g->devices.size(); i++) {
        DeviceGuard dg(g->devices[i]);
        if (i < g->streams.size() && g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
        if (g->comms_up && i < g->comms.size() && g->comms[i]) (void)rccl().CommDestroy(g->comms[i]);
        if (i < g->shards.size() && g->shards[i]) ge_batch_destroy(g->shards[i]);
        if (i < g->send.size() && g->send[i]) (void)hipFree(g->send[i]);
        if (i < g->recv.size() && g->recv[i]) (void)hipFree(g->recv[i]);
        if (i < g->streams.size() && g->streams[i]) (void)hipStreamDestroy(g->streams[i]);
    }
    delete g;
}

static int group_create_impl(const ge_batch_desc *desc, const int *devices, int n, ge_group **out) {
    if (!desc || !devices || !out || n < 1 || n > 64 || desc->n_segments == 0 || desc->n_segments > GE_MAX_SEGMENTS) return GE_ERR_ARG;
    *out = nullptr;
    const int n_dev = ge_device_count();
    if (n_dev <= 0) return GE_ERR_NO_DEVICE;
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= n_dev) return GE_ERR_ARG;
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return GE_ERR_ARG;      // one rank per device: RCCL refuses duplicates (and would hang on some versions)
    }
    for (uint32_t k = 0; k < desc->n_segments; k++)
        if (desc->seg[k].n_rooms < (uint64_t)n) return GE_ERR_ARG;   // every device gets a part of every segment
    if (!rccl().ok) return GE_ERR_UNSUPPORTED;                    // no RCCL in this process and none to load
    ge_group *g = new (std::nothrow) ge_group();
    if (!g) return GE_ERR_NOMEM;
    g->devices.assign(devices, devices + n);
    g->shards.assign((size_t)n, nullptr);
    g->streams.assign((size_t)n, nullptr);
    g->comms.assign((size_t)n, nullptr);
    g->send.assign((size_t)n, nullptr);
    g->recv.assign((size_t)n, nullptr);
    int st = GE_OK;
    // device i takes the i-th of n contiguous parts of each segment; every room keeps the global index (hence the RNG stream)
    // it has in one batch of `desc` (ge_host.h group_partition: the same arithmetic a host gets from ge_group_partition)
    for (int i = 0; i < n && st == GE_OK; i++) {
        ge_batch_desc d;
        uint64_t first[GE_MAX_SEGMENTS];
        st = group_partition(*desc, n, i, &d, first);
        if (st != GE_OK) break;
        d.device = devices[i];
        st = create_impl(&d, &g->shards[(size_t)i], first);
        if (st != GE_OK) break;
        DeviceGuard dg(devices[i]);
        if (!dg.ok || hipStreamCreateWithFlags(&g->streams[(size_t)i], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&g->send[(size_t)i]), sizeof(ge_summary)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&g->recv[(size_t)i]), sizeof(ge_summary) * (size_t)n) != hipSuccess) st = GE_ERR_HIP;
    }
    if (st == GE_OK) {
        ncclResult_t r = rccl().CommInitAll(g->comms.data(), n, g->devices.data());
        if (r != ncclSuccess) { g_last_comm = (int)r; st = GE_ERR_COMM; } else g->comms_up = true;
    }
    if (st != GE_OK) { group_free(g); return st; }
    *out = g;
    return GE_OK;
}

static int group_summary_impl(ge_group *g, ge_summary *out) {
    const size_t n = g->shards.size();
    for (size_t i = 0; i < n; i++) {
        int st = summary_words_async(g->shards[i], g->streams[i], g->send[i]);
        if (st != GE_OK) return st;
    }
    // the one collective of the path: every device receives every device's summary (RCCL over xGMI)
    NCCL_TRY(rccl().GroupStart());
    for (size_t i = 0; i < n; i++) {
        ncclResult_t r = rccl().AllGather(g->send[i], g->recv[i], (size_t)SUMMARY_WORDS, ncclUint64, g->comms[i], g->streams[i]);
        if (r != ncclSuccess) { (void)rccl().GroupEnd(); g_last_comm = (int)r; return GE_ERR_COMM; }
    }
    NCCL_TRY(rccl().GroupEnd());
    std::vector<unsigned long long> host(n * (size_t)SUMMARY_WORDS);
    {
        DeviceGuard dg(g->devices[0]);
        if (!dg.ok) return GE_ERR_HIP;
        HIP_TRY(hipMemcpyAsync(host.data(), g->recv[0], host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, g->streams[0]));
        HIP_TRY(hipStreamSynchronize(g->streams[0]));
    }
    for (size_t i = 1; i < n; i++) {                              // every rank of the gather has completed before the call returns
        DeviceGuard dg(g->devices[i]);
        if (!dg.ok) return GE_ERR_HIP;
        HIP_TRY(hipStreamSynchronize(g->streams[i]));
    }
    for (size_t i = 0; i < n; i++) g->shards[i]->pending = false;
    uint64_t *o = reinterpret_cast<uint64_t *>(out);
    memset(out, 0, sizeof *out);
    for (size_t i = 0; i < n; i++)
        for (int w = 0; w < SUMMARY_WORDS; w++) o[w] += host[i * (size_t)SUMMARY_WORDS + (size_t)w];   // sums wrap mod 2^64 like the device-side ones
    out->turn = host[offsetof(ge_summary, turn) / sizeof(uint64_t)];                                     // common to all shards, not a sum
    return GE_OK;
}

extern "C" {

int ge_group_create(const ge_batch_desc *desc, const int *devices, int n_devices, ge_group **out) {
    return guarded([&] { return group_create_impl(desc, devices, n_devices, out); });
}

int ge_group_partition(const ge_batch_desc *desc, int n_parts, int part, ge_batch_desc *shard, uint64_t *seg_first) {
    if (!desc) return GE_ERR_ARG;
    return group_partition(*desc, n_parts, part, shard, seg_first);
}

int ge_batch_create_shard(const ge_batch_desc *shard, const uint64_t *seg_first, ge_batch **out) {
    if (!seg_first) return GE_ERR_ARG;
    return guarded([&] { return create_impl(shard, out, seg_first); });
}

int ge_group_size(const ge_group *g) { return g ? (int)g->shards.size() : GE_ERR_ARG; }

int ge_group_shard(ge_group *g, int i, ge_batch **out) {
    if (!g || !out || i < 0 || (size_t)i >= g->shards.size()) return GE_ERR_ARG;
    *out = g->shards[(size_t)i];
    return GE_OK;
}

int ge_group_step(ge_group *g, uint32_t n_turns) {
    if (!g) return GE_ERR_ARG;
    return guarded([&] {
        for (size_t i = 0; i < g->shards.size(); i++) {            // asynchronous: all devices step concurrently
            int st = step_impl(g->shards[i], n_turns, g->streams[i]);
            if (st != GE_OK) return st;
        }
        return (int)GE_OK;
    });
}

int ge_group_sync(ge_group *g) {
    if (!g) return GE_ERR_ARG;
    return guarded([&] {
        for (size_t i = 0; i < g->shards.size(); i++) {
            int st = ge_batch_sync(g->shards[i]);
            if (st != GE_OK) return st;
        }
        return (int)GE_OK;
    });
}

int ge_group_summary(ge_group *g, ge_summary *out) {
    if (!g || !out) return GE_ERR_ARG;
    return guarded([&] { return group_summary_impl(g, out); });
}

void ge_group_destroy(ge_group *g) { group_free(g); }

int ge_last_comm_error(void) { return g_last_comm; }

}  // extern "C"
