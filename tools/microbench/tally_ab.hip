// tally_ab.hip — A/B of the two mappings of the vote tally (BASELINE.json north star vs what ships).
//
//   A  lane = room   (shipped): the room's eight votes are one nibble array in a register; the plurality is
//      nibble-SWAR arithmetic (ge_device.h plurality<8>): 64 rooms per wavefront, no LDS, no cross-lane traffic.
//   B  lane = player (the north star's sketch: "per-room vote reductions staged in LDS and resolved with
//      wavefront shuffles"): 8 lanes per room, 8 rooms per wavefront; every lane adds its vote to its room's
//      per-candidate counters in LDS (atomic add), then the 8 lanes of a room reduce max(count << 4 | 15 - id) by
//      shuffles (ties -> lowest id, as POLICY.md says) and lane 0 writes the victim.
//
// Both compute the same function on the same votes (checked), K times per launch on perturbed votes so that the
// launch measures the tally and not the load of 8 bytes per room.  Build + run: tools/microbench/run_tally_ab.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../game_engine_amd/csrc/ge_device.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int K = 64;      // tallies per room and launch

// the k-th perturbation of a room's votes: rotate the nibble array by k players and flip who abstains
__device__ __forceinline__ uint32_t perturb(uint32_t votes, uint32_t k) { return (votes >> (4u * (k & 7u))) | (votes << ((32u - 4u * (k & 7u)) & 31u)); }

__global__ void __launch_bounds__(256) tally_lane_room(const uint32_t *__restrict__ votes, const uint32_t *__restrict__ voters,
                                                       uint32_t *__restrict__ out, uint32_t rooms) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rooms) return;
    const uint32_t v = votes[r], m = voters[r];
    uint32_t acc = 0;
#pragma unroll 4
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t mk = ((m >> (k & 7u)) | (m << (8u - (k & 7u)))) & 0xFFu;       // the voters rotate with the votes
        acc += ge::plurality<8, uint32_t>(perturb(v, k), mk) << (k & 3u);
    }
    out[r] = acc;
}

__global__ void __launch_bounds__(256) tally_lane_player(const uint32_t *__restrict__ votes, const uint32_t *__restrict__ voters,
                                                         uint32_t *__restrict__ out, uint32_t rooms) {
    __shared__ uint32_t cnt[4][8][16];                       // [wavefront][room of the wavefront][candidate id]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, g = lane >> 3, p = lane & 7u;
    const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;                 // 8 lanes per room
    const bool in = r < rooms;
    const uint32_t v = in ? votes[r] : 0u, m = in ? voters[r] : 0u;                   // same 8 bytes per room, broadcast to its 8 lanes
    uint32_t acc = 0;
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t vk = perturb(v, k), mk = ((m >> (k & 7u)) | (m << (8u - (k & 7u)))) & 0xFFu;
        const uint32_t mine = ((mk >> p) & 1u) ? (vk >> (4u * p)) & 15u : 0u;       // this player's vote (0 = none)
        cnt[wave][g][p + 1u] = 0u;                                                   // each lane clears one candidate's counter
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mine) atomicAdd(&cnt[wave][g][mine], 1u);                                // the reduction is staged in LDS
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t key = (cnt[wave][g][p + 1u] << 4) | (15u - (p + 1u));              // ties -> lowest id
        key = max(key, (uint32_t)__shfl_xor((int)key, 1, 64));                        // ... and resolved with wavefront shuffles
        key = max(key, (uint32_t)__shfl_xor((int)key, 2, 64));
        key = max(key, (uint32_t)__shfl_xor((int)key, 4, 64));
        acc += ((key >> 4) ? 15u - (key & 15u) : 0u) << (k & 3u);
    }
    if (in && p == 0u) out[r] = acc;
}

int main(int argc, char **argv) {
    const int reps = 50;
    printf("# vote tally, K = %d tallies per room and launch; device time per launch (median of %d), MI355X\n", K, reps);
    printf("# rooms  mapping        us/launch   ns/room-tally   rooms x tallies / s\n");
    for (uint32_t rooms : {65536u, 1048576u}) {
        std::vector<uint32_t> hv(rooms), hm(rooms);
        uint32_t x = 12345u;
        for (uint32_t r = 0; r < rooms; r++) {
            uint32_t v = 0;
            for (int i = 0; i < 8; i++) { x = x * 1664525u + 1013904223u; v |= (1u + ((x >> 24) % 8u)) << (4 * i); }   // a vote for 1..8
            x = x * 1664525u + 1013904223u;
            hv[r] = v; hm[r] = (x >> 16) & 0xFFu;                                                                      // who is alive and voted
        }
        uint32_t *dv, *dm, *oa, *ob;
        CHECK(hipMalloc(&dv, rooms * 4)); CHECK(hipMalloc(&dm, rooms * 4)); CHECK(hipMalloc(&oa, rooms * 4)); CHECK(hipMalloc(&ob, rooms * 4));
        CHECK(hipMemcpy(dv, hv.data(), rooms * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dm, hm.data(), rooms * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        double med[2];
        for (int which = 0; which < 2; which++) {
            std::vector<float> ms(reps);
            for (int it = -3; it < reps; it++) {
                CHECK(hipEventRecord(e0, nullptr));
                if (which == 0) hipLaunchKernelGGL(tally_lane_room, dim3((rooms + 255) / 256), dim3(256), 0, nullptr, dv, dm, oa, rooms);
                else hipLaunchKernelGGL(tally_lane_player, dim3((rooms * 8 + 255) / 256), dim3(256), 0, nullptr, dv, dm, ob, rooms);
                CHECK(hipEventRecord(e1, nullptr));
                CHECK(hipEventSynchronize(e1));
                float t;
                CHECK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 0) ms[it] = t;
            }
            std::sort(ms.begin(), ms.end());
            med[which] = ms[reps / 2];
        }
        std::vector<uint32_t> ra(rooms), rb(rooms);
        CHECK(hipMemcpy(ra.data(), oa, rooms * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(rb.data(), ob, rooms * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t r = 0; r < rooms; r++) bad += ra[r] != rb[r];
        if (bad) { fprintf(stderr, "MISMATCH: %zu of %u rooms differ between the mappings\n", bad, rooms); return 1; }
        for (int which = 0; which < 2; which++)
            printf("%8u  %-13s %10.2f %14.4f %18.3e\n", rooms, which == 0 ? "lane=room" : "lane=player", med[which] * 1e3,
                   med[which] * 1e6 / ((double)rooms * K), (double)rooms * K / (med[which] * 1e-3));
        printf("# %u rooms: lane=player / lane=room = %.2fx the time; both mappings give identical victims for every room\n", rooms, med[1] / med[0]);
        CHECK(hipFree(dv)); CHECK(hipFree(dm)); CHECK(hipFree(oa)); CHECK(hipFree(ob));
    }
    return 0;
}
