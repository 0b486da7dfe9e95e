// What does a wavefront that is ALONE on its SIMD pay per instruction on gfx950?  (C2 = 65 536 rooms = one wavefront per
// SIMD; DESIGN.md "issue ceiling".)  Each kernel runs a long straight-line block of one instruction kind, as ONE dependent
// chain or as 2 / 4 independent chains interleaved, and reports wave cycles per instruction (s_memtime, 100 MHz -> scaled
// by the measured kernel time instead: cycles = time * 2.4 GHz / instructions).
//   build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip      run: ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REP = 256;      // instructions per chain per loop trip (unrolled)
constexpr int TRIPS = 200;

#define ADD1(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define XOR1(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k))
#define MUL1(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define BFE1(x) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x))
#define CMPSEL(x) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc")
#define SADD(x) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x) :: "scc")
#define DPP1(x) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x))
#define LDSRD(x) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x))

template <int CHAINS, int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t k) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (uint32_t)(((i * 4) + 1024) & 16380);   // every entry a valid, 4-aligned LDS byte address
    __syncthreads();
    uint32_t a = threadIdx.x * 4u, b = a + 4, c = a + 8, d = a + 12, e = a + 16, f = a + 20, g = a + 24, h = a + 28;
    uint32_t s = __builtin_amdgcn_readfirstlane(k);
    for (int t = 0; t < TRIPS; t++) {
#pragma unroll
        for (int i = 0; i < REP; i++) {
            if (KIND == 0) { ADD1(a); if (CHAINS > 1) ADD1(b); if (CHAINS > 2) { ADD1(c); ADD1(d); } if (CHAINS > 4) { ADD1(e); ADD1(f); ADD1(g); ADD1(h); } }
            if (KIND == 5) { DPP1(a); if (CHAINS > 1) DPP1(b); if (CHAINS > 2) { DPP1(c); DPP1(d); } }
            if (KIND == 6) { LDSRD(a); if (CHAINS > 1) LDSRD(b); if (CHAINS > 2) { LDSRD(c); LDSRD(d); } }
            if (KIND == 1) { MUL1(a); if (CHAINS > 1) MUL1(b); if (CHAINS > 2) { MUL1(c); MUL1(d); } }
            if (KIND == 2) { CMPSEL(a); if (CHAINS > 1) CMPSEL(b); if (CHAINS > 2) { CMPSEL(c); CMPSEL(d); } }
            if (KIND == 3) { ADD1(a); SADD(s); if (CHAINS > 1) { ADD1(b); SADD(s); } if (CHAINS > 2) { ADD1(c); SADD(s); ADD1(d); SADD(s); } }   // VALU + SALU alternating
            if (KIND == 4) { BFE1(a); if (CHAINS > 1) BFE1(b); if (CHAINS > 2) { BFE1(c); BFE1(d); } }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ s;
}

template <int CHAINS, int KIND>
double run(int blocks, int threads, uint32_t *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<CHAINS, KIND>), dim3(blocks), dim3(threads), 0, 0, out, 3u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float m; CHECK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3] * 1e-3;
}

int main() {
    uint32_t *out;
    CHECK(hipMalloc(&out, 1u << 26));
    const double ghz = 2.4e9;
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_cmp + s_nop 1 + v_cndmask (3 instr)", "v_add_u32 + s_add_u32 (2 instr)", "v_bfe_u32",
                           "v_add_u32_dpp row_shr:1", "ds_read_b32 + s_waitcnt (dependent address)"};
    // per-instruction count per chain step: KIND 2 = 3 instructions, KIND 3 = 2
    const int per[] = {1, 1, 3, 2, 1, 1, 1};
    struct Shape { const char *name; int blocks, threads; } shapes[] = {
        {"1 wave/SIMD (256 blocks x 256)", 256, 256}, {"2 waves/SIMD (512 x 256)", 512, 256}, {"4 waves/SIMD (1024 x 256)", 1024, 256}};
    printf("# cycles per instruction and wavefront (kernel time x 2.4 GHz / instructions of one wavefront), gfx950\n");
    for (auto &sh : shapes) {
        printf("## %s\n", sh.name);
#define ROW(KIND) { \
        double t1 = run<1, KIND>(sh.blocks, sh.threads, out), t2 = run<2, KIND>(sh.blocks, sh.threads, out), t4 = run<4, KIND>(sh.blocks, sh.threads, out); \
        double n = (double)REP * TRIPS * per[KIND]; \
        printf("%-42s 1 chain %6.2f   2 chains %6.2f   4 chains %6.2f\n", names[KIND], t1 * ghz / n, t2 * ghz / (2 * n), t4 * ghz / (4 * n)); }
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
        { double t8 = run<8, 0>(sh.blocks, sh.threads, out); printf("%-42s 8 chains %6.2f\n", names[0], t8 * ghz / (8.0 * REP * TRIPS)); }
    }
    return 0;
}
