// Does an instruction's ENCODING SIZE set its issue cost on gfx950?  tools/microbench/issue_probe.hip (round 2) found v_bfe_u32, v_mul_lo_u32 and a
// DPP add at ~4.3 SIMD cycles per instruction where v_add_u32 takes ~2.5 (4 wavefronts per SIMD) - all three are 8-byte encodings.  This probe
// times straight-line blocks of one instruction form at 1, 2, 4 and 8 wavefronts per SIMD, 4 independent chains per wavefront:
//   4-byte forms: VOP2 v_xor_b32 / v_and_b32 / v_lshrrev_b32 (inline constant), VOP1 v_mov_b32-like (v_not_b32), VOP2 v_cndmask_b32 (vcc)
//   8-byte forms: the same VOP2 op with a 32-bit LITERAL, the same op forced into VOP3 (_e64), genuine 3-operand VOP3s (v_and_or_b32, v_lshl_or_b32,
//                 v_bfi_b32, v_perm_b32, v_add3_u32, v_bfe_u32, v_alignbit_b32), v_mul_lo_u32, v_bcnt_u32_b32, a DPP move
//   build: hipcc --offload-arch=gfx950 -O3 -o encoding_probe encoding_probe.hip      run: ./encoding_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REP = 128, TRIPS = 400;

#define I0(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k))
#define I1(x) asm volatile("v_xor_b32 %0, 0x12345678, %0" : "+v"(x))
#define I2(x) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(x) : "v"(k))
#define I3(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m))
#define I4(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(k))
#define I5(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(k), "v"(m))
#define I6(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m))
#define I7(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m))
#define I8(x) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x))
#define I9(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(k))
#define I10(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define I11(x) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(k))
#define I12(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x))
#define I13(x) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x))
#define I14(x) asm volatile("v_not_b32 %0, %0" : "+v"(x))
#define I15(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k))
#define I16(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(k), "s"(sm))
#define I17(x) asm volatile("v_and_b32 %0, 0x0F0F0F0F, %0" : "+v"(x))
#define I18(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(sk))
#define I19(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define I20(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m))
#define I21(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k))
#define I22(x) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x))
#define I23(x) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(y))
#define I24(x) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc")
#define I25(x) asm volatile("v_cmp_gt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(k), "s"(sm))
#define I26(x) asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(x), "v"(k) : "vcc")
#define I27(x) asm volatile("v_cmp_gt_u32_e64 %1, %0, %2" :: "v"(x), "s"(sm), "v"(k))
#define I28(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(k), "v"(m))
#define I29(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m))
#define I30(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(k))
#define I31(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define I32(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(k))
#define I33(x) asm volatile("v_sub_u32 %0, 0, %0\n\tv_xor_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k), "v"(m))

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t k, uint32_t m) {
    uint32_t a = threadIdx.x * 4u + 1, b = a + 4, c = a + 8, d = a + 12;
    unsigned long long sm = __builtin_amdgcn_read_exec();
    uint32_t sk = __builtin_amdgcn_readfirstlane(k | 0x0F0F0F0Fu);
    unsigned long long y = a;
    for (int t = 0; t < TRIPS; t++) {
#pragma unroll
        for (int i = 0; i < REP; i++) {
#define ALL4(M) { M(a); M(b); M(c); M(d); }
            if (KIND == 0) ALL4(I0) if (KIND == 1) ALL4(I1) if (KIND == 2) ALL4(I2) if (KIND == 3) ALL4(I3) if (KIND == 4) ALL4(I4) if (KIND == 5) ALL4(I5)
            if (KIND == 6) ALL4(I6) if (KIND == 7) ALL4(I7) if (KIND == 8) ALL4(I8) if (KIND == 9) ALL4(I9) if (KIND == 10) ALL4(I10) if (KIND == 11) ALL4(I11)
            if (KIND == 12) ALL4(I12) if (KIND == 13) ALL4(I13) if (KIND == 14) ALL4(I14) if (KIND == 15) ALL4(I15) if (KIND == 16) ALL4(I16) if (KIND == 17) ALL4(I17)
            if (KIND == 18) ALL4(I18) if (KIND == 19) ALL4(I19) if (KIND == 20) ALL4(I20) if (KIND == 21) ALL4(I21) if (KIND == 22) ALL4(I22)
            if (KIND == 23) { I23(y); I23(y); I23(y); I23(y); }
            if (KIND == 24) ALL4(I24) if (KIND == 25) ALL4(I25) if (KIND == 26) ALL4(I26) if (KIND == 27) ALL4(I27) if (KIND == 28) ALL4(I28) if (KIND == 29) ALL4(I29)
            if (KIND == 30) ALL4(I30) if (KIND == 31) ALL4(I31) if (KIND == 32) ALL4(I32) if (KIND == 33) ALL4(I33)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)y;
}

template <int KIND> double run(int blocks, uint32_t *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 3u, 0x00FF00FFu);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float m; CHECK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2] * 1e-3;
}

int main() {
    uint32_t *out;
    CHECK(hipMalloc(&out, 1u << 26));
    const char *names[] = {"v_xor_b32 VOP2 (4 B)", "v_xor_b32 + 32-bit literal (8 B)", "v_xor_b32_e64 (VOP3, 8 B)", "v_and_or_b32 (VOP3)", "v_lshl_or_b32 (VOP3)", "v_bfi_b32 (VOP3)",
                           "v_perm_b32 (VOP3)", "v_add3_u32 (VOP3)", "v_bfe_u32 (VOP3)", "v_alignbit_b32 (VOP3)", "v_mul_lo_u32 (VOP3)", "v_bcnt_u32_b32 (VOP3)", "v_mov_b32_dpp (8 B)",
                           "v_lshrrev_b32 VOP2 inline const (4 B)", "v_not_b32 VOP1 (4 B)", "v_cndmask_b32 VOP2 vcc (4 B)", "v_cndmask_b32_e64 sgpr mask (8 B)",
                           "v_and_b32 + 32-bit literal (8 B)", "v_and_b32 VOP2 sgpr operand (4 B)", "v_mul_hi_u32 (VOP3)", "v_mad_u32_u24 (VOP3)", "v_mul_u32_u24 VOP2 (4 B)",
                           "v_ffbl_b32 VOP1 (4 B)", "v_lshlrev_b64 (VOP3, 64-bit)",
                           "v_cmp (vcc) + v_cndmask (vcc): per instruction of the pair", "v_cmp_e64 (sgpr pair) + v_cndmask_e64: per instruction of the pair", "v_cmp_gt_u32 -> vcc alone",
                           "v_cmp_gt_u32_e64 -> sgpr pair alone", "v_bitop3_b32", "v_or3_b32", "v_lshl_add_u32", "v_add_u32 VOP2", "v_mov_b32", "select by mask: sub, xor, and, xor (per instruction of the 4)"};
    const int per[] = {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1, 2,2,1,1,1,1,1,1,1,4};
    printf("# SIMD cycles per wave-instruction = kernel time x 2.4 GHz x (wavefronts per SIMD)^-1 ... reported as cycles per instruction per SIMD (time x 2.4e9 / (instructions of one wavefront x wavefronts per SIMD))\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;      // 256 CUs x wps blocks of 4 wavefronts = wps wavefronts per SIMD
        printf("## %d wavefronts per SIMD, 4 independent chains per wavefront\n", wps);
        const double n = (double)REP * TRIPS * 4.0 * wps;
#define ROW(K) printf("%-72s %6.2f\n", names[K], run<K>(blocks, out) * 2.4e9 / (n * per[K]));
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31) ROW(32) ROW(33)
    }
    return 0;
}
