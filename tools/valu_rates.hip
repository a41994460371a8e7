// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instruction kinds the
// compositing kernels are made of, measured on the whole chip TWO ways:
//   wall : HIP-event time x the NOMINAL clock (2.4 GHz) / instructions per SIMD   (what round 1 reported)
//   cyc  : s_memtime shader cycles elapsed inside a wave / instructions per SIMD  (clock-independent)
// The ratio of the two is the effective clock under that instruction stream (DVFS), which is what
// reconciles round 1's "2.6 cycles per v_mul" with MI355X_MICROARCH.md's 2 cycles per wave64 VALU op.
// Also measured: whether a half-wave with EXEC = 0 is skipped, and operand-form effects (VOP2 vs VOP3,
// SGPR / inline-constant sources, v_cmp to VCC vs to an SGPR pair, integer compare).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/bin/valu_rates && tools/bin/valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

typedef float f2 __attribute__((ext_vector_type(2)));

// EXECMASK: 0 = all lanes, 1 = low half-wave only (lanes 0-31), 2 = even lanes only, 3 = low 16 lanes only
#define KERNEL(NAME, BODY)                                                                       \
    __global__ void NAME(float* out, unsigned long long* cyc, int iters, int execmode) {         \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                            \
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2};                            \
        float c = 1.0001f;                                                                       \
        f2 cc = {1.0001f, 0.9999f};                                                              \
        float sc = __builtin_amdgcn_readfirstlane(1.0001f + (float)(iters & 1) * 1e-7f);          \
        unsigned long long m = 0x5555555555555555ull;                                            \
        const bool live = execmode == 0 || (execmode == 1 && threadIdx.x < 32) ||                \
                          (execmode == 2 && (threadIdx.x & 1) == 0) || (execmode == 3 && threadIdx.x < 16); \
        unsigned long long t0 = 0, t1 = 0;                                                       \
        if (live) {                                                                              \
            t0 = __builtin_amdgcn_s_memtime();                                                   \
            for (int i = 0; i < iters; i++) { REP16(BODY) }                                      \
            t1 = __builtin_amdgcn_s_memtime();                                                   \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                             \
            a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + sc;      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                         \
    }

// 4 independent instructions per body -> 64 instructions per loop iteration
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_fma_sgpr, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc));)
KERNEL(k_fma_const, asm volatile("v_fma_f32 %0, %0, %4, 1.0\n v_fma_f32 %1, %1, %4, 1.0\n v_fma_f32 %2, %2, %4, 1.0\n v_fma_f32 %3, %3, %4, 1.0"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(cc.x));)
KERNEL(k_fmac_sgpr, asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc), "v"(c));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_exp_mul3, asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "s"(m));)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_le_f32 s[20:21], %0, %4\n v_cmp_le_f32 s[22:23], %1, %4\n v_cmp_le_f32 s[20:21], %2, %4\n v_cmp_le_f32 s[22:23], %3, %4"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "s22", "s23");)
KERNEL(k_cmp_u32, asm volatile("v_cmp_le_u32 vcc, %0, %4\n v_cmp_le_u32 vcc, %1, %4\n v_cmp_le_u32 vcc, %2, %4\n v_cmp_le_u32 vcc, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
KERNEL(k_cmp_mul3, asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
KERNEL(k_dpp, asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_swap32, asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_salu_mix, asm volatile("v_mul_f32 %0, %0, %4\n s_and_b64 s[20:21], s[20:21], exec\n v_mul_f32 %1, %1, %4\n s_or_b64 s[22:23], s[22:23], exec"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "s22", "s23");)
KERNEL(k_fma_dep, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_mul_dep, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)

typedef void (*kern_t)(float*, unsigned long long*, int, int);
struct Case { const char* name; kern_t k; };

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz (nominal)
    printf("device %s  CUs %d  nominal clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    std::vector<Case> cases = {
        {"v_mul_f32 (VOP2)", k_mul}, {"v_mul_f32 sgpr src", k_mul_sgpr}, {"v_fmac_f32 (VOP2)", k_fmac}, {"v_fmac_f32 sgpr src", k_fmac_sgpr},
        {"v_fma_f32 (VOP3, 3 vgpr)", k_fma}, {"v_fma_f32 2 vgpr + sgpr", k_fma_sgpr}, {"v_fma_f32 2 vgpr + const", k_fma_const},
        {"v_mov_b32", k_mov}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
        {"1 exp + 3 mul", k_exp_mul3}, {"v_cndmask (sgpr mask)", k_cndmask}, {"v_cmp_le_f32 -> vcc", k_cmp_vcc},
        {"v_cmp_le_f32 -> sgpr pair", k_cmp_sgpr}, {"v_cmp_le_u32 -> vcc", k_cmp_u32}, {"1 cmp + 3 mul", k_cmp_mul3},
        {"v_add_f32_dpp row_ror", k_dpp}, {"v_permlane32_swap", k_swap32}, {"v_readlane_b32", k_readlane},
        {"2 mul + 2 salu", k_salu_mix}, {"v_fma dependent chain", k_fma_dep}, {"v_mul dependent chain", k_mul_dep}};
    const int max_blocks = cus * 4 * 8;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)max_blocks * 64 * sizeof(float));
    hipMalloc(&cyc, (size_t)max_blocks * sizeof(unsigned long long));
    std::vector<unsigned long long> h(max_blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const Case& c, int wps, int execmode) {
        const int blocks = cus * 4 * wps;
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, cyc, 10, execmode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, execmode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.begin() + blocks);
        const double med = (double)h[blocks / 2];
        const double instr_per_simd = (double)wps * iters * 64.0;
        printf("  %-28s wall %6.3f  cyc %6.3f   eff. clock %4.2f GHz  (%.3f ms)\n", c.name, ms * 1e-3 * clk / instr_per_simd,
               med / instr_per_simd, med / (ms * 1e-3) / 1e9, ms);
    };
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        printf("--- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD\n", wps);
        for (auto& c : cases) run(c, wps, 0);
    }
    const char* emn[] = {"all 64 lanes", "lanes 0-31 only", "even lanes only", "lanes 0-15 only"};
    for (int em = 1; em <= 3; em++) {
        printf("--- EXEC = %s, 8 waves per SIMD (is an empty half-wave / quarter-wave pass skipped?)\n", emn[em]);
        for (int k : {0, 4, 10, 14, 19}) run(cases[k], 8, em);
    }
    return 0;
}
