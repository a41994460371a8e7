// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instruction kinds the compositing kernels are made of, measured on the whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/bin/valu_rates && tools/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, BODY)                                                              \
    __global__ void NAME(float* out, int iters) {                                       \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                   \
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2};                   \
        float c = 1.0001f;                                                              \
        f2 cc = {1.0001f, 0.9999f};                                                     \
        unsigned long long m = 0x5555555555555555ull;                                   \
        for (int i = 0; i < iters; i++) { REP16(BODY) }                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                    \
            a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;  \
    }

// 4 independent instructions per body -> 64 instructions per loop iteration
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_pk_fma_bcast, asm volatile("v_pk_fma_f32 %0, %0, %4, %4 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %1, %1, %4, %4 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %2, %2, %4, %4 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %3, %3, %4, %4 op_sel_hi:[1,0,0]"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_exp_fma, asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "s"(m));)
KERNEL(k_cmp, asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
KERNEL(k_dpp, asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_swap32, asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_swap16, asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_fma_nop, asm volatile("v_fma_f32 %0, %0, %4, %4\n s_nop 0\n v_fma_f32 %1, %1, %4, %4\n s_nop 1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_pk_dep, asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));)
KERNEL(k_fma_dep, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)

typedef void (*kern_t)(float*, int);
struct Case { const char* name; kern_t k; };

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s  CUs %d  clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    std::vector<Case> cases = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_fma_f32 bcast", k_pk_fma_bcast},
                               {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
                               {"1 exp + 3 fma", k_exp_fma}, {"v_cndmask (sgpr mask)", k_cndmask}, {"v_cmp_le_f32", k_cmp},
                               {"v_add_f32_dpp row_ror", k_dpp}, {"v_permlane32_swap", k_swap32},
                               {"v_permlane16_swap", k_swap16}, {"2 fma + s_nop 0 + s_nop 1", k_fma_nop}, 
                               {"v_fma dependent chain", k_fma_dep}, {"v_pk_fma dependent chain", k_pk_dep}};
    float* out;
    hipMalloc(&out, (size_t)cus * 32 * 64 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        printf("--- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (instructions of the 4-op body)\n", wps);
        for (auto& c : cases) {
            const int blocks = cus * 4 * wps;
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)wps * iters * 64.0;
            printf("  %-28s %7.3f cyc/instr   (%.3f ms)\n", c.name, ms * 1e-3 * clk / instr_per_simd, ms);
        }
    }
    return 0;
}
