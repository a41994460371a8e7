// Micro-benchmark (round 5): does v_mfma_f32_16x16x4_f32 run BESIDE another wave's VALU stream on the same SIMD?
// A 512-thread workgroup puts two waves on each SIMD of its CU (waves w and w + 4); one workgroup per CU (96 KB of LDS).
// The low four waves run a stream of independent v_fma_f32, the high four a stream of independent (or accumulator-chained)
// f32 MFMAs; each role is timed alone (the partner exits at once) and together, in shader cycles (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// mode bit 0: VALU waves run, bit 1: MFMA waves run; chain: MFMAs accumulate into ONE register quad (dependent) or four
template <int KIND>  // 0: f32 16x16x4, 1: bf16 16x16x16 (for comparison: a matrix op that does not use the fp32 lanes)
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode, int chain) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const bool valu_role = wave < 4;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c = 1.0001f;
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    unsigned long long t0 = 0, t1 = 0;
    if (valu_role && (mode & 1)) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        }
        t1 = __builtin_amdgcn_s_memtime();
    } else if (!valu_role && (mode & 2)) {
        t0 = __builtin_amdgcn_s_memtime();
        if (KIND == 0) {
            for (int i = 0; i < iters; i++) {
                if (chain) {
                    REP4(asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n"
                                      "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                                      : "+v"(d0) : "v"(a0), "v"(a1));)
                } else {
                    REP4(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                                      "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1));)
                }
            }
        } else {
            typedef short s4 __attribute__((ext_vector_type(4)));
            s4 x = {1, 2, 3, 4}, y = {5, 6, 7, 8};
            for (int i = 0; i < iters; i++) {
                REP4(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x16_bf16 %1, %4, %5, %1\n"
                                  "v_mfma_f32_16x16x16_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x16_bf16 %3, %4, %5, %3"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(y));)
            }
        }
        asm volatile("s_nop 15\n s_nop 15");
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + d0[0] + d1[1] + d2[2] + d3[3] + lds[threadIdx.x & 7];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)cus * 512 * 4); (void)hipMalloc(&cyc, (size_t)cus * 8 * 8);
    std::vector<unsigned long long> h(cus * 8);
    const int iters = 2000;
    auto run = [&](int kind, int mode, int chain, const char* what) {
        auto kern = kind == 0 ? k<0> : k<1>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 96 * 1024, 0, out, cyc, 10, mode, chain);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 96 * 1024, 0, out, cyc, iters, mode, chain);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v, m;
        for (int b = 0; b < cus; b++)
            for (int w = 0; w < 8; w++) (w < 4 ? v : m).push_back((double)h[b * 8 + w]);
        std::sort(v.begin(), v.end()); std::sort(m.begin(), m.end());
        printf("  %-64s VALU wave: %6.2f cyc / v_fma   MFMA wave: %6.2f cyc / mfma\n", what, v[v.size() / 2] / (iters * 64.0),
               m[m.size() / 2] / (iters * 16.0));
    };
    printf("two waves per SIMD (one workgroup of 512 threads per CU), %d CUs; shader cycles per instruction of each wave\n", cus);
    run(0, 1, 0, "v_fma wave alone");
    run(0, 2, 0, "f32 16x16x4 MFMA wave alone, 4 accumulators");
    run(0, 2, 1, "f32 16x16x4 MFMA wave alone, ONE accumulator (dependent chain)");
    run(0, 3, 0, "v_fma wave + f32 MFMA wave (4 accumulators) on the same SIMD");
    run(0, 3, 1, "v_fma wave + f32 MFMA wave (dependent chain) on the same SIMD");
    run(1, 2, 0, "bf16 16x16x16 MFMA wave alone, 4 accumulators");
    run(1, 3, 0, "v_fma wave + bf16 16x16x16 MFMA wave on the same SIMD");
    return 0;
}
