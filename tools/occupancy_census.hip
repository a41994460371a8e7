// How many workgroups of a given size does a gfx950 CU hold at once?  Every wave bumps a global counter, spins
// for a fixed number of s_memtime ticks, samples the counter (max seen = concurrent waves on the chip), and leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_census.hip -o tools/bin/occupancy_census && tools/bin/occupancy_census
// Motivation: the compositing kernels launch single-wave (64-thread) workgroups; if the CU admitted only 16
// workgroups, they would run at 4 waves per SIMD whatever their register budget allows.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VG>
__global__ void census(unsigned* live, unsigned* maxseen, unsigned long long spin, float* sink) {
    float keep[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) keep[i] = threadIdx.x * 0.5f + i;
    if ((threadIdx.x & 63) == 0) atomicAdd(live, 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned m = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {
        const unsigned v = __hip_atomic_load(live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = v > m ? v : m;
#pragma unroll
        for (int i = 0; i < VG; i++) keep[i] = keep[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(8);
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(maxseen, m); atomicSub(live, 1u); }
    float s = 0;
#pragma unroll
    for (int i = 0; i < VG; i++) s += keep[i];
    if (s == 123.456f) sink[0] = s;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *live, *maxseen; float* sink;
    (void)hipMalloc(&live, 4); (void)hipMalloc(&maxseen, 4); (void)hipMalloc(&sink, 4);
    const int tpbs[] = {64, 128, 256};
    printf("%s: %d CUs\n", prop.gcnArchName, cus);
    for (int vg : {8, 40}) {
        for (int tpb : tpbs) {
            (void)hipMemset(live, 0, 4); (void)hipMemset(maxseen, 0, 4);
            const int blocks = cus * 64;
            if (vg == 8) hipLaunchKernelGGL(census<8>, dim3(blocks), dim3(tpb), 0, 0, live, maxseen, 400000ull, sink);
            else hipLaunchKernelGGL(census<40>, dim3(blocks), dim3(tpb), 0, 0, live, maxseen, 400000ull, sink);
            (void)hipDeviceSynchronize();
            unsigned m = 0;
            (void)hipMemcpy(&m, maxseen, 4, hipMemcpyDeviceToHost);
            printf("  ~%2d live VGPRs, %3d threads per workgroup: max %5u waves in flight = %.1f per CU = %.1f per SIMD (%.1f workgroups per CU)\n",
                   vg, tpb, m, (double)m / cus, (double)m / cus / 4, (double)m / cus / (tpb / 64));
        }
    }
    return 0;
}
