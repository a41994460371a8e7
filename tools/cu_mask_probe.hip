// Probe: can a latency chain of a few LARGE workgroups (1024 threads, 64 KB of LDS — the multi-workgroup tile sort) make progress
// BESIDE a launch that fills every wave slot and all LDS (the fused sort + forward: 256 threads, ~20 KB, 8 per CU)?
//   (a) one stream                      : chain behind the filler
//   (b) two plain streams               : the chain's workgroups need four filler workgroups of ONE CU to retire together
//   (c) two streams with CU masks       : hipExtStreamCreateWithCUMask — the chain on RESERVED CUs, the filler on the others
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe ; prints the wall time of each form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void filler(float* out, int iters) {
    __shared__ float lds[5 * 1024];  // 20 KB
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1] + v;
}
__global__ __launch_bounds__(1024) void link(float* out, int iters) {
    __shared__ float lds[16 * 1024];  // 64 KB
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1] + v;
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("CUs %d\n", cus);
    float* out;
    CHK(hipMalloc(&out, 1 << 20));
    const int reserve = 16, words = (cus + 31) / 32;
    std::vector<uint32_t> m_chain(words, 0u), m_fill(words, 0u);
    for (int c = 0; c < cus; c++) (c < reserve ? m_chain : m_fill)[c / 32] |= 1u << (c % 32);
    hipStream_t s0, s1, c0, c1;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipExtStreamCreateWithCUMask(&c0, words, m_fill.data()));
    CHK(hipExtStreamCreateWithCUMask(&c1, words, m_chain.data()));
    const int fill_wgs = cus * 8 * 4, fill_iters = 60000, link_iters = 30000, links = 6, link_wgs = 4;
    auto run = [&](const char* what, hipStream_t fs, hipStream_t ls) -> int {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(filler, dim3(fill_wgs), dim3(256), 0, fs, out, fill_iters);
            for (int l = 0; l < links; l++) hipLaunchKernelGGL(link, dim3(link_wgs), dim3(1024), 0, ls, out + 65536, link_iters);
            CHK(hipStreamSynchronize(ls));
            const auto t1 = std::chrono::steady_clock::now();
            CHK(hipStreamSynchronize(fs));
            const auto t2 = std::chrono::steady_clock::now();
            if (rep == 2)
                printf("%-28s chain done %.3f ms, all done %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                       std::chrono::duration<double, std::milli>(t2 - t0).count());
        }
        return 0;
    };
    // each alone
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(filler, dim3(fill_wgs), dim3(256), 0, s0, out, fill_iters);
        CHK(hipStreamSynchronize(s0));
        auto t1 = std::chrono::steady_clock::now();
        for (int l = 0; l < links; l++) hipLaunchKernelGGL(link, dim3(link_wgs), dim3(1024), 0, s0, out + 65536, link_iters);
        CHK(hipStreamSynchronize(s0));
        auto t2 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(filler, dim3(fill_wgs), dim3(256), 0, c0, out, fill_iters);
        CHK(hipStreamSynchronize(c0));
        auto t3 = std::chrono::steady_clock::now();
        for (int l = 0; l < links; l++) hipLaunchKernelGGL(link, dim3(link_wgs), dim3(1024), 0, c1, out + 65536, link_iters);
        CHK(hipStreamSynchronize(c1));
        auto t4 = std::chrono::steady_clock::now();
        if (rep == 2)
            printf("alone: filler %.3f ms, chain %.3f ms; masked: filler (all but %d CUs) %.3f ms, chain (%d CUs) %.3f ms\n",
                   std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), reserve,
                   std::chrono::duration<double, std::milli>(t3 - t2).count(), reserve, std::chrono::duration<double, std::milli>(t4 - t3).count());
    }
    // Does USING the masked streams cost the other launches of the step anything?  A step-like sequence on the plain stream — ten
    // small dependent kernels — timed (i) before the masked streams have run anything here, (ii) with a fork to both masked
    // streams and a join back at the head of every sequence, (iii) the same fork / join through two PLAIN streams.
    {
        hipEvent_t ef, e1, e2;
        CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        auto seq = [&](const char* what, hipStream_t fa, hipStream_t fb) -> int {
            double best = 1e9;
            for (int rep = 0; rep < 20; rep++) {
                CHK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                if (fa) {
                    CHK(hipEventRecord(ef, s0));
                    CHK(hipStreamWaitEvent(fa, ef, 0));
                    CHK(hipStreamWaitEvent(fb, ef, 0));
                    hipLaunchKernelGGL(filler, dim3(64), dim3(256), 0, fa, out, 2000);
                    hipLaunchKernelGGL(link, dim3(4), dim3(1024), 0, fb, out + 65536, 2000);
                    CHK(hipEventRecord(e1, fa));
                    CHK(hipEventRecord(e2, fb));
                    CHK(hipStreamWaitEvent(s0, e1, 0));
                    CHK(hipStreamWaitEvent(s0, e2, 0));
                }
                for (int kx = 0; kx < 10; kx++) hipLaunchKernelGGL(filler, dim3(512), dim3(256), 0, s0, out, 2000);
                CHK(hipStreamSynchronize(s0));
                const auto t1 = std::chrono::steady_clock::now();
                best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count());
            }
            printf("ten small kernels on the plain stream, %-44s %.3f ms\n", what, best);
            return 0;
        };
        hipStream_t none = nullptr;
        if (seq("(i) nothing beside", none, none)) return 1;
        if (seq("(iii) fork / join through two plain streams", s1, s0 == s1 ? s0 : s1)) return 1;
        hipStream_t s2;
        CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        if (seq("(iii') fork / join through two plain streams", s1, s2)) return 1;
        if (seq("(ii) fork / join through the masked streams", c0, c1)) return 1;
        if (seq("(i') nothing beside, afterwards", none, none)) return 1;
    }
    if (run("(a) one stream", s0, s0)) return 1;
    if (run("(b) two plain streams", s0, s1)) return 1;
    if (run("(c) CU-masked streams", c0, c1)) return 1;
    if (run("(d) filler plain, chain masked", s0, c1)) return 1;
    return 0;
}
