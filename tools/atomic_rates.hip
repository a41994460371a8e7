// Aggregate rate of returning device-scope 64-bit atomics spread over W words at a given stride (bytes), on gfx950:
// is preprocess's 28 G atomics/s the chip's rate, or a property of 4 080 counters packed into 32 KB?
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_rates.hip -o tools/bin/atomic_rates && tools/bin/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// lanes of a wave on CONSECUTIVE words (what a spatially ordered scene would give the binning atomics)
__global__ __launch_bounds__(256) void atomics_coalesced_kernel(unsigned long long* words, uint32_t n_words, int rounds,
                                                                unsigned long long* sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; r++) {
        const uint32_t base = mix((tid >> 6) * 977u + (uint32_t)r * 0x9e3779b9u) % (n_words - 64u);
        acc += atomicAdd(words + base + (tid & 63u), 1ull);
    }
    if (acc == 0xdeadbeefcafeull) sink[0] = acc;
}


// lanes of a wave on L random 128-byte LINES of the counter array, random words inside them: what grouping a workgroup's
// requests by cache line (a counting sort in LDS) would give the binning atomics
template <int L>
__global__ __launch_bounds__(256) void atomics_lines_kernel(unsigned long long* words, uint32_t n_words, int rounds,
                                                            unsigned long long* sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, lane = tid & 63u;
    const uint32_t n_lines = n_words / 16u;
    unsigned long long acc = 0, old[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t line = mix((tid >> 6) * 977u + (uint32_t)r * 0x9e3779b9u + (lane % L) * 0x85ebca6bu) % n_lines;
        const uint32_t w = line * 16u + (mix(tid * 31u + r) & 15u);
        old[r] = atomicAdd(words + w, 1ull);
    }
    acc = old[0] + old[1] + old[2];
    if (acc == 0xdeadbeefcafeull) sink[0] = acc;
}

template <int INFLIGHT>
__global__ __launch_bounds__(256) void atomics_kernel(unsigned long long* words, uint32_t n_words, uint32_t stride_words,
                                                      int rounds, unsigned long long* sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; r++) {
        unsigned long long old[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) {
            const uint32_t w = mix(tid * 977u + (uint32_t)(r * INFLIGHT + k) * 0x9e3779b9u) % n_words;
            old[k] = atomicAdd(words + (size_t)w * stride_words, 1ull);
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += old[k];
    }
    if (acc == 0xdeadbeefcafeull) sink[0] = acc;
}

int main() {
    const uint32_t n_words = 4080;
    const int threads = 1'000'000 / 256 * 256, rounds = 1, inflight = 3;  // ~ the per-Gaussian atomics of config 3 (2.5-3 M)
    unsigned long long *buf, *sink;
    const size_t max_bytes = (size_t)n_words * 4096 + 4096;
    hipMalloc(&buf, max_bytes); hipMalloc(&sink, 8);
    hipMemset(buf, 0, max_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t stride_bytes : {8u, 64u, 128u, 256u, 1024u, 4096u}) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            atomics_kernel<3><<<threads / 256, 256>>>(buf, n_words, stride_bytes / 8, rounds, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("words %u stride %5u B: %d atomics in %.3f ms = %.1f G/s\n", n_words, stride_bytes, threads * rounds * inflight, ms,
                                 threads * (double)rounds * inflight / ms * 1e-6);
        }
    }
    for (uint32_t nw : {512u, 65536u, 1048576u}) {
        const uint32_t stride_bytes = 8;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            atomics_kernel<3><<<threads / 256, 256>>>(buf, nw, 1, rounds, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("words %u stride %5u B: %.3f ms = %.1f G/s\n", nw, stride_bytes, ms, threads * (double)rounds * inflight / ms * 1e-6);
        }
    }
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        atomics_coalesced_kernel<<<threads / 256, 256>>>(buf, 4080, 3, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("words 4080, lanes of a wave on 64 consecutive words: %.3f ms = %.1f G/s\n", ms, threads * 3.0 / ms * 1e-6);
    }
#define LINES(Lv)                                                                                                       \
    for (int rep = 0; rep < 3; rep++) {                                                                                 \
        hipEventRecord(a);                                                                                              \
        atomics_lines_kernel<Lv><<<threads / 256, 256>>>(buf, 4080, 1, sink);                                           \
        hipEventRecord(b); hipEventSynchronize(b);                                                                      \
        float ms; hipEventElapsedTime(&ms, a, b);                                                                       \
        if (rep == 2) printf("words 4080, lanes of a wave on %d random lines (random words inside): %.3f ms = %.1f G/s\n", Lv, ms, threads * 3.0 / ms * 1e-6); \
    }
    LINES(4) LINES(8) LINES(16) LINES(32) LINES(64)
    return 0;
}
