// Can the tile counters be sharded per XCD and bumped with L2-local (workgroup-scope) atomics?
// 1. rate of returning 64-bit atomics at agent scope vs workgroup scope (sc1 off) on per-XCD copies of 4 080 words;
// 2. are the workgroup-scope ones still atomic among the workgroups of ONE XCD (they share its L2): every copy must end
//    at exactly the number of atomics issued to it, and the returned values of a word must be a permutation (checked
//    through sum and sum of squares).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_scope.hip -o tools/bin/atomic_scope && tools/bin/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) & 0xF; }

template <int SCOPE>
__global__ __launch_bounds__(256) void k(unsigned long long* words, uint32_t n_words, unsigned long long* issued /* [8][n_words] */,
                                         unsigned long long* sum_old /* [8][n_words] */, int per_thread) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t x = SCOPE == __HIP_MEMORY_SCOPE_AGENT ? 0u : xcc_id();
    for (int r = 0; r < per_thread; r++) {
        const uint32_t w = mix(tid * 977u + (uint32_t)r * 0x9e3779b9u) % n_words;
        unsigned long long* p = words + (size_t)x * n_words + w;
        const unsigned long long old = __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, SCOPE);
        if (issued) {  // bookkeeping with ordinary device atomics (not timed)
            atomicAdd(issued + (size_t)x * n_words + w, 1ull);
            atomicAdd(sum_old + (size_t)x * n_words + w, old);
        }
    }
}

int main() {
    const uint32_t n_words = 4080;
    const int threads = 1'000'000 / 256 * 256, per_thread = 3;
    unsigned long long *words, *issued, *sum_old;
    const size_t bytes = (size_t)8 * n_words * 8;
    hipMalloc(&words, bytes); hipMalloc(&issued, bytes); hipMalloc(&sum_old, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int scope, bool check) {
        hipMemset(words, 0, bytes); hipMemset(issued, 0, bytes); hipMemset(sum_old, 0, bytes);
        float ms = 0;
        for (int rep = 0; rep < (check ? 1 : 3); rep++) {
            if (!check) hipMemset(words, 0, bytes);
            hipEventRecord(a);
            if (scope == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<threads / 256, 256>>>(words, n_words, check ? issued : nullptr, sum_old, per_thread);
            else k<__HIP_MEMORY_SCOPE_WORKGROUP><<<threads / 256, 256>>>(words, n_words, check ? issued : nullptr, sum_old, per_thread);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        if (!check) { printf("%s scope: %d atomics in %.3f ms = %.1f G/s\n", scope ? "workgroup (per-XCD copies)" : "agent", threads * per_thread, ms, threads * (double)per_thread / ms * 1e-6); return; }
        std::vector<unsigned long long> w(8 * n_words), is(8 * n_words), so(8 * n_words);
        hipMemcpy(w.data(), words, bytes, hipMemcpyDeviceToHost); hipMemcpy(is.data(), issued, bytes, hipMemcpyDeviceToHost);
        hipMemcpy(so.data(), sum_old, bytes, hipMemcpyDeviceToHost);
        unsigned long long lost = 0, dup = 0, total = 0; int copies_used = 0;
        for (int x = 0; x < 8; x++) {
            unsigned long long cx = 0;
            for (uint32_t i = 0; i < n_words; i++) {
                const size_t j = (size_t)x * n_words + i;
                cx += is[j]; total += is[j];
                if (w[j] != is[j]) lost += is[j] > w[j] ? is[j] - w[j] : w[j] - is[j];
                if (so[j] != is[j] * (is[j] - 1) / 2) dup++;  // returned values 0..c-1 exactly once each
            }
            if (cx) copies_used++;
        }
        printf("%s scope check: %llu atomics over %d copies, lost updates %llu, words with duplicate / missing returns %llu\n",
               scope ? "workgroup" : "agent", total, copies_used, lost, dup);
    };
    run(0, false); run(1, false); run(0, true); run(1, true);
    return 0;
}
