// What does FETCH_SIZE count for the access patterns of the rasterizer's kernels?  MI355X_MICROARCH.md §HBM: on gfx950 the
// counter reads exactly 1/2 of a wide coalesced stream (128-byte requests tallied at 64 bytes) and "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  profiles/pmc_traffic.json scales every
// kernel with the factor of a 1 GiB streaming copy; the fused forward and the per-Gaussian kernels GATHER 64-byte records.
// Kernels (each over a 2 GiB buffer, beyond L2 + Infinity Cache; run under rocprofv3 --pmc FETCH_SIZE --kernel-trace and
// compare counter / launch with the bytes each kernel names; the durations say what the memory system really moved):
//   stream16     every lane one float4, consecutive                       1 GiB read
//   gather64     every lane the 4 float4 of ONE random 64-byte record      8 Mi records = 512 MiB useful
//   gather128    every lane the 8 float4 of ONE random 128-byte line       4 Mi lines   = 512 MiB useful
//   gather64x2   every lane 64 bytes at a random 128-byte line's start     8 Mi records = 512 MiB useful, 1 GiB of lines
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ src, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = src[i];
    if (v.x == 123.456f) out[0] = v.y + v.z + v.w;
}
// REC = bytes read per lane, STRIDE = alignment / spacing of the records in the buffer
template <int REC, int STRIDE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ src, float* __restrict__ out, uint32_t n_slots, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = mix(i * 2654435761u + 12345u) % n_slots;
    const float4* p = src + (size_t)slot * (STRIDE / 16);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < REC / 16; k++) { const float4 v = p[k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit = [&](const char* name, double useful, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-12s %8.3f ms   useful %.0f MiB -> %.2f TB/s useful\n", name, best, useful / 1048576.0, useful / best * 1e-9);
    };
    const size_t n4 = ((size_t)1 << 30) / 16;
    timeit("stream16", 1 << 30, [&] { stream16<<<(unsigned)(n4 / 256), 256>>>(buf, out, n4); });
    const uint32_t n64 = 8u << 20, n128 = 4u << 20;
    timeit("gather64", 512.0 * 1048576, [&] { gather<64, 64><<<n64 / 256, 256>>>(buf, out, (uint32_t)(bytes / 64), n64); });
    timeit("gather128", 512.0 * 1048576, [&] { gather<128, 128><<<n128 / 256, 256>>>(buf, out, (uint32_t)(bytes / 128), n128); });
    timeit("gather64x2", 512.0 * 1048576, [&] { gather<64, 128><<<n64 / 256, 256>>>(buf, out, (uint32_t)(bytes / 128), n64); });
    timeit("gather32", 256.0 * 1048576, [&] { gather<32, 32><<<n64 / 256, 256>>>(buf, out, (uint32_t)(bytes / 32), n64); });
    timeit("gather16", 128.0 * 1048576, [&] { gather<16, 16><<<n64 / 256, 256>>>(buf, out, (uint32_t)(bytes / 16), n64); });
    return 0;
}
