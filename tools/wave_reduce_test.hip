// Standalone check of csrc/wave_reduce.h on real hardware (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/wave_reduce_test.hip -o /tmp/wrt && /tmp/wrt
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gaussiansplatting.jl_amd/csrc/wave_reduce.h"

template <int N>
__global__ void k(const float* in, float* out, int* kidx) {
    const int lane = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = i < N ? in[i * 64 + lane] : 0.0f;
    gsr::LaneBits L(lane);
    out[lane] = gsr::wave_reduce_transposed<N>(v, L);
    kidx[lane] = gsr::wave_reduce_index<N>(lane);
}

template <int N>
int run() {
    std::vector<float> h(16 * 64);
    for (auto& x : h) x = (float)(rand() % 1000) / 8.0f;  // exactly representable sums
    float *din, *dout; int* dk;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dk, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<N><<<1, 64>>>(din, dout, dk);
    float o[64]; int kk[64];
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    hipMemcpy(kk, dk, sizeof kk, hipMemcpyDeviceToHost);
    int bad = 0; unsigned seen = 0;
    for (int l = 0; l < 64; l++) {
        if ((l & 6) != 4) continue;  // wave_reduce_writer
        float s = 0; for (int j = 0; j < 64; j++) s += h[kk[l] * 64 + j];
        if (kk[l] < 0 || kk[l] >= N || s != o[l]) { bad++; if (bad < 5) printf("N=%d lane %d k=%d got %f want %f\n", N, l, kk[l], o[l], s); }
        else seen |= 1u << kk[l];
    }
    if (seen != (1u << N) - 1) { printf("N=%d: not every input reachable (%x)\n", N, seen); bad++; }
    printf("N=%d %s\n", N, bad ? "FAIL" : "ok");
    return bad;
}

int main() { int b = run<9>() + run<10>() + run<13>() + run<1>() + run<16>() + run<5>(); return b != 0; }
