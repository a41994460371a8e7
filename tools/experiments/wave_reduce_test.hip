// Standalone check of csrc/wave_reduce.h on real hardware (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/wave_reduce_test.hip -o /tmp/wrt && /tmp/wrt
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../gaussiansplatting.jl_amd/csrc/wave_reduce.h"

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

// row-then-column network: totals of the 9 accumulator-row entries against a plain host sum
__global__ void krc(const float* in, float* out, int* slot) {
    const int lane = threadIdx.x;
    gsr::LaneBits L(lane);
    gsr::RowColConsts K(lane);
    const float dx = 3.0f - (float)(lane & 15);
    out[lane] = gsr::wave_reduce_rowcol_rgb(in[lane], in[64 + lane], in[128 + lane], in[192 + lane], in[256 + lane],
                                            in[320 + lane], dx, L, K);
    slot[lane] = K.slot;
}

int run_rowcol() {
    std::vector<float> h(6 * 64);
    for (auto& x : h) x = (float)(rand() % 64) / 4.0f;  // small dyadic values: every sum below is exact in fp32
    float *din, *dout; int* dk;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dk, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    krc<<<1, 64>>>(din, dout, dk);
    float o[64]; int kk[64];
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    hipMemcpy(kk, dk, sizeof kk, hipMemcpyDeviceToHost);
    double want[9] = {0};
    for (int l = 0; l < 64; l++) {
        const double dx = 3.0 - (l & 15), P = h[l], U1 = h[64 + l], U2 = h[128 + l];
        want[0] += h[192 + l]; want[1] += h[256 + l]; want[2] += h[320 + l];
        want[3] += P; want[4] += dx * dx * P; want[5] += dx * U1; want[6] += U2; want[7] += dx * P; want[8] += U1;
    }
    int bad = 0; unsigned seen = 0;
    for (int l = 0; l < 64; l++) {
        if (kk[l] < 0) continue;
        if (kk[l] > 8 || (seen >> kk[l]) & 1u) { printf("rowcol: lane %d bad/duplicate slot %d\n", l, kk[l]); bad++; continue; }
        seen |= 1u << kk[l];
        if ((double)o[l] != want[kk[l]]) { printf("rowcol: lane %d slot %d got %f want %f\n", l, kk[l], o[l], want[kk[l]]); bad++; }
    }
    if (seen != 0x1FFu) { printf("rowcol: slots written %x, expected 1ff\n", seen); bad++; }
    printf("rowcol %s\n", bad ? "FAIL" : "ok");
    return bad;
}

// the same nine totals with the row stage on the matrix pipe (six chained v_mfma_f32_16x16x4_f32)
__global__ void krc_mfma(const float* in, float* out, int* slot) {
    const int lane = threadIdx.x;
    gsr::LaneBits L(lane);
    gsr::RowColConstsM K(lane);
    const float dx = 3.0f - (float)(lane & 15);
    out[lane] = gsr::wave_reduce_rowcol_rgb_mfma(in[lane], in[64 + lane], in[128 + lane], in[192 + lane], in[256 + lane],
                                                 in[320 + lane], dx, L, K);
    slot[lane] = K.slot;
}

int run_rowcol_mfma() {
    std::vector<float> h(6 * 64);
    for (auto& x : h) x = (float)(rand() % 64) / 4.0f;
    float *din, *dout; int* dk;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dk, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    krc_mfma<<<1, 64>>>(din, dout, dk);
    float o[64]; int kk[64];
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    hipMemcpy(kk, dk, sizeof kk, hipMemcpyDeviceToHost);
    double want[9] = {0};
    for (int l = 0; l < 64; l++) {
        const double dx = 3.0 - (l & 15), P = h[l], U1 = h[64 + l], U2 = h[128 + l];
        want[0] += h[192 + l]; want[1] += h[256 + l]; want[2] += h[320 + l];
        want[3] += P; want[4] += dx * dx * P; want[5] += dx * U1; want[6] += U2; want[7] += dx * P; want[8] += U1;
    }
    int bad = 0; unsigned seen = 0;
    for (int l = 0; l < 64; l++) {
        if (kk[l] < 0) continue;
        if (kk[l] > 8 || (seen >> kk[l]) & 1u) { printf("rowcol_mfma: lane %d bad/duplicate slot %d\n", l, kk[l]); bad++; continue; }
        seen |= 1u << kk[l];
        if ((double)o[l] != want[kk[l]]) { printf("rowcol_mfma: lane %d slot %d got %f want %f\n", l, kk[l], o[l], want[kk[l]]); bad++; }
    }
    if (seen != 0x1FFu) { printf("rowcol_mfma: slots written %x, expected 1ff\n", seen); bad++; }
    printf("rowcol_mfma %s\n", bad ? "FAIL" : "ok");
    return bad;
}

int main() { if (run_rowcol()) return 1; if (run_rowcol_mfma()) return 1; int b = run<9>() + run<10>() + run<13>() + run<1>() + run<16>() + run<5>(); return b != 0; }
