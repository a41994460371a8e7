// The per-Gaussian streaming passes either side of rasterize(): the functor prologue
// (rasterizer.jl:200-253: hcat of the SH blocks, sigmoid, exp), its pullback, and the Adam
// update NU.step! applies to the six parameter arrays (training.jl:234-239,778).
// All three are pure HBM streams: one thread per element / Gaussian, float4 where the layout
// allows, no FMA contraction (compiled with -ffp-contract=off) so the update is the same fp32
// expression tree as the oracle's.
#include "gsr_kernels.h"
#include "adam_math.h"

namespace {

using gsr::AdamHyper;
using gsr::adam_update;
using gsr::sigmoidf_;

// shs (3,K,N) <- [sh_color (3,1,N) | sh_remainder (3,K-1,N)]: one thread per output float, coalesced writes
__global__ __launch_bounds__(256) void prologue_shs_kernel(size_t total, int K3, const float* __restrict__ dc,
                                                           const float* __restrict__ rest, float* __restrict__ shs) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t i = e / (size_t)K3;
    const int j = (int)(e - i * (size_t)K3);
    shs[e] = j < 3 ? dc[3 * i + j] : rest[(size_t)(K3 - 3) * i + (j - 3)];
}
// K3 % 4 == 0 (SH degree 1 and 3): one float4 of shs per thread — dwordx4 stores, the four source
// floats are consecutive in (dc | rest) order so the dword loads of a wave stay contiguous
template <bool BWD>
__global__ __launch_bounds__(256) void prologue_shs4_kernel(size_t total4, int K3, float* __restrict__ dc,
                                                            float* __restrict__ rest, float* __restrict__ shs) {
    const size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e4 >= total4) return;
    const int q = K3 / 4;
    const size_t i = e4 / (size_t)q;
    const int j = 4 * (int)(e4 - i * (size_t)q);
    float* const d = dc + 3 * i;
    float* const r = rest + (size_t)(K3 - 3) * i - 3;  // r[j] is element j of the Gaussian's row for j >= 3
    float4* const out = (float4*)shs + e4;
    if (!BWD) {
        float4 v;
        if (j == 0) v = make_float4(d[0], d[1], d[2], r[3]);
        else v = make_float4(r[j], r[j + 1], r[j + 2], r[j + 3]);
        *out = v;
    } else {
        const float4 v = *out;
        if (j == 0) { d[0] = v.x; d[1] = v.y; d[2] = v.z; r[3] = v.w; }
        else { r[j] = v.x; r[j + 1] = v.y; r[j + 2] = v.z; r[j + 3] = v.w; }
    }
}
__global__ __launch_bounds__(256) void prologue_act_kernel(int n, int scale_dims, const float* __restrict__ opac,
                                                           const float* __restrict__ scales,
                                                           float* __restrict__ opac_act, float* __restrict__ scales_act) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    opac_act[i] = sigmoidf_(opac[i]);
    if (scale_dims == 1) {
        const float s = expf(scales[i]);  // isotropic: vcat(s, s, s) (rasterizer.jl:235-247)
        scales_act[3 * (size_t)i] = s; scales_act[3 * (size_t)i + 1] = s; scales_act[3 * (size_t)i + 2] = s;
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) scales_act[3 * (size_t)i + c] = expf(scales[3 * (size_t)i + c]);
    }
}

__global__ __launch_bounds__(256) void prologue_shs_bwd_kernel(size_t total, int K3, const float* __restrict__ vshs,
                                                               float* __restrict__ vdc, float* __restrict__ vrest) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t i = e / (size_t)K3;
    const int j = (int)(e - i * (size_t)K3);
    const float v = vshs[e];
    if (j < 3) vdc[3 * i + j] = v;
    else vrest[(size_t)(K3 - 3) * i + (j - 3)] = v;
}
__global__ __launch_bounds__(256) void prologue_act_bwd_kernel(int n, int scale_dims,
                                                               const float* __restrict__ opac_act,
                                                               const float* __restrict__ scales_act,
                                                               const float* __restrict__ vopac_act,
                                                               const float* __restrict__ vscales_act,
                                                               float* __restrict__ vopac, float* __restrict__ vscales) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = opac_act[i];
    vopac[i] = vopac_act[i] * (a * (1.0f - a));
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = vscales_act[3 * (size_t)i + c] * scales_act[3 * (size_t)i + c];
    if (scale_dims == 1) vscales[i] = (g[0] + g[1]) + g[2];
    else {
#pragma unroll
        for (int c = 0; c < 3; c++) vscales[3 * (size_t)i + c] = g[c];
    }
}

// STREAM triad a = b + q*c over float4: the measured HBM ceiling bench.py quotes next to the
// 8 TB/s data-sheet figure
__global__ __launch_bounds__(256) void triad_kernel(size_t n4, float4* __restrict__ a, const float4* __restrict__ b,
                                                    const float4* __restrict__ c, float q) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + q * y.x, x.y + q * y.y, x.z + q * y.z, x.w + q * y.w);
    }
}

struct AdamGroups {
    float* theta[GSR_ADAM_MAX_GROUPS];
    const float* grad[GSR_ADAM_MAX_GROUPS];
    float* mu[GSR_ADAM_MAX_GROUPS];
    float* nu[GSR_ADAM_MAX_GROUPS];
    long long count[GSR_ADAM_MAX_GROUPS];
    float lr_t[GSR_ADAM_MAX_GROUPS];  // lr · sqrt(1-β2^t) / (1-β1^t), evaluated on the host
    long long block_start[GSR_ADAM_MAX_GROUPS + 1];  // first workgroup of each group
    int n;
};

// One launch for all parameter groups: a workgroup belongs to exactly one group (block_start),
// 4 elements per thread through float4 when the group's pointers are 16-byte aligned.
__global__ __launch_bounds__(256) void adam_kernel(AdamGroups G, float beta1, float beta2, float eps) {
    int g = 0;
#pragma unroll
    for (int k = 1; k < GSR_ADAM_MAX_GROUPS; k++)
        if (k < G.n && (long long)blockIdx.x >= G.block_start[k]) g = k;
    const long long base = (((long long)blockIdx.x - G.block_start[g]) * 256 + threadIdx.x) * 4;
    const long long count = G.count[g];
    if (base >= count) return;
    float* __restrict__ th = G.theta[g];
    const float* __restrict__ gr = G.grad[g];
    float* __restrict__ mu = G.mu[g];
    float* __restrict__ nu = G.nu[g];
    const float lr_t = G.lr_t[g], omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    const bool vec = base + 4 <= count &&
                     ((((uintptr_t)th | (uintptr_t)gr | (uintptr_t)mu | (uintptr_t)nu) & 15) == 0);
    float t[4], d[4], m[4], v[4];
    if (vec) {
        const float4 t4 = *(const float4*)(th + base), d4 = *(const float4*)(gr + base);
        const float4 m4 = *(const float4*)(mu + base), v4 = *(const float4*)(nu + base);
        t[0] = t4.x; t[1] = t4.y; t[2] = t4.z; t[3] = t4.w;
        d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        m[0] = m4.x; m[1] = m4.y; m[2] = m4.z; m[3] = m4.w;
        v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = base + k < count;
            t[k] = ok ? th[base + k] : 0.0f; d[k] = ok ? gr[base + k] : 0.0f;
            m[k] = ok ? mu[base + k] : 0.0f; v[k] = ok ? nu[base + k] : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        m[k] = beta1 * m[k] + omb1 * d[k];
        v[k] = beta2 * v[k] + omb2 * (d[k] * d[k]);
        t[k] = t[k] - lr_t * m[k] / (sqrtf(v[k]) + eps);
    }
    if (vec) {
        *(float4*)(th + base) = make_float4(t[0], t[1], t[2], t[3]);
        *(float4*)(mu + base) = make_float4(m[0], m[1], m[2], m[3]);
        *(float4*)(nu + base) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k < count) { th[base + k] = t[k]; mu[base + k] = m[k]; nu[base + k] = v[k]; }
    }
}

// ---- fused trainer tail: prologue pullback + Adam + prologue of the next step ----
// One pass over the parameters instead of three (gsr_prologue_backward, gsr_adam_step,
// gsr_prologue_forward): the gradients arrive w.r.t. the ACTIVATED values (what gsr_backward
// writes), the chain rule of the prologue is applied in registers, the six NU.Adam states are
// advanced, and the activated copies the next rasterize() consumes are written straight away.
// Same fp32 expression trees as the three separate kernels (bit-identical θ, μ, ν).
// (AdamHyper, adam_update and the per-Gaussian block live in adam_math.h, shared with pergauss.hip)
// SH block: one thread per coefficient float of shs (N x 3K); element j of a Gaussian's row
// belongs to sh_color (j < 3) or sh_remainder — two optimizers, two learning rates.
__global__ __launch_bounds__(256) void tail_sh_kernel(size_t total, int K3, const float* __restrict__ vshs,
                                                      float* __restrict__ dc, float* __restrict__ dc_mu,
                                                      float* __restrict__ dc_nu, float* __restrict__ rest,
                                                      float* __restrict__ rest_mu, float* __restrict__ rest_nu,
                                                      AdamHyper h_dc, AdamHyper h_rest, float* __restrict__ shs) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t i = e / (size_t)K3;
    const int j = (int)(e - i * (size_t)K3);
    const bool is_dc = j < 3;
    const size_t k = is_dc ? 3 * i + j : (size_t)(K3 - 3) * i + (j - 3);
    float* th = is_dc ? dc : rest;
    float* mu = is_dc ? dc_mu : rest_mu;
    float* nu = is_dc ? dc_nu : rest_nu;
    float m = mu[k], v = nu[k];
    const float t = adam_update(th[k], vshs[e], m, v, is_dc ? h_dc : h_rest);
    th[k] = t; mu[k] = m; nu[k] = v;
    shs[e] = t;  // hcat(sh_color, sh_remainder) of the next forward
}

// per-Gaussian block: points (3), opacity logit (1), log-scales (3 or 1), rotations (4)
__global__ __launch_bounds__(256) void tail_gauss_kernel(int n, gsr::TailState S, const float* __restrict__ vmeans,
                                                         const float* __restrict__ vopac_act,
                                                         const float* __restrict__ vscales_act,
                                                         const float* __restrict__ vrot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float vm[3] = {vmeans[3 * (size_t)i], vmeans[3 * (size_t)i + 1], vmeans[3 * (size_t)i + 2]};
    const float vs[3] = {vscales_act[3 * (size_t)i], vscales_act[3 * (size_t)i + 1], vscales_act[3 * (size_t)i + 2]};
    const float vq[4] = {vrot[4 * (size_t)i], vrot[4 * (size_t)i + 1], vrot[4 * (size_t)i + 2], vrot[4 * (size_t)i + 3]};
    gsr::tail_gauss_apply(S, i, vm, vopac_act[i], vs, vq);
}

// ---- boolean-mask compaction: findall(mask) and x[:, idxs] (densification.jl:138-191,279-288) ----
// findall in three small passes: per-1024 popcounts, one-workgroup scan of the block counts,
// per-block ordered emit (ballot + popcount ranks).  Order-preserving, as Julia's logical indexing.
constexpr int FA_BLOCK = 1024;

__global__ __launch_bounds__(256) void findall_count_kernel(long long n, const uint8_t* __restrict__ mask,
                                                            uint32_t* __restrict__ block_count) {
    __shared__ uint32_t wsum[4];
    const long long base = (long long)blockIdx.x * FA_BLOCK;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < FA_BLOCK / 256; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(i < n && mask[i] != 0)) ;
    }
    // every lane of a wave holds the wave's count
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void findall_scan_kernel(int nb, uint32_t* __restrict__ block_count /* in: counts, out: exclusive offsets */,
                                                            uint32_t* __restrict__ total) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + tid;
        const uint32_t v = i < nb ? block_count[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wave_sums[w];
        const uint32_t excl = carry_s + woff + x - v;
        if (i < nb) block_count[i] = excl;
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void findall_emit_kernel(long long n, const uint8_t* __restrict__ mask,
                                                           const uint32_t* __restrict__ block_off,
                                                           uint32_t* __restrict__ indices) {
    __shared__ uint32_t wcount[FA_BLOCK / 64];
    const long long base = (long long)blockIdx.x * FA_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool sel[FA_BLOCK / 256];
    unsigned long long bal[FA_BLOCK / 256];
#pragma unroll
    for (int k = 0; k < FA_BLOCK / 256; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        sel[k] = i < n && mask[i] != 0;
        bal[k] = __builtin_amdgcn_ballot_w64(sel[k]);
        if (lane == 0) wcount[k * 4 + wave] = (uint32_t)__popcll(bal[k]);
    }
    __syncthreads();
    const uint32_t off0 = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < FA_BLOCK / 256; k++) {
        uint32_t before = 0;
        for (int w = 0; w < k * 4 + wave; w++) before += wcount[w];  // chunks of 64 elements in index order
        if (sel[k]) {
            const uint32_t rank = (uint32_t)__popcll(bal[k] & ((1ull << lane) - 1ull));
            indices[off0 + before + rank] = (uint32_t)(base + k * 256 + threadIdx.x);
        }
    }
}

struct GatherGroups {
    const uint32_t* src[GSR_ADAM_MAX_GROUPS];
    uint32_t* dst[GSR_ADAM_MAX_GROUPS];
    int row_words[GSR_ADAM_MAX_GROUPS];
    long long block_start[GSR_ADAM_MAX_GROUPS + 1];
    int n;
};
// dst[r, :] = src[idx[r], :] for 4-byte words; one thread per output word (coalesced stores, the
// loads of one row are contiguous)
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherGroups G, const uint32_t* __restrict__ idx, long long count) {
    int g = 0;
#pragma unroll
    for (int k = 1; k < GSR_ADAM_MAX_GROUPS; k++)
        if (k < G.n && (long long)blockIdx.x >= G.block_start[k]) g = k;
    const long long e = ((long long)blockIdx.x - G.block_start[g]) * 256 + threadIdx.x;
    const int rw = G.row_words[g];
    if (e >= count * rw) return;
    const long long r = e / rw;
    const int j = (int)(e - r * rw);
    G.dst[g][e] = G.src[g][(long long)idx[r] * rw + j];
}

}  // namespace

void gsr_launch_prologue_fwd(hipStream_t s, int n, int k_rest, int scale_dims, const float* sh_color,
                             const float* sh_remainder, const float* opacities, const float* scales, float* shs,
                             float* opacities_act, float* scales_act) {
    if (n <= 0) return;
    const int K3 = 3 * (1 + k_rest);
    const size_t total = (size_t)n * K3;
    if (K3 % 4 == 0 && K3 > 4)
        hipLaunchKernelGGL(prologue_shs4_kernel<false>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s,
                           total / 4, K3, const_cast<float*>(sh_color), const_cast<float*>(sh_remainder), shs);
    else
        hipLaunchKernelGGL(prologue_shs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, K3,
                           sh_color, sh_remainder, shs);
    hipLaunchKernelGGL(prologue_act_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, scale_dims, opacities, scales,
                       opacities_act, scales_act);
}

void gsr_launch_prologue_bwd(hipStream_t s, int n, int k_rest, int scale_dims, const float* opacities_act,
                             const float* scales_act, const float* vshs, const float* vopacities_act,
                             const float* vscales_act, float* v_sh_color, float* v_sh_remainder, float* v_opacities,
                             float* v_scales) {
    if (n <= 0) return;
    const int K3 = 3 * (1 + k_rest);
    const size_t total = (size_t)n * K3;
    if (K3 % 4 == 0 && K3 > 4)
        hipLaunchKernelGGL(prologue_shs4_kernel<true>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s,
                           total / 4, K3, v_sh_color, v_sh_remainder, const_cast<float*>(vshs));
    else
        hipLaunchKernelGGL(prologue_shs_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, K3,
                           vshs, v_sh_color, v_sh_remainder);
    hipLaunchKernelGGL(prologue_act_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, scale_dims, opacities_act,
                       scales_act, vopacities_act, vscales_act, v_opacities, v_scales);
}

void gsr_launch_adam(hipStream_t s, int n_groups, float* const* theta, const float* const* grad, float* const* mu,
                     float* const* nu, const long long* count, const float* lr_t, float beta1, float beta2, float eps) {
    AdamGroups G;
    G.n = n_groups;
    long long blocks = 0;
    for (int g = 0; g < GSR_ADAM_MAX_GROUPS; g++) {
        const bool on = g < n_groups;
        G.theta[g] = on ? theta[g] : nullptr; G.grad[g] = on ? grad[g] : nullptr;
        G.mu[g] = on ? mu[g] : nullptr; G.nu[g] = on ? nu[g] : nullptr;
        G.count[g] = on ? count[g] : 0; G.lr_t[g] = on ? lr_t[g] : 0.0f;
        G.block_start[g] = blocks;
        if (on) blocks += (count[g] + 1023) / 1024;
    }
    G.block_start[GSR_ADAM_MAX_GROUPS] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, beta1, beta2, eps);
}

void gsr_launch_triad(hipStream_t s, size_t n4, float* a, const float* b, const float* c, float q) {
    if (n4 == 0) return;
    const size_t blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(triad_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, n4, (float4*)a,
                       (const float4*)b, (const float4*)c, q);
}

size_t gsr_findall_scratch_bytes(long long n) { return (size_t)((n + FA_BLOCK - 1) / FA_BLOCK + 1) * sizeof(uint32_t); }

void gsr_launch_findall(hipStream_t s, long long n, const uint8_t* mask, uint32_t* indices, uint32_t* count_dev,
                        uint32_t* scratch) {
    const int nb = (int)((n + FA_BLOCK - 1) / FA_BLOCK);
    if (nb == 0) { (void)hipMemsetAsync(count_dev, 0, 4, s); return; }
    hipLaunchKernelGGL(findall_count_kernel, dim3(nb), dim3(256), 0, s, n, mask, scratch);
    hipLaunchKernelGGL(findall_scan_kernel, dim3(1), dim3(1024), 0, s, nb, scratch, count_dev);
    hipLaunchKernelGGL(findall_emit_kernel, dim3(nb), dim3(256), 0, s, n, mask, scratch, indices);
}

void gsr_launch_gather_rows(hipStream_t s, int n_groups, const void* const* src, void* const* dst, const int* row_words,
                            const uint32_t* idx, long long count) {
    GatherGroups G;
    G.n = n_groups;
    long long blocks = 0;
    for (int g = 0; g < GSR_ADAM_MAX_GROUPS; g++) {
        const bool on = g < n_groups;
        G.src[g] = on ? (const uint32_t*)src[g] : nullptr;
        G.dst[g] = on ? (uint32_t*)dst[g] : nullptr;
        G.row_words[g] = on ? row_words[g] : 1;
        G.block_start[g] = blocks;
        if (on) blocks += (count * row_words[g] + 255) / 256;
    }
    G.block_start[GSR_ADAM_MAX_GROUPS] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, idx, count);
}

void gsr_launch_trainer_tail(hipStream_t s, int n, int k_rest, int scale_dims, const float* const* grads /* vmeans, vshs, vopac_act, vscales_act, vrot */,
                             float* const* theta /* points, dc, rest, opac, scales, rots */, float* const* mu,
                             float* const* nu, const float* lr_t /* [6] */, float beta1, float beta2, float eps,
                             float* shs, float* opac_act, float* scales_act) {
    if (n <= 0) return;
    AdamHyper h[6];
    for (int g = 0; g < 6; g++) h[g] = AdamHyper{lr_t[g], beta1, beta2, 1.0f - beta1, 1.0f - beta2, eps};
    const int K3 = 3 * (1 + k_rest);
    const size_t total = (size_t)n * K3;
    hipLaunchKernelGGL(tail_sh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, K3, grads[1],
                       theta[1], mu[1], nu[1], theta[2], mu[2], nu[2], h[1], h[2], shs);
    const gsr::TailState S = gsr_make_tail_state(theta, mu, nu, lr_t, beta1, beta2, eps, scale_dims, shs, opac_act, scales_act);
    hipLaunchKernelGGL(tail_gauss_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, S, grads[0], grads[2], grads[3],
                       grads[4]);
}
