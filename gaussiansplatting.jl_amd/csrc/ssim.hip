// Fused SSIM forward / backward and the photometric loss head.
// Reference behaviour: src/fused_ssim.jl:34-371 (kernels), :373-424 (host wrappers),
// src/training.jl:656,684-694 (L = (1-λ)·L1 + λ·(1 - mean SSIM)); SURVEY.md A.12.
//
// 16x16 output tile + 5-pixel halo staged in LDS, separable 11-tap Gaussian:
// horizontal pass into LDS, vertical pass in registers.  Accumulation order follows the
// reference (symmetric pairs d = 1..5 with weight GAUSS[5-d], centre tap last).  grid.z enumerates
// (channel, batch) planes for the generic entry points; the loss head loops the three channels inside
// one workgroup, as the reference does.
//
// This file is compiled TWICE into the library (csrc/Makefile):
//   SSIM_EXACT = 1, -ffp-contract=off: every fp32 operation as written, IEEE divisions — the maps are
//       bit-reproducible against the CPU oracle (the oracle's twin; gsr_ssim_precision(1) selects it);
//   SSIM_EXACT = 0 (the default path), -ffp-contract=fast: the same expressions with the multiply-adds fused
//       and the six divisions of the SSIM formula replaced by two hardware reciprocals — what any GPU compiler
//       makes of the reference's source ("contraction order unspecified, LLVM may fuse FMAs");
//       parity at the stated fp32 tolerance instead of bit for bit; −46 % VALU instructions per pixel.
#include "gsr_kernels.h"

#ifndef SSIM_EXACT
#define SSIM_EXACT 1
#endif
#if SSIM_EXACT
#define SSIM_NAME(x) x##_exact
#else
#define SSIM_NAME(x) x##_fast
#endif

namespace {

constexpr int HALO = 5;
constexpr int SH_DIM = GSR_TILE + 2 * HALO;  // 26
// LDS row strides (floats): a ds_read_b32 is served in two groups of 32 lanes = two tile rows of
// 16 lanes, so a row stride of 16 (mod 32) puts the two rows on disjoint banks.
constexpr int IN_STRIDE = 48, HC_STRIDE = 16;

__constant__ float GAUSS[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                0.0075987582094967365f, 0.001028380123898387f};

// (W,H,CH,B) planar arrays, x fastest (fused_ssim.jl:27-31)
struct PlanarSrc {
    const float* img;
    const float* ref;
    int W, H;
    __device__ __forceinline__ float x(int gx, int gy, int plane) const {
        return img[(size_t)gx + (size_t)W * gy + (size_t)W * H * plane];
    }
    __device__ __forceinline__ float y(int gx, int gy, int plane) const {
        return ref[(size_t)gx + (size_t)W * gy + (size_t)W * H * plane];
    }
};
// rasterizer output (C,W,H) channel-fastest vs target (W,H,3): folds
// `features[1:3,:,:]` + `permutedims` (training.jl:656,684-685) into the loads
struct RasterSrc {
    const float* image;
    const float* target;
    int W, H, C;
    __device__ __forceinline__ float x(int gx, int gy, int plane) const {
        return image[(size_t)C * ((size_t)gx + (size_t)W * gy) + plane];
    }
    __device__ __forceinline__ float y(int gx, int gy, int plane) const {
        return target[(size_t)gx + (size_t)W * gy + (size_t)W * H * plane];
    }
};

// Workgroup id -> tile, XCD-aware: workgroups are dealt round-robin to the 8 XCDs (one L2 each).
// Tiles that share halo rows and columns should share an L2, so XCD x gets the x-th contiguous
// eighth of the tiles (raster order) instead of every 8th tile.  Grid: (8 * ceil(tiles / 8), 1, planes).
__device__ __forceinline__ bool ssim_tile_of_block(int W, int H, int& tx0, int& ty0) {
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    const int n = gx * gy, per = (n + 7) / 8;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= n) return false;
    tx0 = (t % gx) * GSR_TILE;
    ty0 = (t / gx) * GSR_TILE;
    return true;
}

__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// fused_ssim.jl:34-238.  LOSS: additionally reduce Σ|x-y| and Σssim into partial[0..1].
// NCH = channel planes handled per workgroup (the reference loops channels inside the
// workgroup, fused_ssim.jl:52; the loss head uses 3 so that the three channels of a pixel —
// adjacent floats of the (C,W,H) image — are fetched and written by the same workgroup).
template <class Src, bool LOSS, int NCH>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(Src src, int W, int H, float C1, float C2, int train,
                                                       float* __restrict__ ssim_map, float* __restrict__ d0,
                                                       float* __restrict__ d1, float* __restrict__ d2,
                                                       float* __restrict__ partial) {
    __shared__ float sx[SH_DIM][IN_STRIDE], sy[SH_DIM][IN_STRIDE];
    __shared__ float hc[5][SH_DIM][HC_STRIDE];
    __shared__ float red[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    int x0, y0;
    const bool live = ssim_tile_of_block(W, H, x0, y0);
    float l1 = 0.0f, sv = 0.0f;
    if (!live) {  // padding workgroup of the XCD-aware grid: contributes zero partials
        if (LOSS && tid == 0) { partial[2 * ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z)] = 0.0f; partial[2 * ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) + 1] = 0.0f; }
        return;
    }
  for (int ch = 0; ch < NCH; ch++) {
    const int plane = blockIdx.z * NCH + ch;
    if (ch > 0) __syncthreads();  // previous channel's LDS tiles fully consumed
    for (int f = tid; f < SH_DIM * SH_DIM; f += 256) {
        const int ly = f / SH_DIM, lx = f - ly * SH_DIM;
        const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        sx[ly][lx] = in ? src.x(gx, gy, plane) : 0.0f;
        sy[ly][lx] = in ? src.y(gx, gy, plane) : 0.0f;
    }
    __syncthreads();
    // horizontal 11x1: rows ty and ty+16
    for (int r = ty; r < SH_DIM; r += GSR_TILE) {
        const int cx = tx + HALO;
        float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int d = 1; d <= HALO; d++) {
            const float w = GAUSS[HALO - d];
            const float Xl = sx[r][cx - d], Yl = sy[r][cx - d], Xr = sx[r][cx + d], Yr = sy[r][cx + d];
            s0 += (Xl + Xr) * w;
            s1 += (Xl * Xl + Xr * Xr) * w;
            s2 += (Yl + Yr) * w;
            s3 += (Yl * Yl + Yr * Yr) * w;
            s4 += (Xl * Yl + Xr * Yr) * w;
        }
        const float Xc = sx[r][cx], Yc = sy[r][cx], wc = GAUSS[HALO];
        s0 += Xc * wc; s1 += Xc * Xc * wc; s2 += Yc * wc; s3 += Yc * Yc * wc; s4 += Xc * Yc * wc;
        hc[0][r][tx] = s0; hc[1][r][tx] = s1; hc[2][r][tx] = s2; hc[3][r][tx] = s3; hc[4][r][tx] = s4;
    }
    __syncthreads();
    // vertical 1x11 + SSIM
    const int px = x0 + tx, py = y0 + ty;
    const bool in = px < W && py < H;
    float o[5];
    {
        const int cy = ty + HALO;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            float a = 0;
#pragma unroll
            for (int d = 1; d <= HALO; d++) a += (hc[k][cy - d][tx] + hc[k][cy + d][tx]) * GAUSS[HALO - d];
            a += hc[k][cy][tx] * GAUSS[HALO];
            o[k] = a;
        }
    }
    if (in) {
        const float mu1 = o[0], mu2 = o[2];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
        const float sigma1_sq = o[1] - mu1_sq, sigma2_sq = o[3] - mu2_sq, sigma12 = o[4] - mu1 * mu2;
        const float A = mu1_sq + mu2_sq + C1, Bv = sigma1_sq + sigma2_sq + C2;
        const float Cv = 2.0f * mu1 * mu2 + C1, Dv = 2.0f * sigma12 + C2;
        const size_t oi = (size_t)px + (size_t)W * py + (size_t)W * H * plane;
#if SSIM_EXACT
        const float val = (Cv * Dv) / (A * Bv);
        if (!LOSS) ssim_map[oi] = val;
        if (train) {
            d0[oi] = ((mu2 * 2.0f * Dv) / (A * Bv) - (mu2 * 2.0f * Cv) / (A * Bv) -
                      (mu1 * 2.0f * Cv * Dv) / (A * A * Bv) + (mu1 * 2.0f * Cv * Dv) / (A * Bv * Bv));
            d1[oi] = (-Cv * Dv) / (A * Bv * Bv);
            d2[oi] = (2.0f * Cv) / (A * Bv);
        }
#else
        // the same four quotients (fused_ssim.jl:219-233) over two reciprocals: 1/(AB) = rA·rB, 1/(A²B) = rA·rAB, 1/(AB²) = rB·rAB
        const float rA = __builtin_amdgcn_rcpf(A), rB = __builtin_amdgcn_rcpf(Bv), rAB = rA * rB;
        const float val = (Cv * Dv) * rAB;
        if (!LOSS) ssim_map[oi] = val;
        if (train) {
            d0[oi] = 2.0f * (mu2 * (Dv - Cv) * rAB + mu1 * val * (rB - rA));
            d1[oi] = -val * rB;
            d2[oi] = 2.0f * Cv * rAB;
        }
#endif
        if (LOSS) {
            sv += val;
            l1 += fabsf(sx[ty + HALO][tx + HALO] - sy[ty + HALO][tx + HALO]);
        }
    }
  }
    if (LOSS) {
        // one partial pair per workgroup; 24k workgroups hammering two words with atomics
        // serialise at ~12 ns each (MI355X_MICROARCH.md "fanin")
        const float a = block_sum(l1, red), b = block_sum(sv, red);
        if (tid == 0) {
            const size_t blk = (size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z;
            partial[2 * blk] = a;
            partial[2 * blk + 1] = b;
        }
    }
}

// The scalar loss (training.jl:656,684-694) from the per-workgroup partial sums the forward kernel left: the work of
// ONE workgroup, done by an extra workgroup of the backward launch (a launch of its own was 4 us during which the
// whole GPU waited).  Fixed summation order: bit-reproducible.
__device__ __forceinline__ void loss_finish_body(const float* __restrict__ partial, int n_blocks, float lambda,
                                                 float inv_count, float* __restrict__ loss_out) {
    __shared__ float red[2][4];
    float a = 0.0f, b = 0.0f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        const float2 p = reinterpret_cast<const float2*>(partial)[i];
        a += p.x; b += p.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0f; b = 0.0f;
        for (int w = 0; w < 4; w++) { a += red[0][w]; b += red[1][w]; }
        const float l1 = a * inv_count;
        const float s = 1.0f - b * inv_count;
        loss_out[0] = (1.0f - lambda) * l1 + lambda * s;
    }
}

// fused_ssim.jl:241-371.  LOSS: dL_dmap is the constant -λ/(3P) (pullback of
// λ·(1-mean(map))), the L1 pullback is added, output goes to the (C,W,H) rasterizer layout.
template <class Src, bool LOSS, int NCH>
__global__ __launch_bounds__(256) void ssim_bwd_kernel(Src src, int W, int H, const float* __restrict__ dL_dmap,
                                                       float chain_const, float l1_scale,
                                                       const float* __restrict__ d0, const float* __restrict__ d1,
                                                       const float* __restrict__ d2, float* __restrict__ out,
                                                       int outC, const float* __restrict__ partial, int n_partial,
                                                       float lambda, float inv_count, float* __restrict__ loss_out) {
    __shared__ float sd[3][SH_DIM][IN_STRIDE];
    __shared__ float hc[3][SH_DIM][HC_STRIDE];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (LOSS && blockIdx.x == gridDim.x - 1) {  // the extra workgroup of the loss head's launch
        loss_finish_body(partial, n_partial, lambda, inv_count, loss_out);
        return;
    }
    int x0, y0;
    if (!ssim_tile_of_block(W, H, x0, y0)) return;
    const size_t P = (size_t)W * H;
    float gout[NCH];
  for (int ch = 0; ch < NCH; ch++) {
    const int plane = blockIdx.z * NCH + ch;
    gout[ch] = 0.0f;
    if (ch > 0) __syncthreads();
    for (int f = tid; f < SH_DIM * SH_DIM; f += 256) {
        const int ly = f / SH_DIM, lx = f - ly * SH_DIM;
        const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t gi = (size_t)gx + (size_t)W * gy + P * plane;
        const float chain = in ? (LOSS ? chain_const : dL_dmap[gi]) : 0.0f;
        sd[0][ly][lx] = (in ? d0[gi] : 0.0f) * chain;
        sd[1][ly][lx] = (in ? d1[gi] : 0.0f) * chain;
        sd[2][ly][lx] = (in ? d2[gi] : 0.0f) * chain;
    }
    __syncthreads();
    for (int r = ty; r < SH_DIM; r += GSR_TILE) {
        const int cx = tx + HALO;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float a = 0;
#pragma unroll
            for (int d = 1; d <= HALO; d++) a += (sd[k][r][cx - d] + sd[k][r][cx + d]) * GAUSS[HALO - d];
            a += sd[k][r][cx] * GAUSS[HALO];
            hc[k][r][tx] = a;
        }
    }
    __syncthreads();
    const int px = x0 + tx, py = y0 + ty;
    if (px < W && py < H) {
        const int cy = ty + HALO;
        float s[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float a = 0;
#pragma unroll
            for (int d = 1; d <= HALO; d++) a += (hc[k][cy - d][tx] + hc[k][cy + d][tx]) * GAUSS[HALO - d];
            a += hc[k][cy][tx] * GAUSS[HALO];
            s[k] = a;
        }
        const float p1 = src.x(px, py, plane), p2 = src.y(px, py, plane);
        float g = s[0] + 2.0f * p1 * s[1] + p2 * s[2];
        if (LOSS) {
            const float df = p1 - p2;
            g = g + l1_scale * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
            gout[ch] = g;
        } else {
            out[(size_t)px + (size_t)W * py + P * plane] = g;
        }
    }
  }
    if (LOSS) {
        const int px = x0 + tx, py = y0 + ty;
        if (px < W && py < H) {
            float* o = out + (size_t)outC * ((size_t)px + (size_t)W * py) + (size_t)blockIdx.z * NCH;
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) o[ch] = gout[ch];  // adjacent floats of one pixel
            // the loss head only sees features[1:3]: the other channels of vpixels (C > 3) get their zeros here, with
            // the pixel's colour cotangent (a launch of its own for them was 17 us in :rgbd mode)
            for (int c = NCH; c < outC; c++) o[c] = 0.0f;
        }
    }
}

}  // namespace

static dim3 ssim_grid(int W, int H, int planes) {
    const int n = ((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE);
    return dim3(8 * ((n + 7) / 8), 1, planes);
}

void SSIM_NAME(gsr_launch_ssim_fwd)(hipStream_t s, int W, int H, int CH, int B, const float* img, const float* ref, float C1,
                         float C2, int train, float* ssim_map, float* d0, float* d1, float* d2) {
    PlanarSrc src{img, ref, W, H};
    hipLaunchKernelGGL((ssim_fwd_kernel<PlanarSrc, false, 1>), ssim_grid(W, H, CH * B), dim3(256), 0, s, src, W, H, C1,
                       C2, train, ssim_map, d0, d1, d2, (float*)nullptr);
}

void SSIM_NAME(gsr_launch_ssim_bwd)(hipStream_t s, int W, int H, int CH, int B, const float* img, const float* ref,
                         const float* dL_dmap, const float* d0, const float* d1, const float* d2, float* dL_dimg) {
    PlanarSrc src{img, ref, W, H};
    hipLaunchKernelGGL((ssim_bwd_kernel<PlanarSrc, false, 1>), ssim_grid(W, H, CH * B), dim3(256), 0, s, src, W, H,
                       dL_dmap, 0.0f, 0.0f, d0, d1, d2, dL_dimg, 0, (const float*)nullptr, 0, 0.0f, 0.0f, (float*)nullptr);
}

void SSIM_NAME(gsr_launch_loss_fwd)(hipStream_t s, int W, int H, int C, const float* image, const float* target, float C1,
                         float C2, float* d0, float* d1, float* d2, float* partial) {
    RasterSrc src{image, target, W, H, C};
    hipLaunchKernelGGL((ssim_fwd_kernel<RasterSrc, true, 3>), ssim_grid(W, H, 1), dim3(256), 0, s, src, W, H, C1, C2, 1,
                       (float*)nullptr, d0, d1, d2, partial);
}

void SSIM_NAME(gsr_launch_loss_bwd)(hipStream_t s, int W, int H, int C, const float* image, const float* target, float lambda,
                         const float* d0, const float* d1, const float* d2, const float* partial, float* loss_out,
                         float* vpixels) {
    RasterSrc src{image, target, W, H, C};
    const float count = 3.0f * (float)W * (float)H;
    const float inv_count = 1.0f / count;
    dim3 g = ssim_grid(W, H, 1);
    const int n_partial = (int)(g.x * g.y * g.z);  // one pair per workgroup of the forward launch
    g.x += 1;                                      // + the workgroup that finishes the scalar loss
    hipLaunchKernelGGL((ssim_bwd_kernel<RasterSrc, true, 3>), g, dim3(256), 0, s, src, W, H,
                       (const float*)nullptr, -lambda * inv_count, (1.0f - lambda) * inv_count, d0, d1, d2, vpixels,
                       C, partial, n_partial, lambda, inv_count, loss_out);
}
