// Adaptive density control on the device: the mask construction, row composition
// (append / clone / split / prune of every parameter, both Adam moments and the statistics), the split noise and
// the opacity reset of the reference's DefaultStrategy.
// Reference behaviour: src/densification.jl:1-136 (densify_and_prune!, densify_clone!, densify_split!,
// _add_split_noise!), :138-191 (prune_points!), :193-297 (densification_postfix!, append_gaussians!,
// _append_optimizer!, _prune_optimizer!), src/gaussians.jl:119-137 (_reset_opacity!, inverse_sigmoid).
// Compiled with -ffp-contract=off (bit-reproducible fp32 against the oracle's restatement for everything
// except the transcendental calls exp / log / cos / sin, which differ from glibc's by ulps).
#include "gsr_kernels.h"

namespace {

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }  // NU.sigmoid

// maximum(exp.(gs.scales); dims=1) of Gaussian i (densification.jl:36-38,79-80,22)
__device__ __forceinline__ float max_exp_scale(const float* __restrict__ scales, int scale_dims, long long i) {
    if (scale_dims == 1) return expf(scales[i]);
    const float a = expf(scales[3 * i]), b = expf(scales[3 * i + 1]), c = expf(scales[3 * i + 2]);
    return fmaxf(fmaxf(a, b), c);
}

// ∇means_2d = accum ./ denom with NaN -> 0 (densification.jl:7-10)
__global__ __launch_bounds__(256) void grad_mean_kernel(long long n, const float* __restrict__ accum,
                                                        const float* __restrict__ denom, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = accum[i] / denom[i];
    out[i] = g != g ? 0.0f : g;
}

// kind 0: clone mask  = grad > thr  && max(exp(scales)) < gamma            (densification.jl:34-38)
// kind 1: split mask  = padded_grad >= thr && max(exp(scales)) > gamma     (densification.jl:73-80; grad is padded with
//                       zeros for the rows appended since it was computed)
// kind 2: valid mask  = sigmoid(opacity) > min_opacity [&& max_radii < max_screen_size && max(exp(scales)) < gamma]
//                                                                          (densification.jl:18-25)
__global__ __launch_bounds__(256) void densify_mask_kernel(int kind, long long n, long long n_grad,
                                                           const float* __restrict__ grad,
                                                           const float* __restrict__ scales, int scale_dims,
                                                           const float* __restrict__ opacities,
                                                           const int32_t* __restrict__ max_radii, float thr, float gamma,
                                                           float min_opacity, int max_screen_size,
                                                           uint8_t* __restrict__ mask) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool m;
    if (kind == 0) {
        m = (i < n_grad ? grad[i] : 0.0f) > thr && max_exp_scale(scales, scale_dims, i) < gamma;
    } else if (kind == 1) {
        m = (i < n_grad ? grad[i] : 0.0f) >= thr && max_exp_scale(scales, scale_dims, i) > gamma;
    } else {
        m = sigmoid_(opacities[i]) > min_opacity;
        if (max_screen_size > 0) m = m && max_radii[i] < max_screen_size && max_exp_scale(scales, scale_dims, i) < gamma;
    }
    mask[i] = m ? 1 : 0;
}

// Row composition: for every array g (a parameter, an Adam moment, a statistic; rows of row_words 4-byte words)
//   dst[r]                      = src[keep_idx[r]]            r <  n_keep      (keep_idx == nullptr: identity)
//   dst[n_keep + k * n_sel + j] = new_zero ? 0 : src[sel_idx[j]]   j < n_sel, k < reps
// — `cat(x, x[:, mask])` (clone, reps = 1), `cat(x, repeat(x[:, mask], 1, 2))[:, valid]` (split, reps = 2: the
// block-repeat order of Julia's `repeat`), `x[:, valid]` (prune, n_sel = 0), with zero rows appended to the
// moments (_append_optimizer!).  All arrays in one launch; a workgroup composes a run of whole output rows of ONE array
// (about COMPOSE_WORDS words): it stages the rows' source indices in LDS, then its threads walk the run word by word —
// consecutive lanes store consecutive words, and read consecutive words of (mostly consecutive) source rows.  (Round 6: the
// first form, one thread per output word with a 64-bit division and modulo each, composed 1.1 M Gaussians' 19 arrays at
// 1.3 TB/s — 1.2 ms per composition, three per densification round, on the integer pipe.)
constexpr int COMPOSE_WORDS = 2048;
struct ComposeGroups {
    const uint32_t* src[GSR_COMPOSE_MAX_GROUPS];
    uint32_t* dst[GSR_COMPOSE_MAX_GROUPS];
    int row_words[GSR_COMPOSE_MAX_GROUPS];
    int new_zero[GSR_COMPOSE_MAX_GROUPS];
    long long block_start[GSR_COMPOSE_MAX_GROUPS + 1];
    int n;
};
__host__ __device__ inline int compose_rows_per_block(int rw) { return rw >= COMPOSE_WORDS ? 1 : COMPOSE_WORDS / rw; }
__global__ __launch_bounds__(256) void compose_rows_kernel(ComposeGroups G, const uint32_t* __restrict__ keep_idx,
                                                           long long n_keep, const uint32_t* __restrict__ sel_idx,
                                                           long long n_sel, int reps) {
    __shared__ uint32_t src_row[COMPOSE_WORDS];  // source row of each output row of this run; ~0u: a zero row
    int g = 0;
    for (int k = 1; k < G.n; k++)
        if ((long long)blockIdx.x >= G.block_start[k]) g = k;
    const int rw = G.row_words[g];
    const int rpb = compose_rows_per_block(rw);
    const long long rows = n_keep + n_sel * reps;
    const long long r0 = ((long long)blockIdx.x - G.block_start[g]) * rpb;
    if (r0 >= rows) return;
    const int nr = (int)(rows - r0 < (long long)rpb ? rows - r0 : (long long)rpb);
    const bool zero_new = G.new_zero[g] != 0;
    for (int k = threadIdx.x; k < nr; k += 256) {
        const long long r = r0 + k;
        uint32_t sr;
        if (r < n_keep) {
            sr = keep_idx ? keep_idx[r] : (uint32_t)r;
        } else if (zero_new) {
            sr = ~0u;
        } else {
            long long j = r - n_keep;  // block-repeat order: copy k of selected row j sits at n_keep + k * n_sel + j
            while (j >= n_sel) j -= n_sel;
            sr = sel_idx[j];
        }
        src_row[k] = sr;
    }
    __syncthreads();
    const uint32_t* __restrict__ src = G.src[g];
    uint32_t* __restrict__ dst = G.dst[g] + r0 * rw;
    const uint32_t urw = (uint32_t)rw;
    const long long nw = (long long)nr * rw;   // (<= COMPOSE_WORDS unless one row is wider than that)
    for (long long w = threadIdx.x; w < nw; w += 256) {
        const uint32_t row = rpb == 1 ? 0u : (uint32_t)w / urw;
        const uint32_t j = rpb == 1 ? (uint32_t)w : (uint32_t)w - row * urw;
        const uint32_t sr = src_row[row];
        dst[w] = sr == ~0u ? 0u : src[(size_t)sr * urw + j];
    }
}

// Counter-based generator for the split noise: 32 well-mixed bits from (seed, row, draw) — integer only, so the
// host restatement reproduces the stream bit for bit; the reference draws from the backend's device RNG
// (`randn(Float32)`, densification.jl:128), which no implementation can reproduce.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t row, uint32_t draw) {
    const uint32_t h = mix32(mix32(seed ^ (row * 0x9E3779B9u)) + draw * 0x85EBCA6Bu);
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
}

// The tail of densify_split! on the 2m appended rows (densification.jl:81-104,121-135):
//   sigma = exp(scale)              (stds = repeat(exp.(gs.scales)[:, mask], 1, 2))
//   point += R(q) * (sigma .* randn3)   (_add_split_noise!; R = unnorm_quat2rot(q), render.jl:322-333)
//   scale  = log(sigma / (0.8 * 2))
__global__ __launch_bounds__(256) void split_transform_kernel(long long n_new, int scale_dims, float* __restrict__ points,
                                                              const float4* __restrict__ rots, float* __restrict__ scales,
                                                              uint32_t seed) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_new) return;
    float sg[3];
    if (scale_dims == 1) { sg[0] = sg[1] = sg[2] = expf(scales[i]); }
    else { sg[0] = expf(scales[3 * i]); sg[1] = expf(scales[3 * i + 1]); sg[2] = expf(scales[3 * i + 2]); }
    // Box-Muller: two pairs of uniforms -> three normals
    const float u1 = uniform01(seed, (uint32_t)i, 0), u2 = uniform01(seed, (uint32_t)i, 1);
    const float u3 = uniform01(seed, (uint32_t)i, 2), u4 = uniform01(seed, (uint32_t)i, 3);
    const float r1 = sqrtf(-2.0f * logf(u1)), r2 = sqrtf(-2.0f * logf(u3));
    const float two_pi = 6.2831853071795864f;
    const float xi[3] = {sg[0] * (r1 * cosf(two_pi * u2)), sg[1] * (r1 * sinf(two_pi * u2)), sg[2] * (r2 * cosf(two_pi * u4))};
    const float4 q4 = rots[i];
    const float inv = 1.0f / sqrtf(q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w);
    const float w = q4.x * inv, x = q4.y * inv, y = q4.z * inv, z = q4.w * inv;
    const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const float R[3][3] = {{1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy)},
                           {2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx)},
                           {2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2)}};
#pragma unroll
    for (int r = 0; r < 3; r++)
        points[3 * i + r] = points[3 * i + r] + (R[r][0] * xi[0] + R[r][1] * xi[1] + R[r][2] * xi[2]);
    const float div = 0.8f * 2.0f;
    if (scale_dims == 1) scales[i] = logf(sg[0] / div);
    else {
#pragma unroll
        for (int c = 0; c < 3; c++) scales[3 * i + c] = logf(sg[c] / div);
    }
}

// _reset_opacity! (gaussians.jl:119-126): opacity = inverse_sigmoid(min(0.1, sigmoid(opacity)))
__global__ __launch_bounds__(256) void reset_opacity_kernel(long long n, float* __restrict__ opacities) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float o = fminf(0.1f, sigmoid_(opacities[i]));
    opacities[i] = logf(o / (1.0f - o));
}

// The 3DGS .ply vertex rows (src/gaussians.jl:140-203 `export_ply`, :205-247 `import_ply`): per Gaussian
//   x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_0..3kr-1 (CHANNEL-major: all of R, then G, then B) | opacity |
//   scale_0..2 | rot_0..3                                                     — 17 + 3·kr floats, raw parameters.
// PACK gathers the model's SoA arrays into that AoS row matrix in one pass (one thread per row word: coalesced
// stores), UNPACK scatters a row matrix back; the file header and the disk I/O stay on the host.
template <bool PACK>
__global__ __launch_bounds__(256) void ply_rows_kernel(long long n, int kr, float* __restrict__ points, float* __restrict__ dc,
                                                       float* __restrict__ rest, float* __restrict__ opac,
                                                       float* __restrict__ scales, float* __restrict__ rots,
                                                       float* __restrict__ rows) {
    const int w = 17 + 3 * kr;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * w) return;
    const long long i = e / w;
    const int j = (int)(e - i * w);
    float* p;
    if (j < 3) p = points + 3 * i + j;
    else if (j < 6) p = nullptr;  // normals: zeros on export, ignored on import
    else if (j < 9) p = dc + 3 * i + (j - 6);
    else if (j < 9 + 3 * kr) {
        const int q = j - 9, c = q / kr, k = q - c * kr;  // file: channel-major; model: (gaussian, coefficient, channel)
        p = rest + ((size_t)i * kr + k) * 3 + c;
    } else if (j == 9 + 3 * kr) p = opac + i;
    else if (j < 13 + 3 * kr) p = scales + 3 * i + (j - 10 - 3 * kr);
    else p = rots + 4 * i + (j - 13 - 3 * kr);
    if (PACK) rows[e] = p ? *p : 0.0f;
    else if (p) *p = rows[e];
}

// The GSP_DEBUG gradient guard of `step!` (src/training.jl:772-777) and the per-parameter count of
// `nonfinite_gradient_report` (:534-552): for each gradient array (rows of row_words floats, one row per Gaussian) the
// number of Gaussians with a non-finite entry — one pass over all arrays, one counter per array; `first_bad[g]`
// receives the smallest offending Gaussian index (0xFFFFFFFF if none), the handle for the report's per-Gaussian dump.
struct ScanGroups {
    const float* src[GSR_ADAM_MAX_GROUPS];
    int row_words[GSR_ADAM_MAX_GROUPS];
    long long block_start[GSR_ADAM_MAX_GROUPS + 1];
    int n;
};
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(ScanGroups G, long long n_rows, uint32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ first_bad) {
    int g = 0;
    for (int k = 1; k < G.n; k++)
        if ((long long)blockIdx.x >= G.block_start[k]) g = k;
    const long long r = ((long long)blockIdx.x - G.block_start[g]) * 256 + threadIdx.x;
    bool bad = false;
    if (r < n_rows) {
        const float* row = G.src[g] + r * G.row_words[g];
        for (int j = 0; j < G.row_words[g]; j++) {
            const float v = row[j];
            bad = bad || !(fabsf(v) <= 3.4028234664e38f);  // NaN or ±Inf
        }
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) {
        atomicAdd(&counts[g], (uint32_t)__popcll(m));
        atomicMin(&first_bad[g], (uint32_t)(r + __builtin_ctzll(m)));
    }
}

// Morton (Z-order) code of every Gaussian's position inside the box [lo, lo + 1/inv_extent): 21 bits per axis, x in the
// lowest bit of each triple.  Not a reference function: the key of the optional spatial re-sort after a densification
// (densification.py reorder_spatially) — neighbours in the arrays become neighbours on screen, which is what the
// aggregating preprocess and the record gathers of the tile sort like.
__device__ __forceinline__ unsigned long long spread21(unsigned int v) {
    unsigned long long x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(256) void morton_codes_kernel(long long n, const float* __restrict__ points, float lx, float ly,
                                                           float lz, float ix, float iy, float iz,
                                                           unsigned long long* __restrict__ codes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float q = 2097151.0f;  // 2^21 - 1
    auto cell = [&](float v, float lo, float inv) {
        float t = (v - lo) * inv;
        t = t != t ? 0.0f : fminf(fmaxf(t, 0.0f), 1.0f);  // NaN -> 0
        return (unsigned int)(t * q);
    };
    codes[i] = spread21(cell(points[3 * i], lx, ix)) | (spread21(cell(points[3 * i + 1], ly, iy)) << 1) |
               (spread21(cell(points[3 * i + 2], lz, iz)) << 2);
}

}  // namespace

void gsr_launch_nonfinite_scan(hipStream_t s, int n_groups, const float* const* src, const int* row_words, long long n_rows,
                               uint32_t* counts, uint32_t* first_bad) {
    ScanGroups G;
    G.n = n_groups;
    long long blocks = 0;
    for (int g = 0; g < GSR_ADAM_MAX_GROUPS; g++) {
        const bool on = g < n_groups;
        G.src[g] = on ? src[g] : nullptr;
        G.row_words[g] = on ? row_words[g] : 1;
        G.block_start[g] = blocks;
        if (on) blocks += (n_rows + 255) / 256;
    }
    G.block_start[GSR_ADAM_MAX_GROUPS] = blocks;
    (void)hipMemsetAsync(counts, 0, sizeof(uint32_t) * n_groups, s);
    (void)hipMemsetAsync(first_bad, 0xFF, sizeof(uint32_t) * n_groups, s);
    if (blocks == 0) return;
    hipLaunchKernelGGL(nonfinite_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, n_rows, counts, first_bad);
}

void gsr_launch_ply_rows(hipStream_t s, bool pack, long long n, int kr, float* points, float* dc, float* rest, float* opac,
                         float* scales, float* rots, float* rows) {
    if (n <= 0) return;
    const long long total = n * (17 + 3 * kr);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (pack) hipLaunchKernelGGL(ply_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, n, kr, points, dc, rest, opac, scales, rots, rows);
    else hipLaunchKernelGGL(ply_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, n, kr, points, dc, rest, opac, scales, rots, rows);
}

void gsr_launch_grad_mean(hipStream_t s, long long n, const float* accum, const float* denom, float* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(grad_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, accum, denom, out);
}

void gsr_launch_densify_mask(hipStream_t s, int kind, long long n, long long n_grad, const float* grad, const float* scales,
                             int scale_dims, const float* opacities, const int32_t* max_radii, float thr, float gamma,
                             float min_opacity, int max_screen_size, uint8_t* mask) {
    if (n <= 0) return;
    hipLaunchKernelGGL(densify_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kind, n, n_grad, grad, scales,
                       scale_dims, opacities, max_radii, thr, gamma, min_opacity, max_screen_size, mask);
}

void gsr_launch_compose_rows(hipStream_t s, int n_groups, const void* const* src, void* const* dst, const int* row_words,
                             const int* new_zero, const uint32_t* keep_idx, long long n_keep, const uint32_t* sel_idx,
                             long long n_sel, int reps) {
    ComposeGroups G;
    G.n = n_groups;
    long long blocks = 0;
    const long long rows = n_keep + n_sel * reps;
    for (int g = 0; g < GSR_COMPOSE_MAX_GROUPS; g++) {
        const bool on = g < n_groups;
        G.src[g] = on ? (const uint32_t*)src[g] : nullptr;
        G.dst[g] = on ? (uint32_t*)dst[g] : nullptr;
        G.row_words[g] = on ? row_words[g] : 1;
        G.new_zero[g] = on ? new_zero[g] : 0;
        G.block_start[g] = blocks;
        if (on) {
            const int rpb = compose_rows_per_block(row_words[g]);
            blocks += (rows + rpb - 1) / rpb;
        }
    }
    G.block_start[GSR_COMPOSE_MAX_GROUPS] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(compose_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, keep_idx, n_keep, sel_idx, n_sel, reps);
}

void gsr_launch_split_transform(hipStream_t s, long long n_new, int scale_dims, float* points, const float* rots,
                                float* scales, uint32_t seed) {
    if (n_new <= 0) return;
    hipLaunchKernelGGL(split_transform_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, s, n_new, scale_dims,
                       points, reinterpret_cast<const float4*>(rots), scales, seed);
}

void gsr_launch_morton_codes(hipStream_t s, long long n, const float* points, const float lo[3], const float inv_extent[3],
                             unsigned long long* codes) {
    if (n <= 0) return;
    hipLaunchKernelGGL(morton_codes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, points, lo[0], lo[1], lo[2],
                       inv_extent[0], inv_extent[1], inv_extent[2], codes);
}

void gsr_launch_reset_opacity(hipStream_t s, long long n, float* opacities) {
    if (n <= 0) return;
    hipLaunchKernelGGL(reset_opacity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, opacities);
}
