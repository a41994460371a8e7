// Per-Gaussian kernels: fused preprocess (project! + spherical_harmonics! + tile
// rect + per-tile occupancy count) and the fused per-Gaussian backward
// (∇project! + ∇spherical_harmonics!).
//
// Reference behaviour: src/rasterization/projection.jl:39-257, 259-393,
// spherical_harmonics.jl:1-181, utils.jl:14-29,122-142, render.jl:288-420.
//
// This translation unit is compiled with -ffp-contract=off: every fp32 expression
// is evaluated as written (no FMA), so the discrete outputs (radii, tile rects) and
// the per-Gaussian floats are bit-reproducible against the CPU oracle.  These
// kernels are HBM-bound (192 B of SH per Gaussian), the extra VALU ops are free.
#include "gsr_kernels.h"
#include <type_traits>
#ifndef GSR_PGB_MINWAVES
#define GSR_PGB_MINWAVES 1
#endif
#include "tile_mask.h"
#include "wave_reduce.h"
#include "adam_math.h"
#include "agg_plan.h"

namespace {

constexpr uint32_t EMIT_COOP = 48;  // tiles per rect above which the wave emits cooperatively
constexpr int kAggThreads = gsr_agg::kThreads;  // workgroup of preprocess_kernel's aggregating form (agg_plan.h: shared with the policy layer)
// Gradient-row slots of a Gaussian (Gaussian-major, gsr_kernels.h): rects of at most DENSE_RECT tiles get one slot per
// EMITTED tile — preprocess keeps the bit mask of the rect's tiles that passed the footprint test in the record, the
// sort's emit ranks a tile by a popcount below its bit, the per-Gaussian backward sums popcount(mask) contiguous rows
// without repeating the tests.  Larger rects keep one slot per tile of the rect (the culled ones are never written or
// read).  Exact culling drops 31 % of config 3's instances: the row buffer and its read in pergauss_bwd shrink with it.
constexpr uint32_t DENSE_RECT = GSR_DENSE_RECT;
constexpr float SH0 = 0.28209479177387814f;
constexpr float SH1 = 0.4886025119029199f;
constexpr float SH2C1 = 1.0925484305920792f;
constexpr float SH2C2 = -1.0925484305920792f;
constexpr float SH2C3 = 0.31539156525252005f;
constexpr float SH2C4 = -1.0925484305920792f;
constexpr float SH2C5 = 0.5462742152960396f;
constexpr float SH3C1 = -0.5900435899266435f;
constexpr float SH3C2 = 2.890611442640554f;
constexpr float SH3C3 = -0.4570457994644658f;
constexpr float SH3C4 = 0.3731763325901154f;
constexpr float SH3C5 = -0.4570457994644658f;
constexpr float SH3C6 = 1.445305721320277f;
constexpr float SH3C7 = -0.5900435899266435f;

struct M33 { float m[3][3]; };
struct M22 { float m[2][2]; };

__device__ __forceinline__ M33 mul33(const M33& a, const M33& b) {
    M33 o;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return o;
}
__device__ __forceinline__ M33 tr33(const M33& a) {
    M33 o;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o.m[i][j] = a.m[j][i];
    return o;
}
__device__ __forceinline__ M33 add33(const M33& a, const M33& b) {
    M33 o;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o.m[i][j] = a.m[i][j] + b.m[i][j];
    return o;
}
__device__ __forceinline__ M22 mul22(const M22& a, const M22& b) {
    M22 o;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) o.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j];
    return o;
}

// world->camera rotation as row-major M33, honouring the device override (pose optimisation)
__device__ __forceinline__ void load_pose(const GsrCam& cam, M33& R, float t[3]) {
    // element-wise selects (a branch over two whole-array initialisations makes the optimiser
    // keep the pose in scratch memory)
    const bool dev = cam.R_dev != nullptr;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float host = cam.R[r * 3 + c];
            R.m[r][c] = dev ? cam.R_dev[c * 3 + r] : host;
        }
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = cam.t_dev ? cam.t_dev[k] : cam.t[k];
}

// render.jl:322-333
__device__ __forceinline__ M33 quat2rot(const float4 q4, float qn[4], float& inv_norm) {
    float n2 = q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w;
    inv_norm = 1.0f / sqrtf(n2);
    float w = q4.x * inv_norm, x = q4.y * inv_norm, y = q4.z * inv_norm, z = q4.w * inv_norm;
    qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    M33 R;
    R.m[0][0] = 1.0f - 2.0f * (y2 + z2); R.m[1][0] = 2.0f * (xy + wz); R.m[2][0] = 2.0f * (xz - wy);
    R.m[0][1] = 2.0f * (xy - wz); R.m[1][1] = 1.0f - 2.0f * (x2 + z2); R.m[2][1] = 2.0f * (yz + wx);
    R.m[0][2] = 2.0f * (xz + wy); R.m[1][2] = 2.0f * (yz - wx); R.m[2][2] = 1.0f - 2.0f * (x2 + y2);
    return R;
}

// render.jl:291-294
__device__ __forceinline__ M33 cov3d(const M33& Rg, const float s[3], M33& M) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) M.m[r][c] = Rg.m[r][c] * s[c];
    return mul33(M, tr33(M));
}

struct Persp {
    float lim[2], lim_neg[2], txy[2], rz;
    float J[2][3];
};
// shared part of projection.jl:259-287 / 289-353
__device__ __forceinline__ Persp persp_common(const float mc[3], const GsrCam& cam) {
    Persp p;
    const int res[2] = {cam.width, cam.height};
    p.rz = 1.0f / mc[2];
    float rz2 = p.rz * p.rz;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float tan_fov = (0.5f * (float)res[k]) / cam.focal[k];
        float stf = 0.3f * tan_fov;
        float pp = cam.principal[k] * (float)res[k];
        p.lim[k] = ((float)res[k] - pp) / cam.focal[k] + stf;
        p.lim_neg[k] = pp / cam.focal[k] + stf;
        float v = mc[k] * p.rz;
        float c = fmaxf(-p.lim_neg[k], v);
        c = fminf(p.lim[k], c);
        p.txy[k] = mc[2] * c;
    }
    p.J[0][0] = cam.focal[0] * p.rz; p.J[1][0] = 0.0f;
    p.J[0][1] = 0.0f;                p.J[1][1] = cam.focal[1] * p.rz;
    p.J[0][2] = -cam.focal[0] * p.txy[0] * rz2;
    p.J[1][2] = -cam.focal[1] * p.txy[1] * rz2;
    return p;
}

// utils.jl:14-29 get_rect; float ceil-div as gpu_cld
__device__ __forceinline__ void get_rect(float mx, float my, int radius, int gx, int gy, int rmin[2], int rmax[2]) {
    const float px[2] = {mx, my};
    const int grid[2] = {gx, gy};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float lo = floorf((px[k] - (float)radius) / 16.0f);
        float hi_arg = (px[k] + (float)radius) + 16.0f - 1.0f;
        float hi = floorf(hi_arg / 16.0f);
        int ilo = (int)lo, ihi = (int)hi;
        rmin[k] = ilo < 0 ? 0 : (ilo > grid[k] ? grid[k] : ilo);
        rmax[k] = ihi < 0 ? 0 : (ihi > grid[k] ? grid[k] : ihi);
    }
}

template <int DEG>
__device__ __forceinline__ void sh_basis(const float d[3], float b[16]) {
    float x = d[0], y = d[1], z = d[2];
    b[0] = SH0;
    if (DEG > 0) {
        b[1] = -SH1 * y; b[2] = SH1 * z; b[3] = -SH1 * x;
    }
    if (DEG > 1) {
        float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
        b[4] = SH2C1 * xy; b[5] = SH2C2 * yz; b[6] = SH2C3 * (2.0f * z2 - x2 - y2);
        b[7] = SH2C4 * xz; b[8] = SH2C5 * (x2 - y2);
        if (DEG > 2) {
            b[9] = SH3C1 * y * (3.0f * x2 - y2);
            b[10] = SH3C2 * xy * z;
            b[11] = SH3C3 * y * (4.0f * z2 - x2 - y2);
            b[12] = SH3C4 * z * (2.0f * z2 - 3.0f * x2 - 3.0f * y2);
            b[13] = SH3C5 * x * (4.0f * z2 - x2 - y2);
            b[14] = SH3C6 * z * (x2 - y2);
            b[15] = SH3C7 * x * (x2 - 3.0f * y2);
        }
    }
}

// projection.jl:14-27
__device__ __forceinline__ void gaussian_normal(const M33& Rw, const M33& Rg, const float s[3], const float mc[3],
                                                float n[3], int& k, float& sign) {
    k = (s[0] <= s[1] && s[0] <= s[2]) ? 0 : (s[1] <= s[2]) ? 1 : 2;
    // (arithmetic select: a ?: chain over Rg.m[r][k] is turned back into a dynamically indexed
    // load by the optimiser, which puts Rg in scratch)
    const float m0 = k == 0 ? 1.0f : 0.0f, m1 = k == 1 ? 1.0f : 0.0f, m2 = k == 2 ? 1.0f : 0.0f;
    float ax[3];
#pragma unroll
    for (int r = 0; r < 3; r++) ax[r] = m0 * Rg.m[r][0] + m1 * Rg.m[r][1] + m2 * Rg.m[r][2];
    float nc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) nc[r] = Rw.m[r][0] * ax[0] + Rw.m[r][1] * ax[1] + Rw.m[r][2] * ax[2];
    float d = nc[0] * mc[0] + nc[1] * mc[1] + nc[2] * mc[2];
    sign = d > 0.0f ? -1.0f : 1.0f;
#pragma unroll
    for (int r = 0; r < 3; r++) n[r] = sign * nc[r];
}

// ---------------------------------------------------------------------------------
// preprocess: project! (projection.jl:69-129) + spherical_harmonics!
// (spherical_harmonics.jl:12-17,41-74) + count_tiles_per_gaussian! (utils.jl:131-141),
// and the per-tile occupancy histogram that replaces cumsum!/duplicate/sort-by-tile.
// ---------------------------------------------------------------------------------
//   Two forms of the binning.  DIRECT (AGG_NT == 0): one returning global atomic per instance pair, as described below.
// AGGREGATING (large scenes on grids whose counter words fit the LDS three times per CU): what the global atomics cost on
// this part is the number of memory transactions a wave instruction makes (tools/atomic_rates.hip: 64 lanes on 64 random
// counter words run at 24 G atomics/s, on 64 consecutive words at 130 G/s), and one Gaussian per lane in the scene's
// order means 64 random words per instruction.  A workgroup of AGG_NT Gaussians therefore (1) adds its requests up per
// counter word in LDS (one 64-bit word per aligned tile pair, the global counters' packing), (2) issues ONE returning
// global atomic per word it counted in — consecutive lanes on consecutive words; the LDS word then holds the bin
// positions this workgroup's instances start at (config 3, 512 Gaussians: ~1 500 requests on ~1 250 of the 4 081 words,
// 16 % fewer global atomics, in address order), (3) re-walks its rects, the footprint tests replayed from the emitted
// mask, every instance taking its position with a returning LDS atomic, and stores the keys.  Both walks are flattened
// over the lanes of each wave (below: 28 % -> ~95 % lane efficiency).  Three workgroups per CU in
// several rounds, so that the phases of different workgroups overlap (1024 Gaussians per workgroup aggregate better and
// were measured slower: 0.19 ms against 0.15; so was one resident wave of persistent workgroups).  The order inside a
// bin differs from the direct form's (it is arbitrary in both; the tile sort fixes it); everything else is bit-identical.
//   n_words: 64-bit words of the counter array = LDS words of the AGG form ((T + 2) / 2).
template <int DEG, int AGG_NT /* 0: direct form, 256 threads; else the threads of the aggregating workgroup */,
          bool W32 = false /* aggregating form with 2 x 16-bit LDS words (grids whose 64-bit words do not fit; positions < 65 024) */,
          bool SCATTER = false /* the compact binning mode's SECOND pass (duplicate_with_keys! as count -> scan -> scatter): no
                                  projection — the geometry records are read back —, keys go to tile_start[t] + arrival rank */,
          bool BANDED = false /* aggregating form over horizontal bands of the tile grid (n_bands > 1); false: ONE band = the whole
                                 grid, the band loop and the rect clipping compile away (round 4's kernel, register for register) */>
// (direct form: AT MOST four waves per SIMD — with its per-lane walks gone it would fit five, and five waves of scattered returning
//  atomics are slower than four: config 5 0.66-0.68 ms against 0.60-0.62)
__global__ __launch_bounds__(AGG_NT ? AGG_NT : 256)
__attribute__((amdgpu_waves_per_eu(AGG_NT ? 3 * AGG_NT / 256 : 1, AGG_NT ? 8 : 4))) void preprocess_kernel(int n, int K, int channels,
                                                                      const float* __restrict__ means,
                                                                      const float* __restrict__ scales,
                                                                      const float4* __restrict__ rots,
                                                                      const float* __restrict__ opac,
                                                                      const float* __restrict__ shs, GsrCam cam,
                                                                      GsrGeom geom, uint32_t* __restrict__ tile_count,
                                                                      uint32_t* __restrict__ n_visible,
                                                                      uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                                      int n_words, const uint32_t* __restrict__ tile_start,
                                                                      int n_bands, int band_rows) {
    static_assert(!SCATTER || AGG_NT != 0, "the scatter pass exists in the aggregating form only");
    static_assert(!BANDED || AGG_NT != 0, "bands are the aggregating form's");
    constexpr bool AGG = AGG_NT != 0;
    constexpr int NT = AGG ? AGG_NT : 256;
    // AGG: per counter word (an aligned tile pair), this workgroup's two counts, then its two bin positions — 2 x 32 bits, or
    // 2 x 16 bits (W32) where a larger grid would not fit otherwise: a workgroup adds at most NT per tile, and a position that
    // does not fit 16 bits is beyond bin_cap (the launcher checks) and is clamped to "not stored"
    using AggWord = typename std::conditional<W32, uint32_t, unsigned long long>::type;
    extern __shared__ unsigned long long agg_raw[];
    AggWord* agg = reinterpret_cast<AggWord*>(agg_raw);
    auto agg_inc = [](uint32_t c0, uint32_t c1) -> AggWord {
        return W32 ? (AggWord)(c0 | (c1 << 16)) : (AggWord)((unsigned long long)c0 | ((unsigned long long)c1 << 32));
    };
    auto agg_lo = [](AggWord v) -> uint32_t { return W32 ? (uint32_t)v & 0xFFFFu : (uint32_t)v; };
    auto agg_hi = [](AggWord v) -> uint32_t { return W32 ? (uint32_t)v >> 16 : (uint32_t)((unsigned long long)v >> 32); };
    const int i = blockIdx.x * NT + threadIdx.x;
    // SCATTER: `bin_cap` carries only_above — 0: every tile; else only the tiles whose list is longer (the lists that overflowed
    // the fixed-capacity bins: the others' keys sit complete in their bins, round 5)
    const uint32_t only_above = SCATTER ? bin_cap : 0u;
    (void)n_words; (void)only_above;
    bool visible = false;
    uint32_t area = 0, clamp_bits = 0, emitted = 0;  // emitted: bit k = tile k of the rect (row-major) got an instance
    float m2[2] = {0, 0}, conic[3] = {0, 0, 0}, rgb[3] = {0, 0, 0}, mc_z = 0.0f, tau = 0.0f, opac_v = 0.0f;
    int rmin[2] = {0, 0}, rmax[2] = {0, 0};
    // SH colour of Gaussian i at world position p (spherical_harmonics.jl:12-17,41-74)
    auto sh_colour = [&](const float p[3]) {
        const float* sh = shs + (size_t)3 * K * i;
        float d[3] = {p[0] - cam.center[0], p[1] - cam.center[1], p[2] - cam.center[2]};
        float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] *= inv; d[1] *= inv; d[2] *= inv;
        float b[16];
        sh_basis<DEG>(d, b);
        constexpr int NB = (DEG + 1) * (DEG + 1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float res = b[0] * sh[c];
#pragma unroll
            for (int k = 1; k < NB; k++) res = res + b[k] * sh[3 * k + c];
            res = res + 0.5f + 1.1920929e-7f;
            if (res < 0.0f) clamp_bits |= 1u << c;
            rgb[c] = fmaxf(0.0f, res);
        }
    };
    if (SCATTER) {
        // second pass of the compact mode: everything the walks need is in the record preprocess wrote
        if (i < n && geom.radii[i] > 0) {
            const GsrGeoRec rec = geom.rec[i];
            visible = true;
            m2[0] = rec.q0.x; m2[1] = rec.q0.y; conic[0] = rec.q0.z; conic[1] = rec.q0.w; conic[2] = rec.q1.x;
            opac_v = rec.q1.y; mc_z = rec.q2.z;
            const uint32_t lo = __float_as_uint(rec.q3.x), hi = __float_as_uint(rec.q3.y);
            rmin[0] = (int)(lo & 0xFFFFu); rmin[1] = (int)(lo >> 16); rmax[0] = (int)(hi & 0xFFFFu); rmax[1] = (int)(hi >> 16);
            area = (uint32_t)((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]));
            tau = footprint_tau(opac_v);
            emitted = __float_as_uint(rec.q3.w);
        }
    } else if (i < n) {
        M33 R; float t[3];
        load_pose(cam, R, t);
        // the small inputs of this Gaussian are requested up front (one round trip instead of three dependent ones:
        // means -> rotation/scale -> opacity); 14 % of them turn out culled and waste 28 bytes each
        const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        const float4 q_in = rots[i];
        const float s_in[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float opac_in = opac[i];
        opac_v = opac_in;
        float mc[3];
#pragma unroll
        for (int r = 0; r < 3; r++) mc[r] = (R.m[r][0] * p[0] + R.m[r][1] * p[1] + R.m[r][2] * p[2]) + t[r];
        mc_z = mc[2];
        int radius = 0;
        M33 Rg; float s[3];
        if (cam.near_plane < mc[2] && mc[2] < cam.far_plane) {
            float qn[4], inv_norm;
            Rg = quat2rot(q_in, qn, inv_norm);
            s[0] = s_in[0]; s[1] = s_in[1]; s[2] = s_in[2];
            M33 M;
            M33 Sigma = cov3d(Rg, s, M);
            M33 Sc = mul33(mul33(R, Sigma), tr33(R));
            Persp pr = persp_common(mc, cam);
            const int res[2] = {cam.width, cam.height};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float pp = cam.principal[k] * (float)res[k];
                m2[k] = pr.rz * cam.focal[k] * mc[k] + pp;
            }
            float JS[2][3];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
                    JS[a][b] = pr.J[a][0] * Sc.m[0][b] + pr.J[a][1] * Sc.m[1][b] + pr.J[a][2] * Sc.m[2][b];
            M22 S2;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    S2.m[a][b] = JS[a][0] * pr.J[b][0] + JS[a][1] * pr.J[b][1] + JS[a][2] * pr.J[b][2];
            // add_blur (render.jl:387-396)
            S2.m[0][0] = S2.m[0][0] + cam.blur_eps;
            S2.m[1][1] = S2.m[1][1] + cam.blur_eps;
            float det = S2.m[0][0] * S2.m[1][1] - S2.m[0][1] * S2.m[1][0];
            if (det > 0.0f) {
                // inverse (render.jl:368-381)
                float det_inv = 1.0f / det;
                float tmp = -S2.m[0][1] * det_inv;
                conic[0] = S2.m[1][1] * det_inv; conic[1] = tmp; conic[2] = S2.m[0][0] * det_inv;
                // max_eigval_2D (render.jl:415-420)
                float mid = 0.5f * (S2.m[0][0] + S2.m[1][1]);
                float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                int rad = (int)ceilf(3.0f * sqrtf(lam));
                bool off = (m2[0] + (float)rad) <= 0.0f || (m2[0] - (float)rad) >= (float)res[0] ||
                           (m2[1] + (float)rad) <= 0.0f || (m2[1] - (float)rad) >= (float)res[1];
                if (rad > cam.radius_clip && !off) radius = rad;
            }
        }
        geom.radii[i] = radius;
        if (radius > 0) {
            visible = true;
            // (the normal first: it is the last reader of R, Rg, s and mc — 24 registers that would otherwise stay allocated
            //  across the SH stage, the kernel's register peak, whatever the render mode: 114 -> 100 VGPRs)
            if (channels > 5) {
                float nn[3]; int k; float sg;
                gaussian_normal(R, Rg, s, mc, nn, k, sg);
                geom.normal[i] = make_float4(nn[0], nn[1], nn[2], 0.0f);
            }
            sh_colour(p);
            get_rect(m2[0], m2[1], radius, cam.grid_x, cam.grid_y, rmin, rmax);
            area = (uint32_t)((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]));
            tau = footprint_tau(opac_in);
        }
    }
    // duplicate_with_keys! (utils.jl:85-120) restated per tile, fused into this kernel: the
    // instance takes the next free position of its tile's BIN (a fixed-capacity segment of
    // `bins`, capacity from the previous view) with a returning atomic on the tile's counter
    // and stores its (depth bits << 32 | id) key there — the counters end up holding the
    // reference's per-tile counts, and no separate count / scan / scatter pass over the
    // instances exists.  Horizontally adjacent tiles are consecutive 32-bit counters, so an
    // aligned pair is served by ONE 64-bit atomic; up to 8 are kept in flight before their
    // keys are stored (a returning device-scope atomic is a ~2 us round trip to the memory
    // side).  A position >= bin_cap is not stored: the host sees max count > capacity in the
    // scan's totals, grows the bins and repeats the pass (first view / a much denser view).
    // exact-cull mode: a tile none of whose pixels can reach alpha >= 1/255 gets no
    // instance (the reference keeps it and skips it pixel by pixel, render.jl:95).
#ifdef GSR_PRE_NO_BINNING
    // (A/B builds only — tools/experiments/r06_preprocess_floor.py: projection + SH + records without any instance, the floor under
    //  every re-formed binning)
    if (!SCATTER) { area = 0u; rmax[0] = rmin[0]; rmax[1] = rmin[1]; }
#endif
    const uint64_t key = ((uint64_t)__float_as_uint(mc_z) << 32) | (uint32_t)i;
    // the rect's aligned tile pairs in row-major order: visit(even tile index, bit 0: even tile emitted, bit 1: odd tile)
    // (rects of more than EMIT_COOP tiles are emitted by the whole wave below)
    auto walk = [&](auto&& visit, bool set_emitted) {
        uint32_t kk = 0u;
        for (int y = rmin[1]; y < rmax[1]; y++) {
            int x = rmin[0];
            while (x < rmax[0]) {
                const int t = y * cam.grid_x + x;
                const bool odd = t & 1;
                const bool pair = !odd && x + 1 < rmax[0];
                uint32_t c0, c1 = 0u;
                if (!set_emitted && area <= 32u) {  // the second walk of the AGG form: the first one's results
                    c0 = (emitted >> (kk & 31u)) & 1u;
                    if (pair) c1 = (emitted >> ((kk + 1u) & 31u)) & 1u;
                } else {
                    c0 = (!cam.exact_cull || tile_may_touch(m2[0], m2[1], conic[0], conic[1], conic[2], tau, x * GSR_TILE,
                                                            y * GSR_TILE)) ? 1u : 0u;
                    if (pair)
                        c1 = (!cam.exact_cull || tile_may_touch(m2[0], m2[1], conic[0], conic[1], conic[2], tau,
                                                               (x + 1) * GSR_TILE, y * GSR_TILE)) ? 1u : 0u;
                }
                // (bit kk = the rect's row-major tile index: the walk visits the tiles in exactly that order; only read
                //  back when area <= DENSE_RECT)
                if (set_emitted) emitted |= (c0 | (c1 << 1)) << (kk & 31u);
                kk += pair ? 2u : 1u;
                if (c0 | c1) visit((uint32_t)(t & ~1) /* aligned pair */, odd ? (c0 << 1) : (c0 | (c1 << 1)));
                x += pair ? 2 : 1;
            }
        }
    };
    // FLATTENED walks (both forms, below).  In
    // scene order a wave's slowest lane has 13.6 pair requests at config 3 and the average lane 3.9 — a per-lane walk runs at
    // 28 % lane efficiency.  Instead every lane announces its count, an owner table in LDS maps item -> lane (wave-local, no
    // barrier), and the wave works its ~250 items off 64 at a time, each lane fetching its item's Gaussian with ds_bpermute.
    // Gaussians of more than FLAT_MAX requests (1 %) join the wave-cooperative path below, which then also leaves their
    // emitted mask.
    constexpr int FLAT_MAX = 16;
#ifdef GSR_NO_FLAT
    const bool flat = false;  // (A/B builds: the per-lane walks)
#else
    const bool flat = true;
#endif
    __shared__ uint8_t own_tab[NT / 64][64 * FLAT_MAX];
    __shared__ uint32_t emit_tab[NT / 64][64];
    int fcnt = 0, fppr = 0;          // this lane's pair requests in the flattened walks, pairs per TWO rows of its rect
    bool coop_small = false;         // ... or too many of them: with the wave-cooperative path
    if (flat && visible && area <= EMIT_COOP) {
        // pairs are aligned on the LINEAR tile index t = y * grid_x + x: a row starting on an even t holds ceil(w / 2) of them, on
        // an odd t floor(w / 2) + 1; on grids of odd width the rows of a rect alternate between the two (fppr = their sum)
        const int w = rmax[0] - rmin[0], h = rmax[1] - rmin[1];
        const int t0 = rmin[1] * cam.grid_x + rmin[0], t0b = t0 + cam.grid_x;
        const int pa = ((t0 + w - 1) >> 1) - (t0 >> 1) + 1, pb = ((t0b + w - 1) >> 1) - (t0b >> 1) + 1;
        fppr = pa + pb;
        fcnt = ((h + 1) >> 1) * pa + (h >> 1) * pb;
        if (fcnt > FLAT_MAX) { fcnt = 0; coop_small = true; }
    }
    const bool walks = visible && area <= EMIT_COOP && !flat;
    if (!AGG && flat) {
        // the direct form with its walk flattened over the lanes: an item = one pair request = (at most) one returning global
        // atomic; two rounds of items in flight per lane
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t x = (uint32_t)fcnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if (ln >= off) x += y;
        }
        const uint32_t fpre = x - (uint32_t)fcnt, ftotal = __shfl(x, 63);
        for (int k = 0; k < fcnt; k++) own_tab[wv][fpre + k] = (uint8_t)ln;
        emit_tab[wv][ln] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t rlo = (uint32_t)rmin[0] | ((uint32_t)rmin[1] << 16);
        const uint32_t rhi = (uint32_t)rmax[0] | ((fppr > 0 ? (256u + (uint32_t)fppr - 1u) / (uint32_t)fppr : 0u) << 16);
        unsigned long long* tc64 = reinterpret_cast<unsigned long long*>(tile_count);
        constexpr int INFL = 2;
        for (uint32_t base = 0; base < ftotal; base += 64u * INFL) {
            uint32_t tt[INFL], cc[INFL], zz[INFL], ss[INFL];
            unsigned long long old[INFL];
#pragma unroll
            for (int u = 0; u < INFL; u++) {
                const uint32_t itu = base + 64u * (uint32_t)u + (uint32_t)ln;
                const bool on = itu < ftotal;
                const uint32_t it = on ? itu : ftotal - 1u;
                const int src = own_tab[wv][it];
                const uint32_t q = it - __shfl(fpre, src);
                const uint32_t lo = __shfl(rlo, src), hi = __shfl(rhi, src);
                const float smx = __shfl(m2[0], src), smy = __shfl(m2[1], src);
                const float sa = __shfl(conic[0], src), sb = __shfl(conic[1], src), sc = __shfl(conic[2], src);
                const float stau = __shfl(tau, src);
                zz[u] = __shfl(__float_as_uint(mc_z), src);
                ss[u] = (uint32_t)src;
                const int x0 = (int)(lo & 0xFFFFu), y0 = (int)(lo >> 16), x1 = (int)(hi & 0xFFFFu);
                const int w = x1 - x0, t0 = y0 * cam.grid_x + x0, t0b = t0 + cam.grid_x;
                const int pa = ((t0 + w - 1) >> 1) - (t0 >> 1) + 1, pb = ((t0b + w - 1) >> 1) - (t0b >> 1) + 1;
                const int r2 = (int)((q * (hi >> 16)) >> 8), rem = (int)q - r2 * (pa + pb);   // (q < 16: the multiply-shift is exact)
                const int row = 2 * r2 + (rem >= pa ? 1 : 0), pc = rem - (rem >= pa ? pa : 0), y = y0 + row;
                const int te = (((t0 + row * cam.grid_x) >> 1) + pc) << 1, xe = te - y * cam.grid_x;  // the pair's even tile (may lie left of the rect)
                const bool va = on && xe >= x0, vb = on && xe + 1 < x1;
                const uint32_t kka = (uint32_t)(row * w + (xe - x0)) & 31u, kkb = (uint32_t)(row * w + (xe + 1 - x0)) & 31u;
                tt[u] = (uint32_t)te;
                uint32_t c0 = 0u, c1 = 0u;
                if (va) c0 = (!cam.exact_cull || tile_may_touch(smx, smy, sa, sb, sc, stau, xe * GSR_TILE, y * GSR_TILE)) ? 1u : 0u;
                if (vb) c1 = (!cam.exact_cull || tile_may_touch(smx, smy, sa, sb, sc, stau, (xe + 1) * GSR_TILE, y * GSR_TILE)) ? 1u : 0u;
                cc[u] = c0 | (c1 << 1);
                if (cc[u]) {
                    atomicOr(&emit_tab[wv][src], (c0 << kka) | (c1 << kkb));
                    old[u] = atomicAdd(tc64 + (tt[u] >> 1), (unsigned long long)c0 | ((unsigned long long)c1 << 32));
                }
            }
#pragma unroll
            for (int u = 0; u < INFL; u++)
                if (cc[u]) {
                    const uint64_t skey = ((uint64_t)zz[u] << 32) | (uint32_t)(blockIdx.x * NT + (threadIdx.x & ~63) + ss[u]);
                    const uint32_t p0 = (uint32_t)old[u], p1 = (uint32_t)(old[u] >> 32);
                    if ((cc[u] & 1u) && p0 < bin_cap) bins[(size_t)tt[u] * bin_cap + p0] = skey;
                    if ((cc[u] & 2u) && p1 < bin_cap) bins[(size_t)(tt[u] + 1) * bin_cap + p1] = skey;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (fcnt > 0) emitted = emit_tab[wv][ln];
    } else if (!AGG) {
        constexpr int PEND = 8;
        uint32_t pend_t[PEND], pend_c[PEND];
        int np = 0;
        auto flush = [&]() {
            unsigned long long old[PEND];
#pragma unroll
            for (int k = 0; k < PEND; k++)
                if (k < np)
                    old[k] = atomicAdd(reinterpret_cast<unsigned long long*>(tile_count + pend_t[k]),
                                       (unsigned long long)(pend_c[k] & 1u) | ((unsigned long long)(pend_c[k] >> 1) << 32));
#pragma unroll
            for (int k = 0; k < PEND; k++)
                if (k < np) {
                    const uint32_t p0 = (uint32_t)old[k], p1 = (uint32_t)(old[k] >> 32);
                    if ((pend_c[k] & 1u) && p0 < bin_cap) bins[(size_t)pend_t[k] * bin_cap + p0] = key;
                    if ((pend_c[k] & 2u) && p1 < bin_cap) bins[(size_t)(pend_t[k] + 1) * bin_cap + p1] = key;
                }
            np = 0;
        };
        if (walks) {
            walk([&](uint32_t t, uint32_t c) {
                pend_t[np] = t;
                pend_c[np] = c;  // bit 0: even tile, bit 1: odd tile
                if (++np == PEND) flush();
            }, true);
            flush();
        }
    } else {
        // AGGREGATING form, in horizontal BANDS of the tile grid (round 5: n_bands == 1 is round 4's kernel).  The counter words of
        // a 4K grid do not fit the LDS three times per CU; the words of HALF the grid do — so the workgroup runs its two walks
        // once per band of `band_rows` tile rows, every rect clipped to the band: the same items, the same tests, the same
        // positions (a rect's rows are walked band by band instead of in one go; the emitted mask is indexed by the row inside the
        // FULL rect).  What the LDS buys on a large grid is not fewer global atomics (a 512-Gaussian workgroup hits mostly distinct
        // words there) but (a) their address order and (b) no same-address serialisation: a hot tile's instances meet in LDS first
        // (dense scenes: 1 % of a 4K grid at 50 x density took the direct form 1.9 ms).  Both walks are FLATTENED over the lanes
        // of the wave: every lane announces its count, an owner table in LDS maps item -> lane (wave-local, no barrier), and the
        // wave works its items off 64 at a time, each lane fetching its item's Gaussian with ds_bpermute.  Gaussians of more than
        // FLAT_MAX requests (1 %) join the wave-cooperative path below, which then also leaves their emitted mask.
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const bool flat_lane = fcnt > 0;  // this lane's rect takes the flattened walks
        const int rw_ = rmax[0] - rmin[0];
        emit_tab[wv][ln] = SCATTER ? emitted : 0u;  // SCATTER replays the record's mask; else the first walk's tests fill it
        unsigned long long* tc64 = reinterpret_cast<unsigned long long*>(tile_count);
        for (int band = 0; band < (BANDED ? n_bands : 1); band++) {
            const int yb0 = BANDED ? band * band_rows : 0, yb1 = BANDED ? min(cam.grid_y, yb0 + band_rows) : cam.grid_y;
            const int wbase = BANDED ? (yb0 * cam.grid_x) >> 1 : 0;
            const int wcount = BANDED ? ((yb1 * cam.grid_x - 1) >> 1) - wbase + 1 : n_words;
            if (BANDED && band > 0) __syncthreads();  // the previous band's second walk has taken its positions
            for (int w = threadIdx.x; w < wcount; w += NT) agg[w] = (AggWord)0;
            // this lane's rect clipped to the band: pairs are aligned on the LINEAR tile index t = y * grid_x + x — a row starting
            // on an even t holds ceil(w / 2) of them, on an odd t floor(w / 2) + 1; on grids of odd width the rows alternate
            const int yc0 = BANDED ? max(rmin[1], yb0) : rmin[1], yc1 = BANDED ? min(rmax[1], yb1) : rmax[1];
            int bcnt = BANDED ? 0 : fcnt, bppr = BANDED ? 0 : fppr;
            if (BANDED && flat_lane && yc1 > yc0) {
                const int h = yc1 - yc0, t0 = yc0 * cam.grid_x + rmin[0], t0b = t0 + cam.grid_x;
                const int pa = ((t0 + rw_ - 1) >> 1) - (t0 >> 1) + 1, pb = ((t0b + rw_ - 1) >> 1) - (t0b >> 1) + 1;
                bppr = pa + pb;
                bcnt = ((h + 1) >> 1) * pa + (h >> 1) * pb;
            }
            uint32_t x = (uint32_t)bcnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off);
                if (ln >= off) x += y;
            }
            const uint32_t fpre = x - (uint32_t)bcnt, ftotal = __shfl(x, 63);
            for (int k = 0; k < bcnt; k++) own_tab[wv][fpre + k] = (uint8_t)ln;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t rlo = (uint32_t)rmin[0] | ((uint32_t)yc0 << 16);
            // pairs per two rect rows, and the multiplier of the small division q / ppr = (q * fmul) >> 8 (q < 16, ppr <= 16: exact)
            // (bits 26..31: the mask bit of the clipped rect's first tile, (yc0 - y0) * w <= 32 — zero when there is ONE band)
            const uint32_t rhi = (uint32_t)rmax[0] | ((bppr > 0 ? (256u + (uint32_t)bppr - 1u) / (uint32_t)bppr : 0u) << 16) |
                                 (BANDED ? (uint32_t)((yc0 - rmin[1]) * rw_) << 26 : 0u);
            // one item = one aligned tile pair of one rect row: decode (every lane of the wave runs every trip — a ds_bpermute
            // reads zero from a lane that is switched off)
#define GSR_AGG_ITEM()                                                                                                          \
                const bool on = base + (uint32_t)ln < ftotal;                                                                   \
                const uint32_t it = on ? base + (uint32_t)ln : ftotal - 1u;                                                     \
                const int src = own_tab[wv][it];                                                                                \
                const uint32_t q = it - __shfl(fpre, src);                                                                      \
                const uint32_t lo = __shfl(rlo, src), hi = __shfl(rhi, src), skb = BANDED ? hi >> 26 : 0u;                       \
                const int x0 = (int)(lo & 0xFFFFu), y0 = (int)(lo >> 16), x1 = (int)(hi & 0xFFFFu);                             \
                const int w = x1 - x0, t0 = y0 * cam.grid_x + x0, t0b = t0 + cam.grid_x;                                        \
                const int pa = ((t0 + w - 1) >> 1) - (t0 >> 1) + 1, pb = ((t0b + w - 1) >> 1) - (t0b >> 1) + 1;                 \
                const int r2 = (int)((q * ((hi >> 16) & 0x3FFu)) >> 8), rem = (int)q - r2 * (pa + pb); /* (q < 16: exact) */    \
                const int row = 2 * r2 + (rem >= pa ? 1 : 0), pc = rem - (rem >= pa ? pa : 0), y = y0 + row;                    \
                /* the pair's even tile (may lie left of the rect) */                                                           \
                const int te = (((t0 + row * cam.grid_x) >> 1) + pc) << 1, xe = te - y * cam.grid_x;                            \
                const bool va = on && xe >= x0, vb = on && xe + 1 < x1;                                                         \
                const uint32_t kka = (skb + (uint32_t)(row * w + (xe - x0))) & 31u, kkb = (skb + (uint32_t)(row * w + (xe + 1 - x0))) & 31u; \
                const uint32_t t = (uint32_t)te;
            __syncthreads();  // agg zeroed
            for (uint32_t base = 0; base < ftotal; base += 64u) {
                GSR_AGG_ITEM()
                uint32_t c0 = 0u, c1 = 0u;
                if (SCATTER) {
                    const uint32_t em = emit_tab[wv][src];
                    c0 = va ? (em >> kka) & 1u : 0u; c1 = vb ? (em >> kkb) & 1u : 0u;
                    if (only_above) {  // (restricted pass: a tile whose list fits its bin takes no position here)
                        if (c0) c0 = tile_start[t + 1u] - tile_start[t] > only_above ? 1u : 0u;
                        if (c1) c1 = tile_start[t + 2u] - tile_start[t + 1u] > only_above ? 1u : 0u;
                    }
                } else {
                    const float smx = __shfl(m2[0], src), smy = __shfl(m2[1], src);
                    const float sa = __shfl(conic[0], src), sb = __shfl(conic[1], src), sc = __shfl(conic[2], src);
                    const float stau = __shfl(tau, src);
                    if (va) c0 = (!cam.exact_cull || tile_may_touch(smx, smy, sa, sb, sc, stau, xe * GSR_TILE, y * GSR_TILE)) ? 1u : 0u;
                    if (vb) c1 = (!cam.exact_cull || tile_may_touch(smx, smy, sa, sb, sc, stau, (xe + 1) * GSR_TILE, y * GSR_TILE)) ? 1u : 0u;
                }
                if (c0 | c1) {
                    if (!SCATTER) atomicOr(&emit_tab[wv][src], (c0 << kka) | (c1 << kkb));
                    atomicAdd(&agg[(t >> 1) - (uint32_t)wbase], agg_inc(c0, c1));
                }
            }
            __syncthreads();
            // one global atomic per word this workgroup counted in; the word then holds the positions its instances start at
            for (int w0 = threadIdx.x; w0 < wcount; w0 += 4 * NT) {  // four in flight per lane
                unsigned long long old[4];
                uint32_t any = 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const AggWord c = w0 + k * NT < wcount ? agg[w0 + k * NT] : (AggWord)0;
                    if (c) {
                        old[k] = atomicAdd(tc64 + wbase + w0 + k * NT, (unsigned long long)agg_lo(c) | ((unsigned long long)agg_hi(c) << 32));
                        any |= 1u << k;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (any & (1u << k)) {
                        if (W32) {
                            const uint32_t lim = 0xFFFFu - (uint32_t)NT;  // (beyond every stored position: "not stored", and no carry into the other half)
                            agg[w0 + k * NT] = (AggWord)(min((uint32_t)old[k], lim) | (min((uint32_t)(old[k] >> 32), lim) << 16));
                        } else {
                            agg[w0 + k * NT] = (AggWord)old[k];
                        }
                    }
            }
            __syncthreads();
            if (SCATTER || bin_cap > 0u) {
                // second walk: the same items, the tests replayed from the emitted mask, every instance takes its position
                for (uint32_t base = 0; base < ftotal; base += 64u) {
                    GSR_AGG_ITEM()
                    const uint32_t em = emit_tab[wv][src];
                    const uint32_t zb = __shfl(__float_as_uint(mc_z), src);
                    uint32_t c0 = va ? (em >> kka) & 1u : 0u, c1 = vb ? (em >> kkb) & 1u : 0u;
                    if (SCATTER && only_above) {
                        if (c0) c0 = tile_start[t + 1u] - tile_start[t] > only_above ? 1u : 0u;
                        if (c1) c1 = tile_start[t + 2u] - tile_start[t + 1u] > only_above ? 1u : 0u;
                    }
                    if (c0 | c1) {
                        const AggWord old = atomicAdd(&agg[(t >> 1) - (uint32_t)wbase], agg_inc(c0, c1));
                        const uint64_t skey = ((uint64_t)zb << 32) | (uint32_t)(blockIdx.x * NT + (threadIdx.x & ~63) + src);
                        const uint32_t p0 = agg_lo(old), p1 = agg_hi(old);
                        if (SCATTER) {
                            if (c0) bins[tile_start[t] + p0] = skey;
                            if (c1) bins[tile_start[t + 1u] + p1] = skey;
                        } else {
                            if (c0 && p0 < bin_cap) bins[(size_t)t * bin_cap + p0] = skey;
                            if (c1 && p1 < bin_cap) bins[(size_t)(t + 1) * bin_cap + p1] = skey;
                        }
                    }
                }
            }
#undef GSR_AGG_ITEM
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!SCATTER && flat_lane) emitted = emit_tab[wv][ln];
    }
    // Large footprints: one thread walking hundreds of tiles serialises the wave (the reference's
    // duplicate_with_keys! has exactly this loop, utils.jl:96-119).  Rects of more than EMIT_COOP
    // tiles are emitted by the whole wave, one tile per lane and round — same tests on the same
    // (broadcast) floats, so the lists are identical.
    {
        const int lane = threadIdx.x & 63;
        unsigned long long big = __builtin_amdgcn_ballot_w64(visible && (area > EMIT_COOP || coop_small));
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const float bmx = __shfl(m2[0], src), bmy = __shfl(m2[1], src);
            const float ba = __shfl(conic[0], src), bb = __shfl(conic[1], src), bc = __shfl(conic[2], src);
            const float btau = __shfl(tau, src);
            const int bx0 = __shfl(rmin[0], src), by0 = __shfl(rmin[1], src);
            const int bx1 = __shfl(rmax[0], src), by1 = __shfl(rmax[1], src);
            const uint32_t bz = __shfl(__float_as_uint(mc_z), src);
            const uint64_t bkey = ((uint64_t)bz << 32) | (uint32_t)(blockIdx.x * NT + (threadIdx.x & ~63) + src);
            const int w = bx1 - bx0, total = w * (by1 - by0);
            uint32_t first32 = 0u;  // which of the rect's first 32 tiles got an instance (its emitted mask when it is a small one)
            for (int e = lane; e < total; e += 64) {
                const int ry = e / w, rx = e - ry * w;
                const int x = bx0 + rx, y = by0 + ry;
                const bool pass = !cam.exact_cull || tile_may_touch(bmx, bmy, ba, bb, bc, btau, x * GSR_TILE, y * GSR_TILE);
                if (e < 64) first32 = (uint32_t)__builtin_amdgcn_ballot_w64(pass);  // (lane 0 is in the first trip)
                if (pass) {
                    const uint32_t t = (uint32_t)(y * cam.grid_x + x);
                    if (SCATTER) {
                        const uint32_t ts = tile_start[t];
                        if (!only_above || tile_start[t + 1u] - ts > only_above) bins[ts + atomicAdd(tile_count + t, 1u)] = bkey;
                    } else {
                        const uint32_t pos = atomicAdd(tile_count + t, 1u);
                        if (pos < bin_cap) bins[(size_t)t * bin_cap + pos] = bkey;
                    }
                }
            }
            first32 = __shfl(first32, 0);
            if (!SCATTER && lane == src && (uint32_t)total <= DENSE_RECT) emitted = first32;
        }
    }
    // Exclusive scan of the tile-rect areas inside the block: with bpre[block] (tile_scan) it
    // gives every Gaussian the offset of its instance slots (gradient rows) — the reference's
    // cumsum!(tiles_touched) (rasterizer.jl:333-335), restated hierarchically.
    if (!SCATTER) {
        // (blocks of 256 Gaussians whatever the workgroup size: bpre[i >> 8] is what the readers index)
        __shared__ uint32_t wsum[NT / 64];
        __shared__ uint32_t wvis[NT / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wave0 = wave & ~3;
        // gradient-row slots of this Gaussian: one per emitted tile (small rects), one per tile of the rect otherwise
        const uint32_t slots = area <= DENSE_RECT ? (uint32_t)__popc(emitted) : area;
        uint32_t x = slots;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        const unsigned long long vm = __ballot(visible);
        // (high half: the visible Gaussians whose rect holds at least one tile — the reference's instance count is > 0 iff any)
        const unsigned long long am = __ballot(visible && area > 0u);
        if (lane == 63) { wsum[wave] = x; wvis[wave] = (uint32_t)__popcll(vm) | ((uint32_t)__popcll(am) << 16); }
        __syncthreads();
        uint32_t woff = 0;
        for (int w = wave0; w < wave; w++) woff += wsum[w];
        const uint32_t lpre = woff + x - slots;
        if ((threadIdx.x & 255) == 255 && i - 255 < n) {
            geom.bsum[i >> 8] = lpre + slots;
            // visible count per block, summed by tile_scan (15 k same-address atomics would
            // serialise at ~12 ns each: 0.19 ms — the "fanin" price of MI355X_MICROARCH.md)
            n_visible[i >> 8] = wvis[wave0] + wvis[wave0 + 1] + wvis[wave0 + 2] + wvis[wave0 + 3];
        }
        if (visible) {
            GsrGeoRec rec;
            rec.q0 = make_float4(m2[0], m2[1], conic[0], conic[1]);
            rec.q1 = make_float4(conic[2], opac_v, rgb[0], rgb[1]);
            rec.q2 = make_float4(rgb[2], __uint_as_float(clamp_bits), mc_z, __uint_as_float(lpre));
            rec.q3 = make_float4(__uint_as_float((uint32_t)rmin[0] | ((uint32_t)rmin[1] << 16)),
                                 __uint_as_float((uint32_t)rmax[0] | ((uint32_t)rmax[1] << 16)),
                                 __uint_as_float(blend_threshold_bits(opac_v)), __uint_as_float(emitted));
            geom.rec[i] = rec;
        }
    }
}

// One SH optimizer group (features_dc: R = 3, k0 = 0; features_rest: R = 3 (K-1), k0 = 1) of the cnt Gaussians of a
// workgroup starting at i0, updated element-major by the whole workgroup so that θ, μ, ν stream as full cache lines
// (float4); grad(il, k, c) supplies the gradient of Gaussian il (0-based inside the workgroup), band k, channel c.  The
// updated coefficient is also written into the hcat copy `shs` the next forward reads (rasterizer.jl:218-228).  Shared by
// the backward's fused epilogue and by the multi-view tail (sh_views_tail_kernel): one definition, identical bits.
template <class Grad>
__device__ __forceinline__ void tail_sh_group(const gsr::TailState& TS, int i0, int cnt, int K3, float* __restrict__ th,
                                              float* __restrict__ mu, float* __restrict__ nu, int R, int k0,
                                              const gsr::AdamHyper& hy, Grad grad) {
    const size_t base = (size_t)i0 * R;
    const int total = cnt * R;
    const uint32_t inv = (1u << 20) / (uint32_t)R + 1u;  // e / R == (e * inv) >> 20 for e < 2^20 / R (R <= 45, e < 256 R)
    th += base; mu += base; nu += base;
    const bool aligned = ((((uintptr_t)th) | ((uintptr_t)mu) | ((uintptr_t)nu)) & 15) == 0;
    const int total4 = aligned ? total >> 2 : 0;
    auto element = [&](int e, float& t, float& m, float& v) {
        const int il = (int)(((uint32_t)e * inv) >> 20);
        const int j = e - il * R;
        const int kb = j / 3, c = j - 3 * kb, k = k0 + kb;
        const float g = grad(il, k, c);
        t = gsr::adam_update(t, g, m, v, hy);
        TS.shs[(size_t)(i0 + il) * K3 + 3 * k0 + j] = t;  // hcat(sh_color, sh_remainder) of the next forward
    };
    for (int f = threadIdx.x; f < total4; f += 256) {
        float4 t4 = reinterpret_cast<float4*>(th)[f], m4 = reinterpret_cast<float4*>(mu)[f],
               v4 = reinterpret_cast<float4*>(nu)[f];
        element(4 * f, t4.x, m4.x, v4.x);
        element(4 * f + 1, t4.y, m4.y, v4.y);
        element(4 * f + 2, t4.z, m4.z, v4.z);
        element(4 * f + 3, t4.w, m4.w, v4.w);
        reinterpret_cast<float4*>(th)[f] = t4;
        reinterpret_cast<float4*>(mu)[f] = m4;
        reinterpret_cast<float4*>(nu)[f] = v4;
    }
    for (int e = 4 * total4 + threadIdx.x; e < total; e += 256) {
        float t = th[e], m = mu[e], v = nu[e];
        element(e, t, m, v);
        th[e] = t; mu[e] = m; nu[e] = v;
    }
}

// ---------------------------------------------------------------------------------
// ∇scales / ∇rotations IN FLOAT64 (round 5).  The reference's chain  vconic -> ∇inverse (render.jl:383-385) ->
// ∇perspective_projection (projection.jl:289-353) -> ∇covar_world_to_cam -> ∇quat_scale_to_cov (render.jl:302-320) ->
// ∇unnorm_quat2rot (render.jl:335-366) multiplies 2x2 and 3x3 matrices whose eigenvalues differ by the SQUARE of a splat's
// aspect ratio.  For the flat, needle-projecting splats a trained scene is full of (scale ratios of 100 and more) fp32 loses
// the thin eigen-direction in every one of those products: the last bit of vconic — any summation order's — moves ∇rotations
// by 1e-3 relative, in the reference's kernels, in the CPU oracle and here alike, although the map itself is well conditioned
// (kappa ~ 3-10: the float64 replay of the oracle, oracle/gsr_oracle.c ORC_REAL_DOUBLE, is insensitive to those bits).  This
// kernel is HBM-bound; ~250 double-precision FMAs per visible Gaussian cost nothing (7 us of 140 at config 3).  So the part
// of the pullback that ends in ∇scales / ∇rotations runs in float64 FROM THE RAW INPUTS (the forward's fp32 conic cannot
// hold the thin eigenvalue either): the same formulas, vSigma = T' vS2 T folded with T = J R so that no 3x3 product of
// ill-scaled factors is ever rounded.  ∇means (well scaled) keeps the fp32 expression trees of the reference.
//   va, vb, vc: the summed conic cotangent (vb on both off-diagonal entries, as the reference places it); vRg: the normal
//   channel's contribution to the rotation matrix cotangent (projection.jl:227-235), zero outside :rgbdn.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void scales_rots_bwd_f64(const GsrCam& cam, const M33& Rf, const float tf[3], const float pf[3],
                                                    const float4 q4, const float sf[3], float va, float vb, float vc,
                                                    const M33& vRg, float vs_out[3], float vq_out[4]) {
    double R[3][3], mc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) R[r][c] = (double)Rf.m[r][c];
        mc[r] = (R[r][0] * (double)pf[0] + R[r][1] * (double)pf[1] + R[r][2] * (double)pf[2]) + (double)tf[r];
    }
    // projection.jl:259-287 (persp_common above, in double)
    const double rz = 1.0 / mc[2], rz2 = rz * rz;
    const int res[2] = {cam.width, cam.height};
    double J02[2], J00[2];  // J = [J00[0] 0 J02[0]; 0 J00[1] J02[1]]
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double fk = (double)cam.focal[k], rk = (double)res[k];
        const double stf = 0.3 * ((0.5 * rk) / fk), pp = (double)cam.principal[k] * rk;
        const double lim = (rk - pp) / fk + stf, lim_neg = pp / fk + stf;
        const double c = fmin(lim, fmax(-lim_neg, mc[k] * rz));
        J00[k] = fk * rz;
        J02[k] = -fk * (mc[2] * c) * rz2;
    }
    // render.jl:322-333 (quat2rot) and M = Rg diag(s)
    const double qw = (double)q4.x, qx = (double)q4.y, qy = (double)q4.z, qz = (double)q4.w;
    const double inv_norm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    const double w = qw * inv_norm, x = qx * inv_norm, y = qy * inv_norm, z = qz * inv_norm;
    double Rg[3][3];
    Rg[0][0] = 1.0 - 2.0 * (y * y + z * z); Rg[1][0] = 2.0 * (x * y + w * z); Rg[2][0] = 2.0 * (x * z - w * y);
    Rg[0][1] = 2.0 * (x * y - w * z); Rg[1][1] = 1.0 - 2.0 * (x * x + z * z); Rg[2][1] = 2.0 * (y * z + w * x);
    Rg[0][2] = 2.0 * (x * z + w * y); Rg[1][2] = 2.0 * (y * z - w * x); Rg[2][2] = 1.0 - 2.0 * (x * x + y * y);
    const double s[3] = {(double)sf[0], (double)sf[1], (double)sf[2]};
    // T = J R (2x3), TM = T Rg diag(s) (2x3)
    double T[2][3], TM[2][3];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) T[a][c] = J00[a] * R[a][c] + J02[a] * R[2][c];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) TM[a][c] = (T[a][0] * Rg[0][c] + T[a][1] * Rg[1][c] + T[a][2] * Rg[2][c]) * s[c];
    // Sigma_2D = TM TM' + blur (projection.jl:96-100), conic = its inverse
    const double S00 = TM[0][0] * TM[0][0] + TM[0][1] * TM[0][1] + TM[0][2] * TM[0][2] + (double)cam.blur_eps;
    const double S01 = TM[0][0] * TM[1][0] + TM[0][1] * TM[1][1] + TM[0][2] * TM[1][2];
    const double S11 = TM[1][0] * TM[1][0] + TM[1][1] * TM[1][1] + TM[1][2] * TM[1][2] + (double)cam.blur_eps;
    const double det_inv = 1.0 / (S00 * S11 - S01 * S01);
    const double Ca = S11 * det_inv, Cb = -S01 * det_inv, Cc = S00 * det_inv;
    // vS2 = -C vC C (render.jl:383-385), symmetric
    const double a_ = (double)va, b_ = (double)vb, c_ = (double)vc;
    const double P00 = Ca * a_ + Cb * b_, P01 = Ca * b_ + Cb * c_, P10 = Cb * a_ + Cc * b_, P11 = Cb * b_ + Cc * c_;
    const double V00 = -(P00 * Ca + P01 * Cb), V01 = -(P00 * Cb + P01 * Cc), V11 = -(P10 * Cb + P11 * Cc);
    // vM = (vSigma + vSigma') M with vSigma = T' vS2 T:  vM = 2 T' (vS2 TM)
    double X[2][3], vM[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) { X[0][c] = V00 * TM[0][c] + V01 * TM[1][c]; X[1][c] = V01 * TM[0][c] + V11 * TM[1][c]; }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) vM[r][c] = 2.0 * (T[0][r] * X[0][c] + T[1][r] * X[1][c]);
    double vRq[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        vs_out[c] = (float)(Rg[0][c] * vM[0][c] + Rg[1][c] * vM[1][c] + Rg[2][c] * vM[2][c]);
#pragma unroll
        for (int r = 0; r < 3; r++) vRq[r][c] = vM[r][c] * s[c] + (double)vRg.m[r][c];
    }
    // ∇unnorm_quat2rot (render.jl:335-366)
#define V(i_, j_) vRq[(i_) - 1][(j_) - 1]
    double vqn[4];
    vqn[0] = 2.0 * (x * (V(3, 2) - V(2, 3)) + y * (V(1, 3) - V(3, 1)) + z * (V(2, 1) - V(1, 2)));
    vqn[1] = 2.0 * (-2.0 * x * (V(2, 2) + V(3, 3)) + y * (V(2, 1) + V(1, 2)) + z * (V(3, 1) + V(1, 3)) + w * (V(3, 2) - V(2, 3)));
    vqn[2] = 2.0 * (x * (V(2, 1) + V(1, 2)) - 2.0 * y * (V(1, 1) + V(3, 3)) + z * (V(3, 2) + V(2, 3)) + w * (V(1, 3) - V(3, 1)));
    vqn[3] = 2.0 * (x * (V(3, 1) + V(1, 3)) + y * (V(3, 2) + V(2, 3)) - 2.0 * z * (V(1, 1) + V(2, 2)) + w * (V(2, 1) - V(1, 2)));
#undef V
    const double qn[4] = {w, x, y, z};
    const double dq = vqn[0] * w + vqn[1] * x + vqn[2] * y + vqn[3] * z;
#pragma unroll
    for (int k = 0; k < 4; k++) vq_out[k] = (float)((vqn[k] - dq * qn[k]) * inv_norm);
}

// ---------------------------------------------------------------------------------
// fused ∇project! (projection.jl:170-256) + ∇spherical_harmonics!
// (spherical_harmonics.jl:32-37,76-181).  Every output element is written exactly
// once (zeros for culled Gaussians and for SH bands above the active degree), so the
// 59·N-float gradient arena needs no memset.
//
// FUSED: the single-GPU trainer step.  The 59 gradient floats of a Gaussian never reach HBM: the
// trainer tail (pullback of the sigmoid / exp prologue + the six NU.Adam updates + the activated
// copies of the next forward; trainer.hip, training.jl:768-779) is applied in the epilogue.  The
// per-Gaussian groups are updated by the owning thread; the SH groups (48 of the 59 floats) by the
// whole workgroup, element-major over the workgroup's contiguous slice of features_dc /
// features_rest so that θ, μ, ν stream as full cache lines — the per-Gaussian factors of the SH
// gradient (basis x colour cotangent) wait in LDS.  `means` / `rots` ARE the raw points / rotations,
// `scales` the activated copy; the SH coefficients are read from features_rest (the hcat copy `shs`
// is only written).  Same expression trees as the unfused chain (adam_math.h): identical bits.
// ---------------------------------------------------------------------------------
// F32CHAIN (gsr_config.grad_precision = GSR_GRAD_FP32_REFERENCE): ∇scales / ∇rotations through the reference's own fp32
// expression trees instead of the float64 chain — reference-parity runs; a template parameter, not a branch, so
// that the default kernels are round 5's register for register.
template <int DEG, bool FUSED, bool F32CHAIN = false>
__global__ __launch_bounds__(256, GSR_PGB_MINWAVES) void pergauss_bwd_kernel(int n, int K, int channels, const float* __restrict__ means,
                                                           const float* __restrict__ scales,
                                                           const float4* __restrict__ rots,
                                                           const float* __restrict__ shs, GsrCam cam, GsrGeom geom,
                                                           GsrInst inst, float2* __restrict__ vmean2d_out,
                                                           float* __restrict__ vmeans,
                                                           float* __restrict__ vshs, float* __restrict__ vopac,
                                                           float* __restrict__ vscales, float4* __restrict__ vrots,
                                                           float* __restrict__ vR_out, float* __restrict__ vt_out,
                                                           float* __restrict__ vcolors, gsr::TailState TS) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float poseR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, poset[3] = {0, 0, 0};
    // ---- sum this Gaussian's per-instance gradient rows (written by composite_bwd) ----
    // acc: [0..2] v rgb, [3] v opacity, [4..6] v conic, [7] v depth, [8..9] v mean2d, [10..12] v normal
    // The radius, the geometry record and the Gaussian's own inputs are all requested at once (the record and the
    // inputs of a culled Gaussian are fetched for nothing: 104 bytes x 14 %); fetched one after the other — radius ->
    // record -> rows -> inputs -> SH — they would be five dependent round trips at 3 waves per SIMD.
    GsrGeoRec rec;
    rec.q0 = rec.q1 = rec.q2 = rec.q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float p_in[3] = {0, 0, 0}, s_in[3] = {0, 0, 0};
    float4 q_in = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    int radius_in = 0;
    if (i < n) {
        radius_in = geom.radii[i];
        rec = geom.rec[i];
        // (FUSED: these arrays are updated in place by the epilogue — read through the same non-restrict pointers)
        const float* mp = FUSED ? TS.points : means;
        const float* sp = FUSED ? TS.scales_act : scales;
        const float4* qp = FUSED ? reinterpret_cast<const float4*>(TS.rots) : rots;
        p_in[0] = mp[3 * i]; p_in[1] = mp[3 * i + 1]; p_in[2] = mp[3 * i + 2];
        s_in[0] = sp[3 * i]; s_in[1] = sp[3 * i + 1]; s_in[2] = sp[3 * i + 2];
        q_in = qp[i];
    }
    const bool visible = radius_in > 0;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.0f;
    uint32_t area = 0, goff = 0;
    if (visible) {
        const uint32_t lo = __float_as_uint(rec.q3.x), hi = __float_as_uint(rec.q3.y);
        area = ((hi & 0xFFFFu) - (lo & 0xFFFFu)) * ((hi >> 16) - (lo >> 16));
        goff = geom.bpre[i >> 8] + __float_as_uint(rec.q2.w);
    }
    constexpr uint32_t BIG = 48;  // larger footprints are summed by the whole wave
    // Rects beyond DENSE_RECT tiles: slots of tiles the exact footprint test culled at binning were never written by
    // composite_bwd (no instance exists): they are skipped by repeating the SAME test, on the
    // same record floats, in the same translation unit as the count / scatter kernels.
    uint32_t rx0 = 0, ry0 = 0, rw = 1;
    float tau = 0.0f;
    if (visible) {
        const uint32_t lo = __float_as_uint(rec.q3.x), hi = __float_as_uint(rec.q3.y);
        rx0 = lo & 0xFFFFu; ry0 = lo >> 16; rw = (hi & 0xFFFFu) - rx0;
        tau = footprint_tau(rec.q1.y);
    }
    if (area <= BIG) {
        // Which of the rect's tiles hold an instance (pure ALU), then the row loads in batches of ROWS_IN_FLIGHT:
        // a thread that waits for each 64-byte row before asking for the next one spends its life in memory
        // latency (3 waves per SIMD cannot hide it).  The sums run in ascending slot order whatever the batching
        // (fixed order -> bit-reproducible gradients).
        unsigned long long emask = 0ull;
        if (area <= DENSE_RECT) {
            // small rect: its emitted tiles own consecutive slots (preprocess left their mask in the record)
            // (the record of a culled Gaussian was never written: stale bits)
            const uint32_t cnt = visible ? (uint32_t)__popc(__float_as_uint(rec.q3.w)) : 0u;
            emask = cnt ? (~0ull >> (64u - cnt)) : 0ull;
        } else {
            uint32_t tx = rx0, ty = ry0;
            for (uint32_t k = 0; k < area; k++) {
                const bool emitted = !cam.exact_cull || tile_may_touch(rec.q0.x, rec.q0.y, rec.q0.z, rec.q0.w, rec.q1.x,
                                                                       tau, (int)tx * GSR_TILE, (int)ty * GSR_TILE);
                if (++tx == rx0 + rw) { tx = rx0; ty++; }
                emask |= emitted ? (1ull << k) : 0ull;
            }
        }
        while (emask) {
            // up to four rows in flight (explicitly unrolled: arrays indexed by the batch slot end up in scratch)
            const uint32_t k0 = (uint32_t)__builtin_ctzll(emask); emask &= emask - 1ull;
            const bool h1 = emask != 0ull; const uint32_t k1 = h1 ? (uint32_t)__builtin_ctzll(emask) : k0; emask &= emask - 1ull;
            const bool h2 = emask != 0ull; const uint32_t k2 = h2 ? (uint32_t)__builtin_ctzll(emask) : k0; emask &= emask - 1ull;
            const bool h3 = emask != 0ull; const uint32_t k3 = h3 ? (uint32_t)__builtin_ctzll(emask) : k0; emask &= emask - 1ull;
            const float4* r0 = inst.rows + (size_t)GSR_ROW_F4(channels) * (goff + k0);
            const float4* r1 = inst.rows + (size_t)GSR_ROW_F4(channels) * (goff + k1);
            const float4* r2 = inst.rows + (size_t)GSR_ROW_F4(channels) * (goff + k2);
            const float4* r3 = inst.rows + (size_t)GSR_ROW_F4(channels) * (goff + k3);
            const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2];
            const float4 b0 = r1[0], b1 = r1[1], b2 = r1[2];
            const float4 c0 = r2[0], c1 = r2[1], c2 = r2[2];
            const float4 d0 = r3[0], d1 = r3[1], d2 = r3[2];
            float4 a3 = a2, b3 = a2, c3 = a2, d3 = a2;
            if (channels > 5) { a3 = r0[3]; b3 = r1[3]; c3 = r2[3]; d3 = r3[3]; }
#define GSR_ADD_ROW(F0, F1, F2, F3)                                                                      \
            acc[0] += F0.x; acc[1] += F0.y; acc[2] += F0.z; acc[3] += F0.w;                              \
            acc[4] += F1.x; acc[5] += F1.y; acc[6] += F1.z; acc[7] += F1.w;                              \
            acc[8] += F2.x; acc[9] += F2.y;                                                              \
            if (channels > 5) { acc[10] += F2.z; acc[11] += F2.w; acc[12] += F3.x; }
            GSR_ADD_ROW(a0, a1, a2, a3)
            if (h1) { GSR_ADD_ROW(b0, b1, b2, b3) }  // (a duplicate load stands in for a missing row; it is not added)
            if (h2) { GSR_ADD_ROW(c0, c1, c2, c3) }
            if (h3) { GSR_ADD_ROW(d0, d1, d2, d3) }
#undef GSR_ADD_ROW
        }
    }
    {
        __shared__ float bounce[4][16];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const gsr::LaneBits lb(lane);
        const int slot = gsr::wave_reduce_index<13>(lane);
        unsigned long long big = __ballot(area > BIG);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const uint32_t a = __shfl(area, src), o = __shfl(goff, src);
            const uint32_t sx0 = __shfl(rx0, src), sy0 = __shfl(ry0, src), sw = __shfl(rw, src);
            const float smx = __shfl(rec.q0.x, src), smy = __shfl(rec.q0.y, src), sa = __shfl(rec.q0.z, src),
                        sb = __shfl(rec.q0.w, src), sc = __shfl(rec.q1.x, src), stau = __shfl(tau, src);
            float part[16];
#pragma unroll
            for (int k = 0; k < 16; k++) part[k] = 0.0f;
            for (uint32_t k = lane; k < a; k += 64) {
                const uint32_t tx = sx0 + k % sw, ty = sy0 + k / sw;
                if (cam.exact_cull && !tile_may_touch(smx, smy, sa, sb, sc, stau, (int)tx * GSR_TILE, (int)ty * GSR_TILE))
                    continue;
                const float4* row = inst.rows + (size_t)GSR_ROW_F4(channels) * (o + k);
                const float4 f0 = row[0], f1 = row[1], f2 = row[2];
                part[0] += f0.x; part[1] += f0.y; part[2] += f0.z; part[3] += f0.w;
                part[4] += f1.x; part[5] += f1.y; part[6] += f1.z; part[7] += f1.w;
                part[8] += f2.x; part[9] += f2.y;
                if (channels > 5) {
                    const float4 f3 = row[3];
                    part[10] += f2.z; part[11] += f2.w; part[12] += f3.x;
                }
            }
            const float total = gsr::wave_reduce_transposed<13>(part, lb);
            if (gsr::wave_reduce_writer(lane)) bounce[wave][slot] = total;
            __builtin_amdgcn_wave_barrier();
            if (lane == src) {
#pragma unroll
                for (int k = 0; k < 13; k++) acc[k] = bounce[wave][k];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // FUSED: the gradients of this Gaussian stay here (zeros for a culled one)
    float f_vmean[3] = {0, 0, 0}, f_vs[3] = {0, 0, 0}, f_vq[4] = {0, 0, 0, 0}, f_vopac = 0.0f;
    float f_vc[3] = {0, 0, 0}, f_b[16];
    int f_nb = 0;  // SH bands carrying a gradient: 0 for a culled Gaussian
#pragma unroll
    for (int k = 0; k < 16; k++) f_b[k] = 0.0f;
    if (i < n) {
        if (!visible) {
            vmean2d_out[i] = make_float2(0.0f, 0.0f);
            if constexpr (!FUSED) {
#pragma unroll
                for (int c = 0; c < 3; c++) { vmeans[3 * i + c] = 0.0f; vscales[3 * i + c] = 0.0f; }
                vrots[i] = make_float4(0, 0, 0, 0);
                vopac[i] = 0.0f;
                if (vcolors) { vcolors[3 * i] = 0.0f; vcolors[3 * i + 1] = 0.0f; vcolors[3 * i + 2] = 0.0f; }
                // (else: f_nb = 0 — the cooperative store below writes this Gaussian's zeros)
            }
        } else {
            const float4 a0 = make_float4(acc[0], acc[1], acc[2], acc[3]);
            const float4 a1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
            const float2 vm2 = make_float2(acc[8], acc[9]);
            vmean2d_out[i] = vm2;  // gstate.∇means_2d, read by densification (strategy.jl:85-86)
            const float4 g0 = rec.q0, g1 = rec.q1, g2 = rec.q2;
            if constexpr (FUSED) f_vopac = a0.w; else vopac[i] = a0.w;
            M33 R; float t[3];
            load_pose(cam, R, t);
            // ---- ∇project ----
            M22 Ci, vCi;
            Ci.m[0][0] = g0.z; Ci.m[1][0] = g0.w; Ci.m[0][1] = g0.w; Ci.m[1][1] = g1.x;
            vCi.m[0][0] = a1.x; vCi.m[1][0] = a1.y; vCi.m[0][1] = a1.y; vCi.m[1][1] = a1.z;
            M22 nC;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) nC.m[a][b] = -Ci.m[a][b];
            M22 vS2 = mul22(mul22(nC, vCi), Ci);  // ∇inverse (render.jl:383-385)
            const float p[3] = {p_in[0], p_in[1], p_in[2]};
            float mc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) mc[r] = (R.m[r][0] * p[0] + R.m[r][1] * p[1] + R.m[r][2] * p[2]) + t[r];
            const float4 q4 = q_in;
            float qn[4], inv_norm;
            M33 Rg = quat2rot(q4, qn, inv_norm);
            const float s[3] = {s_in[0], s_in[1], s_in[2]};
            M33 M;
            M33 Sigma = cov3d(Rg, s, M);
            M33 Sc = mul33(mul33(R, Sigma), tr33(R));
            Persp pr = persp_common(mc, cam);
            const float rz = pr.rz, rz2 = rz * rz, rz3 = rz2 * rz;
            const float* f = cam.focal;
            // ∇perspective_projection (projection.jl:289-353)
            float JtV[3][2];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) JtV[a][b] = pr.J[0][a] * vS2.m[0][b] + pr.J[1][a] * vS2.m[1][b];
            M33 vSc;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) vSc.m[a][b] = JtV[a][0] * pr.J[0][b] + JtV[a][1] * pr.J[1][b];
            float A[2][3], B[2][3], vJ[2][3];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    A[a][b] = vS2.m[a][0] * pr.J[0][b] + vS2.m[a][1] * pr.J[1][b];
                    B[a][b] = vS2.m[0][a] * pr.J[0][b] + vS2.m[1][a] * pr.J[1][b];
                }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    float u = A[a][0] * Sc.m[b][0] + A[a][1] * Sc.m[b][1] + A[a][2] * Sc.m[b][2];
                    float v = B[a][0] * Sc.m[0][b] + B[a][1] * Sc.m[1][b] + B[a][2] * Sc.m[2][b];
                    vJ[a][b] = u + v;
                }
            float vx = f[0] * rz * vm2.x;
            float vy = f[1] * rz * vm2.y;
            float vz = -rz2 * (f[0] * mc[0] * vm2.x + f[1] * mc[1] * vm2.y);
            float rx = mc[0] * rz, ry = mc[1] * rz;
            if (-pr.lim_neg[0] <= rx && rx <= pr.lim[0]) vx += -f[0] * rz2 * vJ[0][2];
            else vz += -f[0] * rz3 * vJ[0][2] * pr.txy[0];
            if (-pr.lim_neg[1] <= ry && ry <= pr.lim[1]) vy += -f[1] * rz2 * vJ[1][2];
            else vz += -f[1] * rz3 * vJ[1][2] * pr.txy[1];
            vz += -f[0] * rz2 * vJ[0][0] - f[1] * rz2 * vJ[1][1] + 2.0f * f[0] * pr.txy[0] * rz3 * vJ[0][2] +
                  2.0f * f[1] * pr.txy[1] * rz3 * vJ[1][2];
            float vmc[3] = {vx, vy, vz};
            if (channels > 3) vmc[2] = vmc[2] + a1.w;  // vdepth (projection.jl:218-222)
            // ∇pos_world_to_cam, ∇covar_world_to_cam
            float vmean[3];
#pragma unroll
            for (int c = 0; c < 3; c++) vmean[c] = R.m[0][c] * vmc[0] + R.m[1][c] * vmc[1] + R.m[2][c] * vmc[2];
            M33 vSigma = mul33(mul33(tr33(R), vSc), R);
            if (vR_out) {
                M33 vR0;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) vR0.m[r][c] = vmc[r] * p[c];
                M33 u = mul33(mul33(vSc, R), tr33(Sigma));
                M33 v = mul33(mul33(tr33(vSc), R), Sigma);
                M33 vR2 = add33(add33(vR0, u), v);
#pragma unroll
                for (int r = 0; r < 3; r++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) poseR[c * 3 + r] = fabsf(vR2.m[r][c]) > 1e-7f ? vR2.m[r][c] : 0.0f;
                    poset[r] = fabsf(vmc[r]) > 1e-7f ? vmc[r] : 0.0f;
                }
            }
            // normal channel (projection.jl:227-235)
            M33 vRg;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) vRg.m[a][b] = 0.0f;
            if (channels > 5) {
                const float4 a2 = make_float4(acc[10], acc[11], acc[12], 0.0f);
                float nn[3]; int k; float sg;
                gaussian_normal(R, Rg, s, mc, nn, k, sg);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    float g = R.m[0][r] * a2.x + R.m[1][r] * a2.y + R.m[2][r] * a2.z;
                    float v = sg * g;
                    vRg.m[r][0] = k == 0 ? v : 0.0f;
                    vRg.m[r][1] = k == 1 ? v : 0.0f;
                    vRg.m[r][2] = k == 2 ? v : 0.0f;
                }
            }
            // ∇quat_scale_to_cov (render.jl:302-320) + ∇unnorm_quat2rot (render.jl:335-366): in float64 from the raw inputs
            // (scales_rots_bwd_f64 above — the fp32 chain loses a needle's thin eigen-direction).  F32CHAIN keeps the
            // reference's fp32 expression trees (gsr_config.grad_precision = GSR_GRAD_FP32_REFERENCE).
            float vs[3], vq[4];
            if constexpr (!F32CHAIN) {
                scales_rots_bwd_f64(cam, R, t, p, q4, s, a1.x, a1.y, a1.z, vRg, vs, vq);
                (void)vSigma; (void)M; (void)qn; (void)inv_norm;
            } else {
                M33 S;
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) S.m[a][b] = 0.0f;
                S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
                M33 vM = mul33(add33(vSigma, tr33(vSigma)), M);
                M33 vRq = add33(mul33(vM, S), vRg);
#pragma unroll
                for (int c = 0; c < 3; c++)
                    vs[c] = Rg.m[0][c] * vM.m[0][c] + Rg.m[1][c] * vM.m[1][c] + Rg.m[2][c] * vM.m[2][c];
                const float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
#define V(i_, j_) vRq.m[(i_) - 1][(j_) - 1]
                float vqn[4];
                vqn[0] = 2.0f * (x * (V(3, 2) - V(2, 3)) + y * (V(1, 3) - V(3, 1)) + z * (V(2, 1) - V(1, 2)));
                vqn[1] = 2.0f * (-2.0f * x * (V(2, 2) + V(3, 3)) + y * (V(2, 1) + V(1, 2)) + z * (V(3, 1) + V(1, 3)) +
                                 w * (V(3, 2) - V(2, 3)));
                vqn[2] = 2.0f * (x * (V(2, 1) + V(1, 2)) - 2.0f * y * (V(1, 1) + V(3, 3)) + z * (V(3, 2) + V(2, 3)) +
                                 w * (V(1, 3) - V(3, 1)));
                vqn[3] = 2.0f * (x * (V(3, 1) + V(1, 3)) + y * (V(3, 2) + V(2, 3)) - 2.0f * z * (V(1, 1) + V(2, 2)) +
                                 w * (V(2, 1) - V(1, 2)));
#undef V
                float dq = vqn[0] * qn[0] + vqn[1] * qn[1] + vqn[2] * qn[2] + vqn[3] * qn[3];
#pragma unroll
                for (int k = 0; k < 4; k++) vq[k] = (vqn[k] - dq * qn[k]) * inv_norm;
            }
            if constexpr (FUSED) {
#pragma unroll
                for (int k = 0; k < 4; k++) f_vq[k] = vq[k];
#pragma unroll
                for (int c = 0; c < 3; c++) f_vs[c] = vs[c];
            } else {
                vrots[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
#pragma unroll
                for (int c = 0; c < 3; c++) vscales[3 * i + c] = vs[c];
            }

            // ---- ∇SH ----
            // coefficients of band k >= 1 (band 0 has no directional gradient)
            const float* sh = FUSED ? TS.rest + (size_t)3 * (K - 1) * i : shs + (size_t)3 * K * i + 3;
            const uint32_t clamp_bits = __float_as_uint(g2.y);
            float vc[3] = {a0.x * (1.0f - (float)(clamp_bits & 1u)), a0.y * (1.0f - (float)((clamp_bits >> 1) & 1u)),
                           a0.z * (1.0f - (float)((clamp_bits >> 2) & 1u))};
            float d0[3] = {p[0] - cam.center[0], p[1] - cam.center[1], p[2] - cam.center[2]};
            float inv = 1.0f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            const float dx = d0[0] * inv, dy = d0[1] * inv, dz = d0[2] * inv;
            const float dir[3] = {dx, dy, dz};
            float b[16];
            sh_basis<DEG>(dir, b);
            constexpr int NB = (DEG + 1) * (DEG + 1);
            if constexpr (FUSED) {
                f_nb = NB;
#pragma unroll
                for (int c = 0; c < 3; c++) f_vc[c] = vc[c];
#pragma unroll
                for (int k = 0; k < NB; k++) f_b[k] = b[k];
            } else if (vcolors) {
                // factored form for the multi-view exchange: ∇shs of a view is the outer product
                // basis(dir) x vc, so 3 floats per Gaussian travel instead of 3K (sh_grad_views_kernel
                // rebuilds Σ_views on every rank)
                vcolors[3 * i] = vc[0]; vcolors[3 * i + 1] = vc[1]; vcolors[3 * i + 2] = vc[2];
            } else {
                // ∇shs = basis x vc is stored by the whole workgroup below (coalesced float4 rows instead of 48
                // dword stores 192 bytes apart per lane)
                f_nb = NB;
#pragma unroll
                for (int c = 0; c < 3; c++) f_vc[c] = vc[c];
#pragma unroll
                for (int k = 0; k < NB; k++) f_b[k] = b[k];
            }
            float dcx[3] = {0, 0, 0}, dcy[3] = {0, 0, 0}, dcz[3] = {0, 0, 0};
#define SHC(k_, c_) sh[3 * ((k_) - 1) + (c_)]
            if (DEG > 0) {
                const float x = dx, y = dy, z = dz;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dcx[c] = -SH1 * SHC(3, c); dcy[c] = -SH1 * SHC(1, c); dcz[c] = SH1 * SHC(2, c);
                }
                if (DEG > 1) {
                    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        dcx[c] = dcx[c] + SH2C1 * y * SHC(4, c) + SH2C3 * 2.0f * -x * SHC(6, c) +
                                 SH2C4 * z * SHC(7, c) + SH2C5 * 2.0f * x * SHC(8, c);
                        dcy[c] = dcy[c] + SH2C1 * x * SHC(4, c) + SH2C2 * z * SHC(5, c) +
                                 SH2C3 * 2.0f * -y * SHC(6, c) + SH2C5 * 2.0f * -y * SHC(8, c);
                        dcz[c] = dcz[c] + SH2C2 * y * SHC(5, c) + SH2C3 * 4.0f * z * SHC(6, c) + SH2C4 * x * SHC(7, c);
                    }
                    if (DEG > 2) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            dcx[c] = dcx[c] + SH3C1 * SHC(9, c) * 3.0f * 2.0f * xy + SH3C2 * SHC(10, c) * yz +
                                     SH3C3 * SHC(11, c) * -2.0f * xy + SH3C4 * SHC(12, c) * -3.0f * 2.0f * xz +
                                     SH3C5 * SHC(13, c) * (-3.0f * x2 + 4.0f * z2 - y2) +
                                     SH3C6 * SHC(14, c) * 2.0f * xz + SH3C7 * SHC(15, c) * 3.0f * (x2 - y2);
                            dcy[c] = dcy[c] + SH3C1 * SHC(9, c) * 3.0f * (x2 - y2) + SH3C2 * SHC(10, c) * xz +
                                     SH3C3 * SHC(11, c) * (-3.0f * y2 + 4.0f * z2 - x2) +
                                     SH3C4 * SHC(12, c) * -3.0f * 2.0f * yz + SH3C5 * SHC(13, c) * -2.0f * xy +
                                     SH3C6 * SHC(14, c) * -2.0f * yz + SH3C7 * SHC(15, c) * -3.0f * 2.0f * xy;
                            dcz[c] = dcz[c] + SH3C2 * SHC(10, c) * xy + SH3C3 * SHC(11, c) * 4.0f * 2.0f * yz +
                                     SH3C4 * SHC(12, c) * 3.0f * (2.0f * z2 - x2 - y2) +
                                     SH3C5 * SHC(13, c) * 4.0f * 2.0f * xz + SH3C6 * SHC(14, c) * (x2 - y2);
                        }
                    }
                }
            }
#undef SHC
            float vdir[3];
            vdir[0] = dcx[0] * vc[0] + dcx[1] * vc[1] + dcx[2] * vc[2];
            vdir[1] = dcy[0] * vc[0] + dcy[1] * vc[1] + dcy[2] * vc[2];
            vdir[2] = dcz[0] * vc[0] + dcz[1] * vc[1] + dcz[2] * vc[2];
            // ∇normalize (spherical_harmonics.jl:174-181)
            float s2 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
            float inv_s = 1.0f / sqrtf(s2 * s2 * s2);
            float vmsh[3];
            vmsh[0] = ((s2 - d0[0] * d0[0]) * vdir[0] - d0[1] * d0[0] * vdir[1] - d0[2] * d0[0] * vdir[2]) * inv_s;
            vmsh[1] = (-d0[0] * d0[1] * vdir[0] + (s2 - d0[1] * d0[1]) * vdir[1] - d0[2] * d0[1] * vdir[2]) * inv_s;
            vmsh[2] = (-d0[0] * d0[2] * vdir[0] - d0[1] * d0[2] * vdir[1] + (s2 - d0[2] * d0[2]) * vdir[2]) * inv_s;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if constexpr (FUSED) f_vmean[c] = vmean[c] + vmsh[c]; else vmeans[3 * i + c] = vmean[c] + vmsh[c];
            }
        }
    }
    // The SH part of the gradient is the outer product basis x colour cotangent: its factors wait in LDS and the whole
    // workgroup writes (FUSED: updates) the workgroup's contiguous slice element-major, as full cache lines.
    __shared__ float tail_b[256][17];  // SH basis of each Gaussian of the workgroup (odd stride: no bank conflicts)
    __shared__ float tail_vc[256][3];  // its colour cotangent (clamp mask applied)
    __shared__ int tail_nb[256];       // SH bands carrying a gradient (0: culled Gaussian)
    const int i0 = blockIdx.x * 256;
    const int cnt = min(256, n - i0);
    const int K3 = 3 * K;
    if (FUSED || !vcolors) {
#pragma unroll
        for (int k = 0; k < 16; k++) tail_b[threadIdx.x][k] = f_b[k];
#pragma unroll
        for (int c = 0; c < 3; c++) tail_vc[threadIdx.x][c] = f_vc[c];
        tail_nb[threadIdx.x] = f_nb;
    }
    if constexpr (!FUSED) {
        if (!vcolors) {
            __syncthreads();
            const uint32_t inv = (1u << 20) / (uint32_t)K3 + 1u;  // e / K3 == (e * inv) >> 20 for e < 2^20 / K3 (K3 <= 48, e < 256 K3)
            float* __restrict__ dst = vshs + (size_t)i0 * K3;
            const int total = cnt * K3;
            auto value = [&](int e) {
                const int il = (int)(((uint32_t)e * inv) >> 20);
                const int j = e - il * K3;
                const int k = j / 3, c = j - 3 * k;
                return k < tail_nb[il] ? tail_b[il][k] * tail_vc[il][c] : 0.0f;
            };
            const int total4 = (((uintptr_t)dst & 15) == 0) ? total >> 2 : 0;
            for (int f = threadIdx.x; f < total4; f += 256)
                reinterpret_cast<float4*>(dst)[f] = make_float4(value(4 * f), value(4 * f + 1), value(4 * f + 2), value(4 * f + 3));
            for (int e = 4 * total4 + threadIdx.x; e < total; e += 256) dst[e] = value(e);
        }
    }
    if constexpr (FUSED) {
        if (i < n) gsr::tail_gauss_apply(TS, i, f_vmean, f_vopac, f_vs, f_vq);
        __syncthreads();
        // (g(il, k, c): the SH-coefficient gradient of Gaussian il of the workgroup, band k, channel c)
        auto grad = [&](int il, int k, int c) { return k < tail_nb[il] ? tail_b[il][k] * tail_vc[il][c] : 0.0f; };
        auto sh_group = [&](float* __restrict__ th, float* __restrict__ mu, float* __restrict__ nu, int R, int k0,
                            const gsr::AdamHyper& hy) { tail_sh_group(TS, i0, cnt, K3, th, mu, nu, R, k0, hy, grad); };
        sh_group(TS.dc, TS.dc_mu, TS.dc_nu, 3, 0, TS.h_dc);
        if (K > 1) sh_group(TS.rest, TS.rest_mu, TS.rest_nu, 3 * (K - 1), 1, TS.h_rest);
    }
    if (vR_out) {  // projection.jl:243-256: thresholded per Gaussian, then summed
        // wave sum -> workgroup sum in LDS -> ONE atomic per workgroup and component (the 12
        // destinations are shared by every workgroup; per-wave atomics would serialise 4x longer)
        __shared__ float pose_red[4][12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            float v = k < 9 ? poseR[k] : poset[k - 9];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if ((threadIdx.x & 63) == 0) pose_red[threadIdx.x >> 6][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const int k = threadIdx.x;
            const float v = pose_red[0][k] + pose_red[1][k] + pose_red[2][k] + pose_red[3][k];
            if (v != 0.0f) atomicAdd(k < 9 ? &vR_out[k] : &vt_out[k - 9], v);
        }
    }
}

// ∇shs of a batch of views from the factored per-view colour cotangents:
//   vshs[i, k, c] = Σ_v basis_k(normalize(mean_i - center_v)) · vc_v[i, c]
// — exactly the ∇spherical_harmonics! coefficient gradient (spherical_harmonics.jl:32-37) of each
// view, summed over views in ascending view order on every rank.  The exchange then carries
// 3 floats per (Gaussian, view) instead of an all-reduce over 3K floats per Gaussian.
template <int DEG>
__global__ __launch_bounds__(256) void sh_grad_views_kernel(int n, int K, int n_views, const float* __restrict__ centers,
                                                            const float* __restrict__ means,
                                                            const float* __restrict__ vc_all,
                                                            float* __restrict__ vshs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float acc[3 * NB];
#pragma unroll
    for (int k = 0; k < 3 * NB; k++) acc[k] = 0.0f;
    if (i < n) {
        const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        for (int v = 0; v < n_views; v++) {
            const float* vcp = vc_all + ((size_t)v * n + i) * 3;
            const float vc[3] = {vcp[0], vcp[1], vcp[2]};
            if (vc[0] == 0.0f && vc[1] == 0.0f && vc[2] == 0.0f) continue;  // culled in this view (or zero cotangent)
            float d0[3] = {p[0] - centers[3 * v], p[1] - centers[3 * v + 1], p[2] - centers[3 * v + 2]};
            float inv = 1.0f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            const float dir[3] = {d0[0] * inv, d0[1] * inv, d0[2] * inv};
            float b[16];
            sh_basis<DEG>(dir, b);
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc[3 * k + c] = acc[3 * k + c] + b[k] * vc[c];
        }
    }
    // The rows leave through LDS: a lane's 3K floats are 12K bytes apart from its neighbour's, so storing them
    // directly is 3K store instructions of 64 different cache lines each; the workgroup's slice is contiguous and
    // goes out as coalesced float4 (same lesson as pergauss_bwd's ∇shs store).
    __shared__ float stage[256 * (3 * NB + 1)];  // odd stride: conflict-free
    constexpr int ST = 3 * NB + 1;
#pragma unroll
    for (int k = 0; k < 3 * NB; k++) stage[threadIdx.x * ST + k] = acc[k];
    __syncthreads();
    const int i0 = blockIdx.x * 256, cnt = min(256, n - i0), K3 = 3 * K;
    float* __restrict__ dst = vshs + (size_t)i0 * K3;
    const int total = cnt * K3;
    const uint32_t inv = (1u << 20) / (uint32_t)K3 + 1u;  // e / K3 == (e * inv) >> 20 for e < 2^20 / K3 (K3 <= 48, e < 256 K3)
    auto value = [&](int e) {
        const int il = (int)(((uint32_t)e * inv) >> 20);
        const int j = e - il * K3;
        return j < 3 * NB ? stage[il * ST + j] : 0.0f;  // bands above the active degree: zeros
    };
    const int total4 = (((uintptr_t)dst & 15) == 0) ? total >> 2 : 0;
    for (int f = threadIdx.x; f < total4; f += 256)
        reinterpret_cast<float4*>(dst)[f] = make_float4(value(4 * f), value(4 * f + 1), value(4 * f + 2), value(4 * f + 3));
    for (int e = 4 * total4 + threadIdx.x; e < total; e += 256) dst[e] = value(e);
}

// The multi-GPU trainer step made self-contained (training.jl:768-779): after the exchange of the
// factored arena — the 11·N small gradients all-reduced, the per-view colour cotangents all-gathered — ONE pass rebuilds
// Σ_v basis(dir_v) ⊗ vc_v per Gaussian (as sh_grad_views_kernel), keeps it in LDS instead of writing the 3K·N-float
// ∇shs, and applies the trainer tail right there: pullback of the functor prologue + the six NU.Adam updates + the
// activated copies of the next forward (adam_math.h / tail_sh_group: the definitions gsr_trainer_tail_step and the
// backward's fused epilogue use).  Bit-identical θ, μ, ν to gsr_sh_grad_from_views followed by gsr_trainer_tail_step;
// 192 B/Gaussian of gradient writes and reads less.  `S.points` are the means the directions are taken from; they are
// read before they are updated (same thread).
template <int DEG>
__global__ __launch_bounds__(256) void sh_views_tail_kernel(int n, int K, int n_views, const float* __restrict__ centers,
                                                            const float* __restrict__ vc_all,
                                                            const float* __restrict__ vmeans,
                                                            const float* __restrict__ vopac_act,
                                                            const float* __restrict__ vscales_act,
                                                            const float* __restrict__ vrot, gsr::TailState S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int ST = 3 * NB + 1;                 // odd stride: conflict-free
    __shared__ float stage[256 * ST];
    float acc[3 * NB];
#pragma unroll
    for (int k = 0; k < 3 * NB; k++) acc[k] = 0.0f;
    if (i < n) {
        const float p[3] = {S.points[3 * (size_t)i], S.points[3 * (size_t)i + 1], S.points[3 * (size_t)i + 2]};
        for (int v = 0; v < n_views; v++) {
            const float* vcp = vc_all + ((size_t)v * n + i) * 3;
            const float vc[3] = {vcp[0], vcp[1], vcp[2]};
            if (vc[0] == 0.0f && vc[1] == 0.0f && vc[2] == 0.0f) continue;  // culled in this view (or zero cotangent)
            float d0[3] = {p[0] - centers[3 * v], p[1] - centers[3 * v + 1], p[2] - centers[3 * v + 2]};
            float inv = 1.0f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            const float dir[3] = {d0[0] * inv, d0[1] * inv, d0[2] * inv};
            float b[16];
            sh_basis<DEG>(dir, b);
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc[3 * k + c] = acc[3 * k + c] + b[k] * vc[c];
        }
    }
#pragma unroll
    for (int k = 0; k < 3 * NB; k++) stage[threadIdx.x * ST + k] = acc[k];
    if (i < n) {
        const float vm[3] = {vmeans[3 * (size_t)i], vmeans[3 * (size_t)i + 1], vmeans[3 * (size_t)i + 2]};
        const float vs[3] = {vscales_act[3 * (size_t)i], vscales_act[3 * (size_t)i + 1], vscales_act[3 * (size_t)i + 2]};
        const float vq[4] = {vrot[4 * (size_t)i], vrot[4 * (size_t)i + 1], vrot[4 * (size_t)i + 2], vrot[4 * (size_t)i + 3]};
        gsr::tail_gauss_apply(S, i, vm, vopac_act[i], vs, vq);
    }
    __syncthreads();
    const int i0 = blockIdx.x * 256, cnt = min(256, n - i0), K3 = 3 * K;
    auto grad = [&](int il, int k, int c) { return k < NB ? stage[il * ST + 3 * k + c] : 0.0f; };  // bands above the degree: 0
    tail_sh_group(S, i0, cnt, K3, S.dc, S.dc_mu, S.dc_nu, 3, 0, S.h_dc, grad);
    if (K > 1) tail_sh_group(S, i0, cnt, K3, S.rest, S.rest_mu, S.rest_nu, 3 * (K - 1), 1, S.h_rest, grad);
}

// _update_stats! (src/strategy.jl:118-136): densification statistics from the side outputs of
// the last forward/backward pair (radii, ∇means_2d).
__global__ __launch_bounds__(256) void update_stats_kernel(int n, const int32_t* __restrict__ radii,
                                                           const float2* __restrict__ vmean2d, float res_x,
                                                           float res_y, int32_t* __restrict__ max_radii,
                                                           float* __restrict__ accum, float* __restrict__ denom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (!(r > 0)) return;
    max_radii[i] = max(max_radii[i], r);
    const float2 g = vmean2d[i];
    const float gx = g.x * res_x * 0.5f, gy = g.y * res_y * 0.5f;
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

}  // namespace

void gsr_launch_update_stats(hipStream_t s, int n, const int32_t* radii, const float2* vmean2d, int width, int height,
                             int32_t* max_radii, float* accum, float* denom) {
    if (n <= 0) return;
    hipLaunchKernelGGL(update_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, radii, vmean2d, (float)width,
                       (float)height, max_radii, accum, denom);
}


// The aggregating form's LDS plan (agg_plan.h): whole grid or bands, 2 x 32-bit or 2 x 16-bit words.
namespace {
using AggPlan = gsr_agg::Plan;
inline AggPlan agg_plan(int grid_x, int grid_y, uint32_t max_pos) { return gsr_agg::plan(grid_x, grid_y, max_pos); }
}  // namespace

// `agg`: the aggregating form (else the direct one) — WHICH is gsr_policy.cpp's decision (gsr_policy_begin_view: the handle's
// pin, the process default, the scene / grid rule with the previous view's skew, or the form tuner's measurement).  Returns
// the form that ran (gsr_stats.preprocess_form = gsr_agg::form_code: the same number the policy announced).
int gsr_launch_preprocess(hipStream_t s, int n, int K, int degree, int channels, const float* means,
                          const float* scales, const float* rots, const float* opac, const float* shs, GsrCam cam,
                          GsrGeom geom, uint32_t* tile_count, uint32_t* n_visible, uint64_t* bins, uint32_t bin_cap,
                          int n_tiles, bool agg) {
    if (n <= 0) return 0;
    const float4* r4 = reinterpret_cast<const float4*>(rots);
    // The aggregating form keeps its counter words in LDS three times per CU: the whole grid up to ~21 500 tiles, bands of it
    // beyond (4K: two).
    const int n_words = (n_tiles + 2) / 2;
    const AggPlan pl = agg_plan(cam.grid_x, cam.grid_y, bin_cap);
    const uint32_t* no_start = nullptr;
    const dim3 agg_grid((n + kAggThreads - 1) / kAggThreads), agg_block(kAggThreads);
#define LAUNCH_AGG(D, W, B)                                                                                                 \
    hipLaunchKernelGGL((preprocess_kernel<D, kAggThreads, W, false, B>), agg_grid, agg_block, pl.lds, s, n, K, channels,    \
                       means, scales, r4, opac, shs, cam, geom, tile_count, n_visible, bins, bin_cap, n_words, no_start,    \
                       pl.n_bands, pl.band_rows)
#define LAUNCH(D)                                                                                                           \
    do {                                                                                                                    \
        if (agg && pl.n_bands > 1) { if (pl.w32) LAUNCH_AGG(D, true, true); else LAUNCH_AGG(D, false, true); }              \
        else if (agg) { if (pl.w32) LAUNCH_AGG(D, true, false); else LAUNCH_AGG(D, false, false); }                         \
        else                                                                                                                \
            hipLaunchKernelGGL((preprocess_kernel<D, 0, false, false, false>), dim3((n + 255) / 256), dim3(256), 0, s, n, K, \
                               channels, means, scales, r4, opac, shs, cam, geom, tile_count, n_visible, bins, bin_cap,     \
                               n_words, no_start, 1, cam.grid_y);                                                           \
    } while (0)
    switch (degree) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
#undef LAUNCH_AGG
    return gsr_agg::form_code(agg, pl);
}

// duplicate_with_keys! (utils.jl:85-120) as the SECOND pass of the compact binning mode (count -> scan -> scatter): the
// aggregating form of preprocess_kernel in its SCATTER instantiation — the records are read back, the walks replay the emitted
// masks, the positions handed out are arrival ranks inside tile_start[t] ... — so that a hot tile's instances meet in LDS
// instead of serialising on one global counter (round 4's per-lane kernel: 0.85 ms for one tile of 32 k instances, 2.4 ms
// on a 4K grid with 1 % of the tiles at 50 x density).  `max_list`: the longest tile list (the scan's total), which decides
// between 2 x 16-bit and 2 x 32-bit LDS words.
void gsr_launch_emit_compact(hipStream_t s, int n, GsrCam cam, GsrGeom geom, const uint32_t* tile_start, uint32_t* tile_fill,
                             uint64_t* keys, uint32_t max_list, uint32_t only_above) {
    if (n <= 0) return;
    const AggPlan pl = agg_plan(cam.grid_x, cam.grid_y, max_list);
    const int n_words = (cam.grid_x * cam.grid_y + 2) / 2;
    const dim3 grid((n + kAggThreads - 1) / kAggThreads), block(kAggThreads);
    const float* nf = nullptr;
    const float4* nq = nullptr;
#define LAUNCH_SC(W, B)                                                                                                            \
    hipLaunchKernelGGL((preprocess_kernel<0, kAggThreads, W, true, B>), grid, block, pl.lds, s, n, 0, 3, nf, nf, nq, nf, nf, cam,  \
                       geom, tile_fill, (uint32_t*)nullptr, keys, only_above, n_words, tile_start, pl.n_bands, pl.band_rows)
    if (pl.n_bands > 1) { if (pl.w32) LAUNCH_SC(true, true); else LAUNCH_SC(false, true); }
    else if (pl.w32) LAUNCH_SC(true, false);
    else LAUNCH_SC(false, false);
#undef LAUNCH_SC
}

void gsr_launch_pergauss_bwd(hipStream_t s, int n, int K, int degree, int channels, const float* means,
                             const float* scales, const float* rots, const float* shs, GsrCam cam, GsrGeom geom,
                             GsrInst inst, float2* vmean2d, float* vmeans, float* vshs, float* vopac,
                             float* vscales, float* vrots, float* vR, float* vt, float* vcolors, bool fp32_chain) {
    if (n <= 0) return;
    dim3 grid((n + 255) / 256), block(256);
    const float4* r4 = reinterpret_cast<const float4*>(rots);
    float4* vr4 = reinterpret_cast<float4*>(vrots);
    const gsr::TailState none{};
#define LAUNCH_C(D, F32)                                                                                              \
    hipLaunchKernelGGL((pergauss_bwd_kernel<D, false, F32>), grid, block, 0, s, n, K, channels, means, scales, r4, shs, \
                       cam, geom, inst, vmean2d, vmeans, vshs, vopac, vscales, vr4, vR, vt, vcolors, none)
#define LAUNCH(D) do { if (fp32_chain) LAUNCH_C(D, true); else LAUNCH_C(D, false); } while (0)
    switch (degree) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
#undef LAUNCH_C
}

void gsr_launch_pergauss_bwd_tail(hipStream_t s, int n, int K, int degree, int channels, GsrCam cam, GsrGeom geom,
                                  GsrInst inst, float2* vmean2d, const gsr::TailState& S, bool fp32_chain) {
    if (n <= 0) return;
    dim3 grid((n + 255) / 256), block(256);
    const float4* r4 = reinterpret_cast<const float4*>(S.rots);
#define LAUNCH_C(D, F32)                                                                                              \
    hipLaunchKernelGGL((pergauss_bwd_kernel<D, true, F32>), grid, block, 0, s, n, K, channels, S.points, S.scales_act, \
                       r4, (const float*)nullptr, cam, geom, inst, vmean2d, (float*)nullptr, (float*)nullptr,       \
                       (float*)nullptr, (float*)nullptr, (float4*)nullptr, (float*)nullptr, (float*)nullptr,        \
                       (float*)nullptr, S)
#define LAUNCH(D) do { if (fp32_chain) LAUNCH_C(D, true); else LAUNCH_C(D, false); } while (0)
    switch (degree) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
#undef LAUNCH_C
}

void gsr_launch_sh_views_tail(hipStream_t s, int n, int K, int degree, int n_views, const float* centers,
                              const float* vc_all, const float* vmeans, const float* vopac_act, const float* vscales_act,
                              const float* vrot, const gsr::TailState& S) {
    if (n <= 0) return;
    dim3 grid((n + 255) / 256), block(256);
#define LAUNCH(D) hipLaunchKernelGGL(sh_views_tail_kernel<D>, grid, block, 0, s, n, K, n_views, centers, vc_all, vmeans, \
                                     vopac_act, vscales_act, vrot, S)
    switch (degree) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
}

void gsr_launch_sh_grad_views(hipStream_t s, int n, int K, int degree, int n_views, const float* centers,
                              const float* means, const float* vc_all, float* vshs) {
    if (n <= 0) return;
    dim3 grid((n + 255) / 256), block(256);
#define LAUNCH(D) hipLaunchKernelGGL(sh_grad_views_kernel<D>, grid, block, 0, s, n, K, n_views, centers, means, vc_all, vshs)
    switch (degree) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
}
