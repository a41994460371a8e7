// Per-tile alpha compositing, forward and backward.
// Reference behaviour: src/rasterization/render.jl:1-130 (`render!`) and :132-286
// (`∇render!`); SURVEY.md A.8 / A.9.
//
// A tile's depth-sorted splat list is a contiguous slice of the packed stream written by
// tile_sort (three coalesced float4 planes) and carries a 16-bit row mask per instance.
//
// Forward (`composite_fwd_strip_kernel`): one wave64 per 8x8 pixel quadrant, lane <-> pixel.
// Per 64 instances one ballot over the quadrant masks gives the wave's work list; only those
// splats are staged in LDS and read back with wave-uniform (broadcast) ds_read_b128.
//
// Backward: instead of the reference's (C+6) global float atomics per (pixel, splat)
// (render.jl:242,275-282) ONE wave64 owns the whole tile (4 pixels per lane), reduces the
// partials of a splat over its 64 lanes with a transposed permlane-swap / DPP network
// (wave_reduce.h), and one 64-byte gradient ROW per (tile, splat) instance leaves the
// workgroup as plain stores; the per-Gaussian kernel sums a Gaussian's rows.
#include "gsr_kernels.h"
#include "wave_reduce.h"
#include "tile_sort_device.h"


namespace {

struct Bg { float v[8]; };

// wave64 ballot straight from the compare mask (HIP's __ballot goes through v_cndmask + v_cmp_ne)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The Gaussian exponent  sigma = b·dx·dy + ½(a·dx² + c·dy²)  (render.jl:90-91 and :226-227) — ONE definition, with
// explicitly rounded operations, used by the forward AND the backward kernel: the backward replays exactly the
// contributor set the forward blended (`sigma >= 0`, `alpha >= 1/255` and the `T' < 1e-4` stop are decided on
// bit-identical sigma / alpha in both passes, whatever the compiler's FMA contraction would have done to two
// differently written expressions).  The sorted stream carries ha = a/2 and hc = c/2 (exact halvings).
struct SigmaX { float hxx, bdx; };  // the dx-dependent part, shared by the pixels of a column
__device__ __forceinline__ SigmaX sigma_x(float ha, float b, float dx) {
    return SigmaX{__fmul_rn(ha, __fmul_rn(dx, dx)), __fmul_rn(b, dx)};
}
__device__ __forceinline__ float sigma_of(const SigmaX sx, float hc, float dy, float dy2) {
    return __fmaf_rn(sx.bdx, dy, __fmaf_rn(hc, dy2, sx.hxx));
}
__device__ __forceinline__ float alpha_of(float opacity, float G) { return fminf(0.99f, __fmul_rn(opacity, G)); }

// N splats staged in LDS: the three (four with a normal) float4 planes of their stream entries in ONE array, so that the
// per-visit reads of a wave-uniform entry are one address (16·j: a scalar shift + one v_mov) and compile-time immediate
// offsets (plane · 16·N) instead of one address computation per plane — 2 VALU instructions per visit less in loops that
// are VALU-issue-bound; planes (not 48-byte records) keep the lane-contiguous staging stores conflict-free.
template <int C, int N> struct LdsSplats {
    float4 q[C > 3 ? 4 : 3][N];  // (fourth plane: normal xyz (:rgbdn) + the blend threshold in w, :rgbd / :rgbdn)
    __device__ __forceinline__ const float4& operator()(int plane, int j) const { return q[plane][j]; }
    __device__ __forceinline__ float4& operator()(int plane, int j) { return q[plane][j]; }
};

// feature c of a staged splat: rgb | depth | 1 | normal  (rasterizer.jl:380-385)
template <int C>
__device__ __forceinline__ void unpack_features(const float4& s1, const float4& s2, const float4& s3, float f[C]) {
    f[0] = s1.z; f[1] = s1.w; f[2] = s2.x;
    if (C > 3) { f[3] = s2.z; f[4] = 1.0f; }
    if (C > 5) { f[5] = s3.x; f[6] = s3.y; f[7] = s3.z; }
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
// Forward: one wave64 per 8x8 quadrant (a square footprint is visited by 9.5 % fewer splats than a 16x4 strip).
// No workgroup barriers inside a batch, a quadrant stops as soon as ITS pixels have saturated, and a wave visits only
// the splats its quadrant-mask ballot selected.  Same per-pixel arithmetic in the same order as render.jl:82-117.
//
// The per-pixel state of a lane and the blend of the LDS-resident splats a ballot selected are shared by
//   * composite_fwd_strip_kernel: four independent single-wave workgroups per tile, each staging the splats its ballot
//     selected from the global stream into its own 64-entry LDS arrays (tiers of long lists; views in compact mode);
//   * sort_composite_fwd_kernel: the fused sort + forward, which composites straight from the 256-entry LDS chunk the
//     workgroup has just emitted — the stream in HBM is written for the backward and never read back here.
template <int C> struct FwdPixel {
    bool done;
    float T, unc;
    uint32_t last;
    float color[C];
};

template <int C>
__device__ __forceinline__ void fwd_pixel_init(FwdPixel<C>& p, bool inside) {
    p.done = !inside; p.T = 1.0f; p.unc = 0.0f; p.last = 0;
#pragma unroll
    for (int c = 0; c < C; c++) p.color[c] = 0.0f;
}

// Blend the LDS entries whose bits are set in `m` (bit k <-> entry e0 + k), front to back.  Plane 2 of an entry holds
// (third colour, 1-BASED LIST POSITION, depth, -): `last` is then a select between two VGPRs, not a v_mov of the scalar
// position + a select.  `ids`: Gaussian id of entry 0, 1, ... (only read for the covisibility output).
template <int C, bool AUX, int N>
__device__ __forceinline__ void fwd_blend_selected(FwdPixel<C>& p, unsigned long long m, const LdsSplats<C, N>& e, int e0, float fx, float fy,
                                                   const uint32_t* __restrict__ ids, uint8_t* __restrict__ covis) {
    while (m) {
        const int jb = __builtin_ctzll(m), j = e0 + jb;
        m &= m - 1;
        // (x = third colour, y = list position: ONE 64-bit LDS read — left as two fields of q[2] the compiler sinks
        // the position's read under EXEC = ok, which costs a masked region per visit)
        float4 c2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        c2 = e(2, j);  // (x = third colour, y = list position, z = blend threshold (:rgb) or depth)
        const float4 a = e(0, j), b = e(1, j);
        const float4 c3 = C > 3 ? e(C > 3 ? 3 : 0, j) : c2;
        const float dx = a.x - fx, dy = a.y - fy;
        const float sigma = sigma_of(sigma_x(a.z, a.w, dx), b.x, dy, __fmul_rn(dy, dy));
        const float alpha = alpha_of(b.y, __expf(-sigma));
        // Branch-free body: a lane that is done, or whose pixel this splat does not touch, blends
        // with weight 0 (what `continue`/`break` leave behind, render.jl:92-101).  (Handling the
        // saturating lanes in a wave-uniform rare path instead measured 10 % slower.)
        const float Tn = p.T * (1.0f - alpha);
        const bool small = Tn < 1e-4f;
        // sigma >= 0 && alpha >= 1/255 as one unsigned compare against the instance's threshold (tile_sort_device.h)
        const bool touch = __float_as_uint(sigma) < __float_as_uint(C == 3 ? c2.z : c3.w);
        bool ok = !p.done && touch;
        const bool stop = ok && small;
        p.done = p.done || stop;
        ok = ok != stop;  // stop implies ok: the xor stays on the scalar unit (`ok && !small` costs a second v_cmp)
        float f[C];
        unpack_features<C>(b, c2, c3, f);
        const float w = ok ? alpha * p.T : 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) p.color[c] += f[c] * w;
        if (AUX) {
            p.unc += w;
            if (covis && ok && p.T > 0.5f) covis[ids[jb]] = 1;
        }
        p.T = ok ? Tn : p.T;
        p.last = ok ? __float_as_uint(c2.y) : p.last;
    }
}

template <int C, bool AUX>
__device__ __forceinline__ void fwd_pixel_store(const FwdPixel<C>& p, bool inside, int px, int py, int W, const Bg& bg,
                                                float* __restrict__ image, uint32_t* __restrict__ n_contrib,
                                                float* __restrict__ final_T, float* __restrict__ uncert) {
    if (inside) {
        const size_t pi = (size_t)px + (size_t)W * py;
        final_T[pi] = p.T;
        n_contrib[pi] = p.last;
#pragma unroll
        for (int c = 0; c < C; c++) image[(size_t)C * pi + c] = p.color[c] + p.T * bg.v[c];
        if (AUX && uncert) uncert[pi] = p.unc;
    }
}

// One quadrant wave walking its tile's slice of the GLOBAL stream (e: its 64-entry LDS staging array).
template <int C, bool AUX>
__device__ __forceinline__ void composite_fwd_quadrant(int W, int H, int grid_x, int tile, int quad, int lane,
                                                       const uint32_t* __restrict__ tile_start, GsrStream stream,
                                                       const Bg& bg, float* __restrict__ image,
                                                       uint32_t* __restrict__ n_contrib, float* __restrict__ final_T,
                                                       const uint32_t* __restrict__ values_sorted,
                                                       uint8_t* __restrict__ covis, float* __restrict__ uncert,
                                                       LdsSplats<C, 64>& e) {
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const uint32_t strip_bits = 0x10000u << quad;  // the instance's quadrant bit (tile_mask.h)
    const int px = tile_x * GSR_TILE + 8 * (quad & 1) + (lane & 7), py = tile_y * GSR_TILE + 8 * (quad >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    const int to_do = (int)(end - start);
    FwdPixel<C> p;
    fwd_pixel_init<C>(p, inside);
    for (int base = 0; base < to_do; base += 64) {
        if (wave_ballot(!p.done) == 0ull) break;  // the whole quadrant has saturated
        const int jj = base + lane;
        float4 r2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (jj < to_do) r2 = stream.s2[start + jj];
        const bool cand = jj < to_do && (__float_as_uint(r2.w) & strip_bits) != 0u;
        const unsigned long long m = wave_ballot(cand);
        if (m == 0ull) continue;
        // (Fetching the wave-uniform splats with scalar loads straight from the stream — s_load_dwordx4,
        // one candidate ahead, no LDS — measured 2.6x slower: the scalar cache does not keep up.)
        __builtin_amdgcn_wave_barrier();  // previous batch's LDS reads are done (single wave, in order)
        if (cand) {
            e(0, lane) = stream.s0[start + jj];
            e(1, lane) = stream.s1[start + jj];
            r2.y = __uint_as_float((uint32_t)(jj + 1));  // the forward has no use for the gradient-row slot: list position
            e(2, lane) = r2;
            if (C > 3) e(C > 3 ? 3 : 0, lane) = stream.s3[start + jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fwd_blend_selected<C, AUX, 64>(p, m, e, 0, fx, fy, values_sorted + start + base, covis);
    }
    fwd_pixel_store<C, AUX>(p, inside, px, py, W, bg, image, n_contrib, final_T, uncert);
}

template <int C, bool AUX>
__global__ __launch_bounds__(64) void composite_fwd_strip_kernel(int W, int H, int grid_x,
                                                                 const uint32_t* __restrict__ tile_start,
                                                                 const uint32_t* __restrict__ tile_order,
                                                                 GsrStream stream, Bg bg, float* __restrict__ image,
                                                                 uint32_t* __restrict__ n_contrib,
                                                                 float* __restrict__ final_T,
                                                                 const uint32_t* __restrict__ values_sorted,
                                                                 uint8_t* __restrict__ covis,
                                                                 float* __restrict__ uncert, GsrTierLists tiers) {
    __shared__ LdsSplats<C, 64> e;
    // 1-D grid, workgroup id -> (launch slot, strip).  Workgroups are dealt round-robin to the 8
    // XCDs (each with its own L2): the 4 strips of a tile get ids that are equal mod 8, so a tile's
    // splat stream is fetched into ONE L2; slots follow tile_order (longest lists first).
    const int id = blockIdx.x, k = id >> 3;
    const int quad = k & 3;  // 8x8 quadrant (qx = quad & 1, qy = quad >> 1) of the tile
    const int slot = ((k >> 2) << 3) | (id & 7);
    int tile;
    if (tile_order) {  // every tile, in launch order
        if (slot >= grid_x * ((H + GSR_TILE - 1) / GSR_TILE)) return;
        tile = (int)tile_order[slot];
    } else {           // only the tiles of the scan's tier lists (the fused kernel did the others), longest tier first
        uint32_t b = (uint32_t)slot;
        if (b >= tiers.n_big + tiers.n_mid8 + tiers.n_mid4) return;
        const uint32_t* list = tiers.lists;
        if (b >= tiers.n_big) { b -= tiers.n_big; list = tiers.lists + tiers.n_tiles;
            if (b >= tiers.n_mid8) { b -= tiers.n_mid8; list = tiers.lists + 2 * (size_t)tiers.n_tiles; } }
        tile = (int)list[b];
        // beside the fused launch (gsr_forward): these few waves are the forward's critical path and share their SIMDs with it
        if (tiers.split_len) __builtin_amdgcn_s_setprio(3);
    }
    composite_fwd_quadrant<C, AUX>(W, H, grid_x, tile, quad, (int)threadIdx.x, tile_start, stream, bg, image,
                                   n_contrib, final_T, values_sorted, covis, uncert, e);
}

// Sort + forward of a tile in ONE workgroup (fixed-capacity bins; a tile with more than 1024 instances is left to the
// tier sorts and a forward launch over the tier lists).
// Wave 0 sorts the tile's keys in registers; then, 256 instances at a time, all four waves EMIT a chunk of the sorted
// list — record gather -> packed stream entry, stored to HBM for the backward AND kept in LDS — and each wave composites
// its 8x8 quadrant straight from that LDS chunk.  Round 2 read the entries back from HBM (the workgroup's own stores,
// four times: 950 MB counted against 322 MB algorithmic, because a line lives ~18 us in an L2 the sort's gathers stream
// through); now the forward never reads the stream, needs no workgroup-scope release/acquire on global memory between
// the two phases, and no per-batch staging copy.  tile_sort alone is bound by HBM (gather + stream write) and the
// compositing alone by VALU issue; as phases of independent workgroups they share the CU at the same time.  Launched
// BEFORE the host has read the scan's totals, like the sort's main pass was: every workgroup checks the totals against
// the capacity of the buffers and leaves everything untouched when this view needs more (the host then runs the
// separate sort and forward launches).
// KEEP = false (GSR_FORWARD_ONLY, the reference's non-AD branch rasterizer.jl:214-248): the emitted chunk lives in LDS only —
// the 52-68 bytes per instance of stream + id that nothing but the backward reads are not stored.
template <int C, bool AUX, bool KEEP>
// (8 waves per SIMD: the sort needs 66 VGPRs left to itself; at 64 it does not spill and an eighth workgroup fits the CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void sort_composite_fwd_kernel(int W, int H, int grid_x,
                                                                 const uint32_t* __restrict__ tile_start,
                                                                 const uint32_t* __restrict__ tile_order,
                                                                 uint32_t* __restrict__ tile_count,
                                                                 const uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                                 GsrGeom geom, GsrStream stream, Bg bg,
                                                                 float* __restrict__ image,
                                                                 uint32_t* __restrict__ n_contrib,
                                                                 float* __restrict__ final_T,
                                                                 uint32_t* __restrict__ values_sorted,
                                                                 uint32_t* __restrict__ ranges,
                                                                 uint8_t* __restrict__ covis, float* __restrict__ uncert,
                                                                 const uint32_t* __restrict__ totals,
                                                                 uint32_t cap_instances) {
    // instances resident in LDS at a time: one per thread and round (512 / 1024 — whole lists — cost 5 / 3 workgroups per CU
    // instead of 8: +13 % / +48 %, profiles/r03/experiments/small_ab_notes.txt)
    constexpr int CHUNK = 256;
    __shared__ uint32_t ids[1024];
    __shared__ LdsSplats<C, CHUNK> e;
    // [0] instances, [1] longest list.  Bins of >= 1024 keys hold every list this launch takes complete whatever the longest one
    // is (lists beyond the capacity are tier tiles; the host scatters their keys again) — smaller bins must hold them all
    if (totals[0] > cap_instances || (totals[1] > bin_cap && bin_cap < 1024u)) return;
    const int tile = (int)tile_order[blockIdx.x];  // launch order: longest lists first
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    const uint32_t n = end - start;
    if (tid == 0) {
        tile_count[tile] = 0u;  // counter ready for the next view
        // identify_tile_range! (utils.jl:56-78): empty tiles keep the (0,0) of the prior fill!
        ranges[2 * tile] = n ? start : 0u;
        ranges[2 * tile + 1] = n ? end : 0u;
    }
    if (n > 1024u) return;  // a tier launch's tile (sort and forward)
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int X0 = tile_x * GSR_TILE, Y0 = tile_y * GSR_TILE;
    const int quad = wave;  // 8x8 quadrant (qx = quad & 1, qy = quad >> 1) of the tile
    const uint32_t strip_bits = 0x10000u << quad;
    const int px = X0 + 8 * (quad & 1) + (lane & 7), py = Y0 + 8 * (quad >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    float fx = (float)px, fy = (float)py;
    // (opaque to the optimiser: left alone it re-converts px / py inside the per-visit loop to save two registers —
    // two more VALU instructions per visit in a VALU-bound loop)
    asm volatile("" : "+v"(fx), "+v"(fy));
    FwdPixel<C> p;
    fwd_pixel_init<C>(p, inside);
    if (n > 0) {
        // (the sorting wave rotates with the workgroup id: wave w of every workgroup sits on SIMD w of its CU, and a fixed
        // sorter would load one SIMD of four with all the sorting)
        if (wave == (int)(blockIdx.x & 3u)) gsr_sort::wave_sort_ids_any(ids, n, lane, bins + (size_t)tile * bin_cap);
        __syncthreads();
        for (uint32_t cbase = 0; cbase < n; cbase += CHUNK) {
            const uint32_t i = cbase + (uint32_t)tid;
            if (i < n) {
                const uint32_t id = ids[i];
                const gsr_sort::InstanceVals v = gsr_sort::instance_vals<C>(id, X0, Y0, geom);
                if (KEEP) {
                    const uint32_t pos = start + i;
                    values_sorted[pos] = id;
                    stream.s0[pos] = v.v0; stream.s1[pos] = v.v1; stream.s2[pos] = v.v2;
                    if (C > 3) stream.s3[pos] = v.v3;
                }
                e(0, tid) = v.v0; e(1, tid) = v.v1;
                // LDS copy: (third colour, 1-based list position, depth | :rgb blend threshold, footprint masks)
                e(2, tid) = make_float4(v.v2.x, __uint_as_float(i + 1u), v.v2.z, v.v2.w);
                if (C > 3) e(C > 3 ? 3 : 0, tid) = v.v3;
            }
            __syncthreads();
            const int cnt = (int)min((uint32_t)CHUNK, n - cbase);
            for (int b = 0; b < cnt; b += 64) {
                if (wave_ballot(!p.done) == 0ull) break;  // the whole quadrant has saturated
                const int k = b + lane;
                const bool cand = k < cnt && (__float_as_uint(e(2, k).w) & strip_bits) != 0u;
                const unsigned long long m = wave_ballot(cand);
                if (m == 0ull) continue;
                fwd_blend_selected<C, AUX, CHUNK>(p, m, e, b, fx, fy, ids + cbase + b, covis);
            }
            if (cbase + CHUNK < n) __syncthreads();  // the chunk is consumed before the next one overwrites it
        }
    }
    fwd_pixel_store<C, AUX>(p, inside, px, py, W, bg, image, n_contrib, final_T, uncert);
}

// ---------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------
// accumulator row per staged splat: [0..2] v rgb, [3] v opacity, [4..6] v conic,
// [7..8] v mean2d, [9] v depth (C>=5), [10..12] v normal (C==8)
template <int C> struct AccRow { static constexpr int N = C == 3 ? 9 : (C == 5 ? 10 : 13); static constexpr int STRIDE = N | 1; };

#ifndef GSR_BWD_MINWAVES
#define GSR_BWD_MINWAVES 1
#endif
#ifndef GSR_BWD_BATCH
#define GSR_BWD_BATCH 64
#endif
#ifndef GSR_BWD_PPL
#define GSR_BWD_PPL 4
#endif
constexpr int BWD_BATCH = GSR_BWD_BATCH;  // splats staged per round (LDS: one accumulator slab per wave)

// ACC — the backward's per-pixel arithmetic in its ACCURATE form (round 6; gsr_config.grad_precision = GSR_GRAD_ACCURATE /
// GSR_GRAD_FP32_REFERENCE): libm-accurate expf instead of v_exp_f32 on the rounded product sigma x log2(e), IEEE division
// instead of v_rcp_f32 for T / (1 - alpha) — what the reference's Julia source means by `exp` and `/` (render.jl:236-259).
// On needle-shaped splats the backward's sums cancel by up to the square of the 2-D axis ratio, and the few-ulp SYSTEMATIC
// errors of the two fast instructions — fed through the T recursion of hundreds of contributors — are what put ∇means of
// a 90 : 1 needle of radius 100 px (edge 8498) 4.1e-4 from float64: with ACC 2.3e-5, the oracle's own distance
// (profiles/r06/experiments/needle_exact_exp_div_variants.txt; the wave reduction round 5 suspected is NOT the source).
// Costs +12 % of this kernel, so it is a per-handle choice, not the default; gating it per instance on a "needle" bit of the
// stream was built and dropped: +2.5 % at config 3 (which has no needles), +9 % on the trained-like scene, and with
// ~1-ulp refinements instead of libm the gain drowned in last-bit noise (needle_gated_refinement_*.txt / .patch).
template <bool ACC> __device__ __forceinline__ float bwd_exp_neg(float sigma) { return ACC ? expf(-sigma) : __expf(-sigma); }
template <bool ACC> __device__ __forceinline__ float bwd_rcp(float x) { return ACC ? __fdiv_rn(1.0f, x) : __builtin_amdgcn_rcpf(x); }

// PPL = pixels per lane.  PPL == 1: 4 waves per tile, a wave owns a 16x4 strip.  PPL == 2:
// 2 waves per tile, a wave owns 16x8 pixels and lane l the pixels (x, y) and (x, y+4).
// PPL == 4 (default): ONE wave64 per tile, lane l owns (x, y), (x, y+4), (x, y+8), (x, y+12).
// The LDS reads, the cross-lane reduction and the row store — ~60 % of the instructions of a
// visited (strip, splat) pair — are paid once for 2x / 4x the pixels (measured at config 3:
// 1.115 / 0.975 / 0.917 ms for PPL = 1 / 2 / 4).
// LISTED: the tiles of one tier list of the scan (lists longer than GSR_BWD_SPLIT_LEN) — launched with PPL = 1, four
// waves per tile on a second stream next to the main PPL = 4 launch, which leaves those tiles out: one wave walking a
// list of 30 k instances is milliseconds long (real captures have such tiles; config 3 has none).
// BG0: the background is exactly (0, 0, 0) — the reference's default (rasterizer.jl:209) and what a trainer without a sky
// colour passes: the term -T_final/(1-α)·(bg·v) (render.jl:259) vanishes identically, and with it the per-pixel bgT state and
// one FMA per active visit.  WHO LAUNCHES WHAT (gsr_launch_composite_bwd, below; measured in
// profiles/r04/experiments/bwd_occupancy_ab.txt, A/B in one run):
//   C == 8 (:rgbdn): BG0 kernel WITH the rebuilt row coordinates (FY_REBUILD): 101 -> 94 VGPRs, four -> five waves per SIMD,
//                    0.907 -> 0.876 ms — launched whenever the background is zero;
//   C == 5 (:rgbd) : BG0 kernel WITHOUT the rebuilt coordinates: 88 -> 83 VGPRs, five waves as before, one FMA per active
//                    visit less, 0.720 -> 0.713 ms — launched whenever the background is zero (with the coordinates rebuilt it
//                    reaches 80 VGPRs = six waves and is 6 % SLOWER: 0.724 -> 0.770 ms);
//   C == 3 (:rgb)  : NEVER the BG0 kernel (74 -> 70 VGPRs = seven waves: 0.659 -> 0.676 ms; round 3's forced-occupancy probes:
//                    789 / 735 / 677 / 695 us at 4 / 5 / 6 / 7 waves) — <3, PPL, false, true> is not instantiated.
// VC: channels of the pixel cotangent that can be non-zero.  VC == 3 < C (the cotangent comes from the photometric loss head, which
// only sees features[1:3] — training.jl:656,684-685: depth / alpha / normal channels of vpixels are exact zeros and their feature
// gradients too): the pixels' cotangent state, the colour·v dot product and the reduction are the :rgb kernel's; the stream, the
// blend thresholds and the row layout stay the mode's.
template <int C, int PPL, bool LISTED, bool BG0, int VC = C, bool ACC = false>
__global__ __launch_bounds__(256 / PPL, GSR_BWD_MINWAVES) void composite_bwd_kernel(int W, int H, int grid_x,
                                                                const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ tile_order,
                                                                GsrStream stream, Bg bg,
                                                                const float* __restrict__ vpixels,
                                                                const uint32_t* __restrict__ n_contrib,
                                                                const float* __restrict__ final_T, GsrInst inst,
                                                                GsrTierLists tiers) {
    static_assert(VC == C || VC == 3, "all channels, or colour only");
    constexpr int NA = AccRow<VC>::N, ST = AccRow<VC>::STRIDE;
    constexpr int BB = BWD_BATCH, NT = 256 / PPL, NW = NT / 64, ROWS = 4 * PPL;
    // the long tiles are the critical path of the step and share their SIMDs with the main launch's waves: issue priority
    if (LISTED) __builtin_amdgcn_s_setprio(3);
    static_assert(BB <= NT && BB % 64 == 0, "one staging thread per splat");
    __shared__ float4 l0[BB], l1[BB], l2[BB];
    __shared__ float4 l3[C > 3 ? BB : 1];
    // One accumulator slab per wave: a wave stores its reduced partials with plain ds_write
    // (no LDS atomics: hipcc wraps those in a per-lane "atomic optimizer" loop), and a
    // per-wave bit mask records which rows it touched so nothing has to be zero-filled.
    __shared__ float lacc[NW][BB * ST];
    __shared__ unsigned long long lmask[NW][BB / 64];
    __shared__ int tile_last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t strip_bits = ((1u << ROWS) - 1u) << (ROWS * wave);  // this wave's pixel rows
    const gsr::LaneBits lane_bits(lane);
    const int red_slot = gsr::wave_reduce_index<NA>(lane);  // which partial this lane ends up holding
    const bool red_writer = gsr::wave_reduce_writer(lane);
#ifdef GSR_BWD_MFMA
    const gsr::RowColConstsM rowcol(lane);
#else
    const gsr::RowColConsts rowcol(lane);
#endif
    const gsr::RowColConstsD rowcol_d(lane);
    // main launch: 1-D grid in launch order, longest lists first; LISTED: the scan's three tier lists back to back,
    // longest tier first
    int tile;
    if (LISTED) {
        uint32_t b = blockIdx.x;
        const uint32_t* list = tiers.lists;                                              // lists > 8192
        if (b >= tiers.n_big) { b -= tiers.n_big; list = tiers.lists + tiers.n_tiles;    // (4096, 8192]
            if (b >= tiers.n_mid8) { b -= tiers.n_mid8; list = tiers.lists + 2 * (size_t)tiers.n_tiles; } }  // (1024, 4096]
        tile = (int)list[b];
    } else tile = (int)tile_order[blockIdx.x];
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int px = tile_x * GSR_TILE + (lane & 15);
    const int py0 = tile_y * GSR_TILE + ROWS * wave + (lane >> 4);
    const float fx = (float)px;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    if (end == start) return;
    if (!LISTED && end - start > tiers.split_len) return;  // the four-wave launch over the tier lists owns this tile

    // per-pixel state (PPL pixels per lane: rows py0 and py0 + 4)
    // (:rgbdn: the rows' y coordinates are rebuilt from the first one — fy0 + 4q, exact in fp32, so dy and sigma keep their
    //  bits — instead of living in PPL registers: one more add per visited group, three registers less)
    constexpr bool FY_REBUILD = C > 5 && PPL > 1;
    float fy[FY_REBUILD ? 1 : PPL], T[PPL], A[PPL], bgT[BG0 ? 1 : PPL], vp[PPL][VC];
    int last_contributor[PPL];
    int wave_last = 0;  // deepest list position any pixel of this wave blended
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int py = py0 + 4 * q;
        const bool inside = px < W && py < H;
        const size_t pi = (size_t)px + (size_t)W * py;
        if (!FY_REBUILD || q == 0) fy[FY_REBUILD ? 0 : q] = (float)py;
        const float T_final = inside ? final_T[pi] : 0.0f;
        T[q] = T_final;
        last_contributor[q] = inside ? (int)n_contrib[pi] : 0;
        float bg_dot = 0.0f;
#pragma unroll
        for (int c = 0; c < VC; c++) {
            vp[q][c] = inside ? vpixels[(size_t)C * pi + c] : 0.0f;
            bg_dot += bg.v[c] * vp[q][c];
        }
        if (!BG0) bgT[BG0 ? 0 : q] = -T_final * bg_dot;
        // The reference carries accum_rec[c], last_color[c], last_alpha per channel and forms
        //   vα = Σ_c (color[c] - accum_rec[c])·v[c]            (render.jl:245-252).
        // Only the dot product with the pixel cotangent is ever used, so the state is folded to
        // one scalar A = accum_rec·v with the same recurrence  A' = α·(color·v) + (1-α)·A.
        A[q] = 0.0f;
        wave_last = max(wave_last, last_contributor[q]);
    }

    // Splats behind every pixel's last contributor are skipped by each lane in the reference
    // (render.jl:223); start the back-to-front walk at the deepest one any pixel blended.
    if (tid == 0) tile_last_s = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
    if (lane == 0) atomicMax(&tile_last_s, wave_last);
    __syncthreads();
    const int tile_last = tile_last_s;  // process list positions tile_last-1 ... 0

    for (int base = 0; base < tile_last; base += BB) {
        const int cnt = min(BB, tile_last - base);
        __syncthreads();  // previous batch fully flushed
        if (tid < cnt) {
            const uint32_t idx = start + (uint32_t)(tile_last - 1 - base - tid);
            l0[tid] = stream.s0[idx];
            l1[tid] = stream.s1[idx];
            // staged as (third colour, footprint mask, depth, gradient-row slot): what the walk reads per splat is then
            // one aligned 64-bit LDS read in :rgb mode
            const float4 t2 = stream.s2[idx];
            l2[tid] = make_float4(t2.x, t2.w, t2.z, t2.y);
            if (C > 3) l3[tid] = stream.s3[idx];
        }
        __syncthreads();

        unsigned long long touched[BB / 64];
#pragma unroll
        for (int w = 0; w < BB / 64; w++) touched[w] = 0ull;
        float* const my = lacc[wave];
        for (int c0 = 0; c0 < cnt; c0 += 64) {
          const int jj = c0 + lane;
          const bool cand = jj < cnt && (__float_as_uint(l2[jj].y) & strip_bits) != 0u &&
                            (tile_last - 1 - base - jj) < wave_last;
          unsigned long long wl = wave_ballot(cand);  // splats whose footprint can touch this wave's rows
          while (wl) {
            const int j = c0 + __builtin_ctzll(wl);
            wl &= wl - 1;
            const int contributor = tile_last - 1 - base - j;  // 0-based position in the tile list
            const float4 a = l0[j], b = l1[j];
            float4 c2;  // (x: third colour, w: footprint mask, z: depth) as the stream has them
            const float4 t2 = l2[j];  // staged as (third colour, footprint mask, depth | :rgb blend threshold, slot)
            c2 = make_float4(t2.x, 0.0f, t2.z, t2.y);
            float4 t3 = t2;
            if (C > 3) t3 = l3[j];  // (normal xyz, blend threshold)
            const uint32_t thr_bits = __float_as_uint(C == 3 ? t2.z : t3.w);
            const float o = b.y;
            const float dx = a.x - fx;
            const SigmaX sx = sigma_x(a.z, a.w, dx);
            float f[C];
            unpack_features<C>(b, c2, t3, f);
            // A lane accumulates only  P = Σ G·vα,  U1 = Σ G·vα·dy,  U2 = Σ G·vα·dy²  and the
            // feature sums.  The conic / mean2d gradients (render.jl:262-272) are linear in
            // {dx²·P, dx·U1, U2, dx·P, U1}: those five are what the wave reduces, and the
            // per-instance flush applies the wave-uniform factors (-o/2, conic) once per row.
            float P = 0.0f, U1 = 0.0f, U2 = 0.0f, col[VC];
#pragma unroll
            for (int c = 0; c < VC; c++) col[c] = 0.0f;
            unsigned long long any_active = 0ull;
#ifndef GSR_BWD_NO_ROW_SKIP
            const uint32_t rowbits = __builtin_amdgcn_readfirstlane(__float_as_uint(c2.w)) >> (ROWS * wave);
#endif
#pragma unroll
            for (int q = 0; q < PPL; q++) {
#ifndef GSR_BWD_NO_ROW_SKIP
                // pixel rows 4q..4q+3 of this wave: untouched by the splat's footprint -> wave-uniform skip
                // (readfirstlane of the already-uniform bits: tells the compiler that this branch, and everything that
                // merges behind it — the ballot accumulator —, is scalar; without it the test and `any_active` were VALU)
                if (PPL > 1 && __builtin_amdgcn_readfirstlane((int)((rowbits >> (4 * q)) & 0xFu)) == 0) continue;
#endif
                const float fyq = FY_REBUILD ? fy[0] + (float)(4 * q) : fy[FY_REBUILD ? 0 : q];
                const float dy = a.y - fyq, dy2 = __fmul_rn(dy, dy);
                const float sigma = sigma_of(sx, b.x, dy, dy2);
                // the blend test is one unsigned compare of sigma against the instance's threshold; exp and alpha are only
                // computed for the lanes that pass (they run under EXEC = active)
                const bool c_live = contributor < last_contributor[q];
                const bool c_touch = __float_as_uint(sigma) < thr_bits;
                const bool active = c_live && c_touch;
                // ballots of the bare compares are their SGPR masks; the AND/OR runs on the scalar unit
                any_active |= wave_ballot(c_live) & wave_ballot(c_touch);
                if (active) {
                    const float G = bwd_exp_neg<ACC>(sigma);
                    const float alpha = alpha_of(o, G);
                    // T /= (1-α) and -T_final/(1-α) (render.jl:237,259) share one hardware reciprocal
                    const float rinv = bwd_rcp<ACC>(1.0f - alpha);
                    T[q] = T[q] * rinv;
                    const float fac = alpha * T[q];
                    float cv = f[0] * vp[q][0];
#pragma unroll
                    for (int c = 1; c < VC; c++) cv += f[c] * vp[q][c];
                    const float d = cv - A[q];                 // (color - accum_rec)·v
                    const float valpha = BG0 ? d * T[q] : d * T[q] + bgT[BG0 ? 0 : q] * rinv;
                    A[q] = A[q] + alpha * d;                   // α·cv + (1-α)·A for the next (nearer) splat
                    const float t = G * valpha;
                    P += t;
                    U1 += t * dy;
                    U2 += t * dy2;
#pragma unroll
                    for (int c = 0; c < VC; c++) col[c] += fac * vp[q][c];
                }
            }
            if (any_active == 0ull) continue;  // wave-uniform: none of this wave's pixels is touched
            // row of this splat in the wave's accumulator slab: j is wave-uniform, the multiply belongs on the scalar
            // unit (left to itself the compiler folds `j * ST + slot` into a quarter-rate v_mad_u64_u32 per store)
            int jrow;
            asm("s_mul_i32 %0, %1, %2" : "=s"(jrow) : "s"(j), "n"(ST));
            float* const my_row = my + jrow;
#pragma unroll
            for (int w = 0; w < BB / 64; w++)
                if ((j >> 6) == w) touched[w] |= 1ull << (j & 63);
            if (VC == 3) {
                // :rgb — reduce {P, U1, U2, rgb} over the 4 lanes of each pixel column first, apply the
                // column's dx weights, then finish over the 16 columns (wave_reduce.h: 5 swaps, not 8)
#ifdef GSR_BWD_MFMA
                const float total = gsr::wave_reduce_rowcol_rgb_mfma(P, U1, U2, col[0], col[1], col[2], dx, lane_bits, rowcol);
#else
                const float total = gsr::wave_reduce_rowcol_rgb(P, U1, U2, col[0], col[1], col[2], dx, lane_bits, rowcol);
#endif
                if (rowcol.slot >= 0) my_row[rowcol.slot] = total;
            } else if (C == 5) {
                // :rgbd (the reference's default training mode): the same row-then-column scheme with the depth sum
                // riding along — six swaps instead of the generic network's eight; feature 4 (constant 1) is not a parameter
                const float total = gsr::wave_reduce_rowcol_rgbd(P, U1, U2, col[0], col[1], col[2], col[VC > 3 ? 3 : 0], dx, lane_bits, rowcol_d);
                if (rowcol_d.slot >= 0) my_row[rowcol_d.slot] = total;
            } else {
                float part[16];
#pragma unroll
                for (int k = 0; k < 16; k++) part[k] = 0.0f;
                const float dxp = dx * P;
                part[0] = col[0]; part[1] = col[1]; part[2] = col[2];
                part[3] = P;         // Σ G·vα          -> v opacity
                part[4] = dx * dxp;  // Σ dx²·G·vα      -> v conic.x  (× -o/2 in the flush)
                part[5] = dx * U1;   // Σ dx·dy·G·vα    -> v conic.y
                part[6] = U2;        // Σ dy²·G·vα      -> v conic.z
                part[7] = dxp;       // Σ dx·G·vα   }   -> v mean2d = -o·(conic · these)
                part[8] = U1;        // Σ dy·G·vα   }
                if (VC > 3) part[9] = col[VC > 3 ? 3 : 0];  // depth feature; channel 4 (constant 1) is not a parameter
                if (VC > 5) { part[10] = col[VC > 5 ? 5 : 0]; part[11] = col[VC > 5 ? 6 : 0]; part[12] = col[VC > 5 ? 7 : 0]; }
                // transposed wave64 reduction: ~3·NA/2 + 6 VALU ops, then ONE ds_write for all NA sums
                const float total = gsr::wave_reduce_transposed<NA>(part, lane_bits);
                // (storing the sums straight into the global row when NW == 1 measured 3 % slower than
                // the LDS slab + coalesced 64-byte row stores below)
                if (red_writer) my_row[red_slot] = total;
            }
          }
        }
        if (lane == 0) {
#pragma unroll
            for (int w = 0; w < BB / 64; w++) lmask[wave][w] = touched[w];
        }
        __syncthreads();
        if (tid < cnt) {
            float r[NA];
#pragma unroll
            for (int k = 0; k < NA; k++) r[k] = 0.0f;
            bool any = false;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if ((lmask[w][tid >> 6] >> (tid & 63)) & 1ull) {
                    any = true;
#pragma unroll
                    for (int k = 0; k < NA; k++) r[k] += lacc[w][tid * ST + k];
                }
            }
            (void)any;
            {
                // one 64-byte gradient row per instance, plain stores, written for EVERY emitted
                // instance exactly once per backward (zeros when no pixel touched it), so the row
                // buffer needs no memset; the per-Gaussian kernel sums a Gaussian's rows in a fixed
                // order — no fp32 atomics (35 M per view before), bit-reproducible gradients
                const float4 a = l0[tid], b = l1[tid];
                const float mo = -b.y, mh = -0.5f * b.y;  // vσ = -o·G·vα (render.jl:260)
                float4* row = inst.rows + (size_t)GSR_ROW_F4(C) * __float_as_uint(l2[tid].w);  // Gaussian-major slot
                row[0] = make_float4(r[0], r[1], r[2], r[3]);
                row[1] = make_float4(mh * r[4], mh * r[5], mh * r[6], VC > 3 ? r[9 < NA ? 9 : 0] : 0.0f);
                // conic a = 2·ha, c = 2·hc (the stream carries the halves)
                row[2] = make_float4(mo * (2.0f * a.z * r[7] + a.w * r[8]), mo * (a.w * r[7] + 2.0f * b.x * r[8]),
                                     VC > 5 ? r[10 < NA ? 10 : 0] : 0.0f, VC > 5 ? r[11 < NA ? 11 : 0] : 0.0f);
                if (C > 5) row[3] = make_float4(VC > 5 ? r[12 < NA ? 12 : 0] : 0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
    // instances behind every pixel's last contributor were never staged: their rows are zero
    for (uint32_t p = start + (uint32_t)tile_last + tid; p < end; p += NT) {
        float4* row = inst.rows + (size_t)GSR_ROW_F4(C) * __float_as_uint(stream.s2[p].y);
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        row[0] = z; row[1] = z; row[2] = z;
        if (C > 5) row[3] = z;
    }
}

// ---------------------------------------------------------------------------------
// backward of LONG lists, split along their LENGTH (round 5)
// ---------------------------------------------------------------------------------
// A list of tens of thousands of instances is serial for whoever walks it: the hot-tile scene's critical path was ONE wave of
// a neighbour of the hot tile visiting 15 000 entries that touch its 16x4 strip (profiles/r05/experiments/long_tiles.txt) —
// more waves per tile by PIXELS (the PPL = 1 kernel above, four strips) do not help where one strip takes all the visits, and
// more waves in ONE workgroup stop at the four SIMDs of its CU (eight segment waves in one workgroup: 1.77 -> 1.24 ms only).
// Here a tile's list is cut into up to LONG_SEGS SEGMENTS, each walked by its own single-wave workgroup anywhere on the chip:
// a wave owns a contiguous run of list positions and ALL 256 pixels (four per lane, the main kernel's layout: row-then-column
// reduction, no cross-wave sums — each instance is reduced by exactly one wave and stored once).  The back-to-front recursions
// of a pixel (render.jl:237-258)
//     T <- T / (1 - alpha)           A <- alpha (colour . v) + (1 - alpha) A
// are affine in (T, A), so a segment's effect is (m, c) with m = prod (1 - alpha), c the A it leaves from A = 0:
//   PASS 1 (first launch): every wave walks its segment evaluating only the blend test, alpha and colour . v  ->  (m, c) per
//           pixel into `state` (global: [listed tile][segment][256 pixels][2]);
//   PASS 2 (second launch, same grid): wave g starts from  T = T_final / prod_{g' behind} m,  A = the composition of the
//           (m, c) behind it, and walks its segment again with the full gradient body.
// ~1.6 x the arithmetic of one walk, spread over the chip; no workgroup barrier anywhere.  The decisions (`bits(sigma) < X`,
// position < n_contrib) are the forward's, per pixel, as everywhere; what differs from the one-wave walk is the association of
// the products of (1 - alpha) across segment boundaries (last-bit level).
constexpr int LONG_SEGS = GSR_BWD_LONG_SEGS;

template <int C, bool BG0, int PASS>
__global__ __launch_bounds__(64) void composite_bwd_long_kernel(int W, int H, int grid_x,
                                                                const uint32_t* __restrict__ tile_start,
                                                                GsrStream stream, Bg bg,
                                                                const float* __restrict__ vpixels,
                                                                const uint32_t* __restrict__ n_contrib,
                                                                const float* __restrict__ final_T, GsrInst inst,
                                                                GsrTierLists tiers, float2* __restrict__ state) {
    constexpr int NA = AccRow<C>::N, ST = AccRow<C>::STRIDE, PPL = 4;
    __shared__ float4 l0[64], l1[64], l2[64];
    __shared__ float4 l3[C > 3 ? 64 : 1];
    __shared__ float my[64 * ST];
    __builtin_amdgcn_s_setprio(3);  // these few waves are the critical path of the step; they share SIMDs with the main launch
    const int lane = threadIdx.x;
    const uint32_t listed = blockIdx.x / LONG_SEGS;
    const int g = (int)(blockIdx.x % LONG_SEGS);  // segment; g = 0 is the BACK of the list
    const gsr::LaneBits lane_bits(lane);
    const int red_slot = gsr::wave_reduce_index<NA>(lane);
    const bool red_writer = gsr::wave_reduce_writer(lane);
    const gsr::RowColConsts rowcol(lane);
    const gsr::RowColConstsD rowcol_d(lane);
    int tile;
    {
        uint32_t b = listed;
        const uint32_t* list = tiers.lists;                                              // lists > 8192
        if (b >= tiers.n_big) { b -= tiers.n_big; list = tiers.lists + tiers.n_tiles;    // (4096, 8192]
            if (b >= tiers.n_mid8) { b -= tiers.n_mid8; list = tiers.lists + 2 * (size_t)tiers.n_tiles; } }  // (1024, 4096]
        tile = (int)list[b];
    }
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int px = tile_x * GSR_TILE + (lane & 15);
    const int py0 = tile_y * GSR_TILE + (lane >> 4);
    const float fx = (float)px;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    if (end == start) return;
    float fy[PPL], T[PPL], A[PPL], bgT[BG0 ? 1 : PPL], vp[PPL][C];
    int last_contributor[PPL];
    int tile_last = 0;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int py = py0 + 4 * q;
        const bool inside = px < W && py < H;
        const size_t pi = (size_t)px + (size_t)W * py;
        fy[q] = (float)py;
        const float T_final = inside ? final_T[pi] : 0.0f;
        T[q] = T_final;
        last_contributor[q] = inside ? (int)n_contrib[pi] : 0;
        float bg_dot = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            vp[q][c] = inside ? vpixels[(size_t)C * pi + c] : 0.0f;
            bg_dot += bg.v[c] * vp[q][c];
        }
        if (!BG0) bgT[BG0 ? 0 : q] = -T_final * bg_dot;
        A[q] = 0.0f;
        tile_last = max(tile_last, last_contributor[q]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tile_last = max(tile_last, __shfl_xor(tile_last, off));
    tile_last = __builtin_amdgcn_readfirstlane(tile_last);  // list positions tile_last-1 ... 0 are walked (the wave holds all 256 pixels)
    // segment g: positions [lo, hi), walked from hi - 1 down; whole 64-entry batches per segment
    const int seg = (((tile_last + LONG_SEGS - 1) / LONG_SEGS) + 63) & ~63;
    const int hi = tile_last - g * seg, lo = max(0, hi - seg);
    float2* const st_tile = state + (size_t)listed * LONG_SEGS * 256;
    if (PASS == 2 && g == 0) {
        // instances behind every pixel's last contributor are never staged: their rows are zero (once per tile)
        for (uint32_t p = start + (uint32_t)tile_last + lane; p < end; p += 64) {
            float4* row = inst.rows + (size_t)GSR_ROW_F4(C) * __float_as_uint(stream.s2[p].y);
            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            row[0] = z; row[1] = z; row[2] = z;
            if (C > 5) row[3] = z;
        }
    }
    if (hi <= 0) {  // an empty segment (a list shorter than LONG_SEGS batches): the identity
        if (PASS == 1) {
#pragma unroll
            for (int q = 0; q < PPL; q++) st_tile[(size_t)g * 256 + lane + 64 * q] = make_float2(1.0f, 0.0f);
        }
        return;
    }

    // one batch of <= 64 entries of the segment, staged into LDS; returns the candidate ballot
    auto stage = [&](int top /* first (highest) position of the batch + 1 */, int cnt) -> unsigned long long {
        __builtin_amdgcn_wave_barrier();  // the previous batch's LDS reads are done (single wave, in order)
        uint32_t mask = 0u;
        if (lane < cnt) {
            const uint32_t idx = start + (uint32_t)(top - 1 - lane);
            l0[lane] = stream.s0[idx];
            l1[lane] = stream.s1[idx];
            const float4 t2 = stream.s2[idx];
            l2[lane] = make_float4(t2.x, t2.w, t2.z, t2.y);  // (third colour, footprint mask, depth | threshold, slot)
            if (C > 3) l3[C > 3 ? lane : 0] = stream.s3[idx];
            mask = __float_as_uint(t2.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return wave_ballot(lane < cnt && (mask & 0xFFFFu) != 0u);
    };

    if (PASS == 1) {
        // ---- (m, c) of this segment, per pixel ----
        float m_[PPL], c_[PPL];
#pragma unroll
        for (int q = 0; q < PPL; q++) { m_[q] = 1.0f; c_[q] = 0.0f; }
        for (int top = hi; top > lo; top -= 64) {
            const int cnt = min(64, top - lo);
            unsigned long long wl = stage(top, cnt);
            while (wl) {
                const int j = __builtin_ctzll(wl);
                wl &= wl - 1;
                const int contributor = top - 1 - j;
                const float4 a = l0[j], b = l1[j];
                const float4 t2 = l2[j];
                const float4 c2 = make_float4(t2.x, 0.0f, t2.z, t2.y);
                float4 t3 = t2;
                if (C > 3) t3 = l3[C > 3 ? j : 0];
                const uint32_t thr_bits = __float_as_uint(C == 3 ? t2.z : t3.w);
                const float o = b.y, dx = a.x - fx;
                const SigmaX sx = sigma_x(a.z, a.w, dx);
                float f[C];
                unpack_features<C>(b, c2, t3, f);
                const uint32_t rowbits = __builtin_amdgcn_readfirstlane(__float_as_uint(t2.y));
#pragma unroll
                for (int q = 0; q < PPL; q++) {
                    if (__builtin_amdgcn_readfirstlane((int)((rowbits >> (4 * q)) & 0xFu)) == 0) continue;
                    const float dy = a.y - fy[q], dy2 = __fmul_rn(dy, dy);
                    const float sigma = sigma_of(sx, b.x, dy, dy2);
                    if (contributor < last_contributor[q] && __float_as_uint(sigma) < thr_bits) {
                        const float alpha = alpha_of(o, __expf(-sigma));
                        float cv = f[0] * vp[q][0];
#pragma unroll
                        for (int c = 1; c < C; c++) cv += f[c] * vp[q][c];
                        c_[q] = c_[q] + alpha * (cv - c_[q]);   // alpha cv + (1 - alpha) c: the A recursion from A = 0
                        m_[q] = m_[q] * (1.0f - alpha);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; q++) st_tile[(size_t)g * 256 + lane + 64 * q] = make_float2(m_[q], c_[q]);
        return;
    }

    // ---- PASS 2.  The state this segment starts from: everything BEHIND it (segments 0 .. g-1), back to front ----
    for (int k = 0; k < g; k++) {
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            const float2 mc = st_tile[(size_t)k * 256 + lane + 64 * q];
            T[q] = T[q] / mc.x;               // (a pixel nothing of the segment blends into: m = 1, c = 0)
            A[q] = mc.x * A[q] + mc.y;
        }
    }
    // ---- the gradient walk of the segment (the main kernel's body: one wave = the whole tile) ----
    for (int top = hi; top > lo; top -= 64) {
        const int cnt = min(64, top - lo);
        unsigned long long wl = stage(top, cnt);
        unsigned long long touched = 0ull;
        while (wl) {
            const int j = __builtin_ctzll(wl);
            wl &= wl - 1;
            const int contributor = top - 1 - j;
            const float4 a = l0[j], b = l1[j];
            const float4 t2 = l2[j];
            const float4 c2 = make_float4(t2.x, 0.0f, t2.z, t2.y);
            float4 t3 = t2;
            if (C > 3) t3 = l3[C > 3 ? j : 0];
            const uint32_t thr_bits = __float_as_uint(C == 3 ? t2.z : t3.w);
            const float o = b.y, dx = a.x - fx;
            const SigmaX sx = sigma_x(a.z, a.w, dx);
            float f[C];
            unpack_features<C>(b, c2, t3, f);
            float P = 0.0f, U1 = 0.0f, U2 = 0.0f, col[C];
#pragma unroll
            for (int c = 0; c < C; c++) col[c] = 0.0f;
            unsigned long long any_active = 0ull;
            const uint32_t rowbits = __builtin_amdgcn_readfirstlane(__float_as_uint(t2.y));
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                if (__builtin_amdgcn_readfirstlane((int)((rowbits >> (4 * q)) & 0xFu)) == 0) continue;
                const float dy = a.y - fy[q], dy2 = __fmul_rn(dy, dy);
                const float sigma = sigma_of(sx, b.x, dy, dy2);
                const bool c_live = contributor < last_contributor[q];
                const bool c_touch = __float_as_uint(sigma) < thr_bits;
                const bool active = c_live && c_touch;
                any_active |= wave_ballot(c_live) & wave_ballot(c_touch);
                if (active) {
                    const float G = __expf(-sigma);
                    const float alpha = alpha_of(o, G);
                    const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
                    T[q] = T[q] * rinv;
                    const float fac = alpha * T[q];
                    float cv = f[0] * vp[q][0];
#pragma unroll
                    for (int c = 1; c < C; c++) cv += f[c] * vp[q][c];
                    const float d = cv - A[q];
                    const float valpha = BG0 ? d * T[q] : d * T[q] + bgT[BG0 ? 0 : q] * rinv;
                    A[q] = A[q] + alpha * d;
                    const float t = G * valpha;
                    P += t;
                    U1 += t * dy;
                    U2 += t * dy2;
#pragma unroll
                    for (int c = 0; c < C; c++) col[c] += fac * vp[q][c];
                }
            }
            if (any_active == 0ull) continue;
            int jrow;
            asm("s_mul_i32 %0, %1, %2" : "=s"(jrow) : "s"(j), "n"(ST));
            float* const my_row = my + jrow;
            touched |= 1ull << j;
            if (C == 3) {
                const float total = gsr::wave_reduce_rowcol_rgb(P, U1, U2, col[0], col[1], col[2], dx, lane_bits, rowcol);
                if (rowcol.slot >= 0) my_row[rowcol.slot] = total;
            } else if (C == 5) {
                const float total = gsr::wave_reduce_rowcol_rgbd(P, U1, U2, col[0], col[1], col[2], col[C > 3 ? 3 : 0], dx, lane_bits, rowcol_d);
                if (rowcol_d.slot >= 0) my_row[rowcol_d.slot] = total;
            } else {
                float part[16];
#pragma unroll
                for (int k = 0; k < 16; k++) part[k] = 0.0f;
                const float dxp = dx * P;
                part[0] = col[0]; part[1] = col[1]; part[2] = col[2];
                part[3] = P; part[4] = dx * dxp; part[5] = dx * U1; part[6] = U2; part[7] = dxp; part[8] = U1;
                if (C > 3) part[9] = col[3];
                if (C > 5) { part[10] = col[5]; part[11] = col[6]; part[12] = col[7]; }
                const float total = gsr::wave_reduce_transposed<NA>(part, lane_bits);
                if (red_writer) my_row[red_slot] = total;
            }
        }
        // flush: one gradient row per staged instance (zeros when nothing touched it), the main kernel's row format
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            float r[NA];
            const bool hit = (touched >> lane) & 1ull;
#pragma unroll
            for (int k = 0; k < NA; k++) r[k] = hit ? my[lane * ST + k] : 0.0f;
            const float4 a = l0[lane], b = l1[lane];
            const float mo = -b.y, mh = -0.5f * b.y;
            float4* row = inst.rows + (size_t)GSR_ROW_F4(C) * __float_as_uint(l2[lane].w);
            row[0] = make_float4(r[0], r[1], r[2], r[3]);
            row[1] = make_float4(mh * r[4], mh * r[5], mh * r[6], C > 3 ? r[9 < NA ? 9 : 0] : 0.0f);
            row[2] = make_float4(mo * (2.0f * a.z * r[7] + a.w * r[8]), mo * (a.w * r[7] + 2.0f * b.x * r[8]),
                                 C > 5 ? r[10 < NA ? 10 : 0] : 0.0f, C > 5 ? r[11 < NA ? 11 : 0] : 0.0f);
            if (C > 5) row[3] = make_float4(r[12 < NA ? 12 : 0], 0.0f, 0.0f, 0.0f);
        }
    }
}

// exactly zero (either sign) in every channel: the kernels' BG0 specialisation; GSR_NO_BG0=1 disables it (A/B runs)
bool bg_is_zero(const Bg& b) {
    static const bool off = [] { const char* e = getenv("GSR_NO_BG0"); return e && e[0] == '1'; }();
    if (off) return false;
    for (int c = 0; c < 8; c++)
        if (b.v[c] != 0.0f) return false;
    return true;
}

Bg make_bg(const float* background, int channels) {
    Bg b;
    for (int c = 0; c < 8; c++) b.v[c] = (c < 3 && c < channels) ? background[c] : 0.0f;  // rasterizer.jl:411-414
    return b;
}

}  // namespace

void gsr_launch_composite_fwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                              const uint32_t* tile_order, GsrStream stream, const float* background, float* image,
                              uint32_t* n_contrib, float* final_T, const uint32_t* values_sorted, uint8_t* covis,
                              float* uncert, const GsrTierLists* listed) {
    // tile_order == NULL: only the tiles of *listed
    const int n_slots = tile_order ? cam.grid_x * cam.grid_y : (int)(listed->n_big + listed->n_mid8 + listed->n_mid4);
    if (n_slots <= 0) return;
    dim3 grid(32 * ((n_slots + 7) / 8)), block(64);  // 8 XCD lanes x 4 strips per launch slot
    Bg bg = make_bg(background, channels);
    const bool aux = covis || uncert;
    const GsrTierLists tiers = listed ? *listed : GsrTierLists{};
#define LAUNCH(CC, AA)                                                                                             \
    hipLaunchKernelGGL((composite_fwd_strip_kernel<CC, AA>), grid, block, 0, s, cam.width, cam.height, cam.grid_x, \
                       tile_start, tile_order, stream, bg, image, n_contrib, final_T, values_sorted, covis, uncert, tiers)
    if (channels == 3) { if (aux) LAUNCH(3, true); else LAUNCH(3, false); }
    else if (channels == 5) { if (aux) LAUNCH(5, true); else LAUNCH(5, false); }
    else { if (aux) LAUNCH(8, true); else LAUNCH(8, false); }
#undef LAUNCH
}

void gsr_launch_sort_composite_fwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                                   const uint32_t* tile_order, uint32_t* tile_count, const uint64_t* bins, uint32_t bin_cap,
                                   GsrGeom geom, GsrStream stream, const float* background, float* image,
                                   uint32_t* n_contrib, float* final_T, uint32_t* values_sorted, uint32_t* ranges,
                                   uint8_t* covis, float* uncert, const uint32_t* totals, uint32_t cap_instances,
                                   bool keep_backward_state) {
    dim3 grid(cam.grid_x * cam.grid_y), block(256);
    Bg bg = make_bg(background, channels);
    const bool aux = covis || uncert;
#define LAUNCH2(CC, AA, KK)                                                                                            \
    hipLaunchKernelGGL((sort_composite_fwd_kernel<CC, AA, KK>), grid, block, 0, s, cam.width, cam.height, cam.grid_x,  \
                       tile_start, tile_order, tile_count, bins, bin_cap, geom, stream, bg, image, n_contrib,          \
                       final_T, values_sorted, ranges, covis, uncert, totals, cap_instances)
#define LAUNCH(CC, AA) do { if (keep_backward_state) LAUNCH2(CC, AA, true); else LAUNCH2(CC, AA, false); } while (0)
    if (channels == 3) { if (aux) LAUNCH(3, true); else LAUNCH(3, false); }
    else if (channels == 5) { if (aux) LAUNCH(5, true); else LAUNCH(5, false); }
    else { if (aux) LAUNCH(8, true); else LAUNCH(8, false); }
#undef LAUNCH
#undef LAUNCH2
}

void gsr_launch_composite_bwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                              const uint32_t* tile_order, GsrStream stream, const float* background,
                              const float* vpixels, const uint32_t* n_contrib, const float* final_T, GsrInst inst,
                              uint32_t split_len, bool color_only, bool accurate) {
    dim3 grid(cam.grid_x * cam.grid_y), block(256 / GSR_BWD_PPL);
    Bg bg = make_bg(background, channels);
    GsrTierLists none{};
    none.split_len = split_len;
    const bool bg0 = bg_is_zero(bg);
#define LAUNCH_K(CC, ZZ, VV, AA)                                                                                       \
    hipLaunchKernelGGL((composite_bwd_kernel<CC, GSR_BWD_PPL, false, ZZ, VV, AA>), grid, block, 0, s, cam.width, cam.height, \
                       cam.grid_x, tile_start, tile_order, stream, bg, vpixels, n_contrib, final_T, inst, none)
    // accurate: the ACC instantiations (libm exp, IEEE division: gsr_config.grad_precision) — the general-background kernels only
#define LAUNCH2(CC, ZZ) do { if (accurate) LAUNCH_K(CC, false, CC, true); else LAUNCH_K(CC, ZZ, CC, false); } while (0)
    // the zero-background kernels where they pay (table above the kernel): C == 5 and C == 8 take BG0, C == 3 never does
    // (:rgb capped at six waves with 1.25 KB of unused dynamic LDS per workgroup takes 0.738 ms: it is the code generated under
    //  the tighter register budget that is slower, not the occupancy)
#define LAUNCH3(CC, ZZ) do { if (accurate) LAUNCH_K(CC, false, 3, true); else LAUNCH_K(CC, ZZ, 3, false); } while (0)
    if (channels == 3) LAUNCH2(3, false);
    else if (color_only) {
        // the cotangent of the loss head (channels >= 3 are zeros): the :rgb arithmetic on the mode's stream.  (BG0 as measured
        // for :rgb: the general kernel)
        if (channels == 5) LAUNCH3(5, false); else LAUNCH3(8, false);
    }
    else if (channels == 5) { if (bg0) LAUNCH2(5, true); else LAUNCH2(5, false); }
    else if (bg0) LAUNCH2(8, true);
    else LAUNCH2(8, false);
#undef LAUNCH_K
#undef LAUNCH2
#undef LAUNCH3
}

void gsr_launch_composite_bwd_listed(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                                     GsrTierLists tiers, GsrStream stream, const float* background,
                                     const float* vpixels, const uint32_t* n_contrib, const float* final_T, GsrInst inst,
                                     float* long_state) {
    const uint32_t n_listed = tiers.n_big + tiers.n_mid8 + tiers.n_mid4;
    if (n_listed == 0) return;
    Bg bg = make_bg(background, channels);
    // GSR_BWD_LONG=0: round 4's form (four waves per tile by pixel strips) for A/B runs; default: the list split along its length
    static const bool by_strips = [] { const char* e = getenv("GSR_BWD_LONG"); return e && e[0] == '0'; }();
    if (by_strips) {
        dim3 grid(n_listed), block(256);
#define LAUNCH(CC)                                                                                                 \
        hipLaunchKernelGGL((composite_bwd_kernel<CC, 1, true, false>), grid, block, 0, s, cam.width, cam.height, cam.grid_x,  \
                           tile_start, (const uint32_t*)nullptr, stream, bg, vpixels, n_contrib, final_T, inst, tiers)
        if (channels == 3) LAUNCH(3);
        else if (channels == 5) LAUNCH(5);
        else LAUNCH(8);
#undef LAUNCH
        return;
    }
    dim3 grid(n_listed * LONG_SEGS), block(64);
    const bool bg0 = bg_is_zero(bg);
    float2* st = reinterpret_cast<float2*>(long_state);
#define LAUNCH(CC, ZZ)                                                                                                    \
    do {                                                                                                                  \
        hipLaunchKernelGGL((composite_bwd_long_kernel<CC, ZZ, 1>), grid, block, 0, s, cam.width, cam.height, cam.grid_x,  \
                           tile_start, stream, bg, vpixels, n_contrib, final_T, inst, tiers, st);                         \
        hipLaunchKernelGGL((composite_bwd_long_kernel<CC, ZZ, 2>), grid, block, 0, s, cam.width, cam.height, cam.grid_x,  \
                           tile_start, stream, bg, vpixels, n_contrib, final_T, inst, tiers, st);                         \
    } while (0)
    if (channels == 3) { if (bg0) LAUNCH(3, true); else LAUNCH(3, false); }
    else if (channels == 5) { if (bg0) LAUNCH(5, true); else LAUNCH(5, false); }
    else if (bg0) LAUNCH(8, true);
    else LAUNCH(8, false);
#undef LAUNCH
}
