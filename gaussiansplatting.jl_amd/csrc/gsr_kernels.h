// Internal interface between the C ABI (gsr_api.cpp) and the gfx950 kernels.
// Not installed; include/gsr.h is the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_TILE 16         // GaussianSplatting.jl:55 BLOCK
#define GSR_BATCH 256       // GaussianSplatting.jl:56 BLOCK_SIZE
#define GSR_SORT_LDS_CAP 8192  // keys per tile sorted in LDS (64 KB); longer lists use the global path

// Camera + config constants, passed BY VALUE as a kernel argument (lands in SGPRs).
struct GsrCam {
    float R[9];  // row-major R[r*3+c] (converted from the ABI's column-major)
    float t[3];
    float focal[2];
    float principal[2];  // normalised
    float center[3];
    int width, height;
    int grid_x, grid_y;
    float near_plane, far_plane;
    int radius_clip;
    float blur_eps;
    int exact_cull;      // GSR_FLAG_EXACT_TILE_CULL
    const float* R_dev;  // optional device overrides (column-major (3,3), (3))
    const float* t_dev;
};

// Per-Gaussian geometry written by preprocess: ONE 64-byte record per Gaussian (a gather by
// id — the tile sort does 5.7 M of them per view — touches a single cache line).
//   q0 = mean2d.x, mean2d.y, conic.a, conic.b
//   q1 = conic.c, opacity, r, g
//   q2 = b, clamped bits (u32), depth, lpre (u32: exclusive prefix of tile-rect areas inside the
//        Gaussian's 256-wide block; + bpre[i >> 8] = Gaussian-major slot of its first instance)
//   q3 = rect xmin | ymin << 16, rect xmax | ymax << 16 (u32), blend-test threshold bits X (u32, tile_mask.h),
//        emitted-tile mask of the rect (u32, row-major bit per tile; meaningful for rects of <= 32 tiles: pergauss.hip DENSE_RECT)
struct GsrGeoRec { float4 q0, q1, q2, q3; };
// Tile rects of at most this many tiles get one gradient-row slot per EMITTED tile (ranked through the record's mask);
// larger ones one per tile of the rect (pergauss.hip: preprocess, pergauss_bwd; tile_sort_device.h: the emit's slot).
constexpr uint32_t GSR_DENSE_RECT = 32;
struct GsrGeom {
    GsrGeoRec* rec;
    float4* normal;   // camera-space normal (C == 8) or nullptr
    int32_t* radii;
    uint32_t* bsum;   // per 256-Gaussian block: sum of tile-rect areas (scanned into bpre by tile_scan)
    uint32_t* bpre;
};

// Sorted per-instance splat stream written by tile_sort (planes of float4, coalesced).
struct GsrStream {
    float4* s0;  // mean2d.x, mean2d.y, conic.a / 2, conic.b
    float4* s1;  // conic.c / 2, opacity, r, g
    float4* s2;  // b, slot (uint bits: Gaussian-major instance slot), depth | :rgb: blend-test threshold bits (tile_sort_device.h), row mask (uint bits: tile rows touched)
    float4* s3;  // :rgbd / :rgbdn: normal xyz (C == 8, else 0) + the blend-test threshold bits in w (tile_sort_device.h); :rgb: nullptr
};

// Backward: per-INSTANCE gradient rows (plain stores, no atomics), indexed by the
// Gaussian-major instance slot, so the rows of one Gaussian are contiguous and the
// per-Gaussian kernel sums them in a fixed order (deterministic gradients).
//   row = 4 x float4: {v r, v g, v b, v opacity}, {v conic a,b,c, v depth}, {v mean2d x,y, v normal x,y}, {v normal z,-,-,-}
// segments a long tile list is cut into by the list-parallel backward (composite_bwd_long_kernel)
#ifndef GSR_BWD_LONG_SEGS
#define GSR_BWD_LONG_SEGS 32
#endif
#ifndef GSR_ROW_F4
#define GSR_ROW_F4(C) ((C) > 3 ? 4 : 3)  // float4s per gradient row: 48 bytes in :rgb mode (9 floats used), 64 otherwise
#endif
struct GsrInst {
    float4* rows;  // D_slots x 4 float4, zero-filled per backward
};

// ---- pergauss.hip (compiled with -ffp-contract=off: bit-reproducible fp32) ----
// aggregating: the binning form gsr_policy_begin_view chose for this view (false = the direct form);
// returns the form that ran (gsr_stats.preprocess_form: 0 direct, 1 / 2 aggregating with 2 x 32 / 2 x 16-bit LDS words, 3 banded)
int gsr_launch_preprocess(hipStream_t s, int n, int K, int degree, int channels, const float* means,
                          const float* scales, const float* rots, const float* opac, const float* shs, GsrCam cam,
                          GsrGeom geom, uint32_t* tile_count, uint32_t* n_visible /* per 256-block */,
                          uint64_t* bins /* (T+1) x bin_cap keys */, uint32_t bin_cap, int n_tiles, bool aggregating);
struct GsrBg8 { float v[8]; };
void gsr_launch_fill_background(hipStream_t s, size_t n_pixels, int channels, const float* background /* host, 3 floats */,
                                float* image, float* final_T, uint32_t* n_contrib);
// compact binning mode: scatter the keys to tile_start[t] + arrival rank (tile_fill zeroed by the caller).
// only_above > 0: only the tiles whose list is LONGER than that (the lists that overflowed fixed-capacity bins) — the others'
// segments of `keys` and their fill cursors are not touched
void gsr_launch_emit_compact(hipStream_t s, int n, GsrCam cam, GsrGeom geom, const uint32_t* tile_start, uint32_t* tile_fill,
                             uint64_t* keys, uint32_t max_list /* longest tile list of the view */, uint32_t only_above);
void gsr_launch_pergauss_bwd(hipStream_t s, int n, int K, int degree, int channels, const float* means,
                             const float* scales, const float* rots, const float* shs, GsrCam cam, GsrGeom geom,
                             GsrInst inst, float2* vmean2d, float* vmeans, float* vshs, float* vopac,
                             float* vscales, float* vrots, float* vR, float* vt,
                             float* vcolors /* (3,N) or NULL: factored SH gradient instead of vshs */,
                             bool fp32_chain /* ∇scales / ∇rotations by the reference's fp32 trees (GSR_GRAD_FP32_REFERENCE) */);
// backward epilogue = trainer tail (single-GPU step): no gradient arrays, the parameters / Adam states in S are
// updated in place and the activated copies of the next forward written (adam_math.h)
namespace gsr { struct TailState; }
void gsr_launch_pergauss_bwd_tail(hipStream_t s, int n, int K, int degree, int channels, GsrCam cam, GsrGeom geom,
                                  GsrInst inst, float2* vmean2d, const gsr::TailState& S, bool fp32_chain);
void gsr_launch_sh_grad_views(hipStream_t s, int n, int K, int degree, int n_views, const float* centers,
                              const float* means, const float* vc_all, float* vshs);
// the same rebuild with the trainer tail applied in place of the ∇shs store (multi-GPU trainer step; S.points = the means)
void gsr_launch_sh_views_tail(hipStream_t s, int n, int K, int degree, int n_views, const float* centers,
                              const float* vc_all, const float* vmeans, const float* vopac_act, const float* vscales_act,
                              const float* vrot, const gsr::TailState& S);

void gsr_launch_update_stats(hipStream_t s, int n, const int32_t* radii, const float2* vmean2d, int width, int height,
                             int32_t* max_radii, float* accum, float* denom);

// ---- binning.hip ----
// exclusive scan of tile_count -> tile_start[T+1]; and of the per-block
// rect-area sums bsum[nb] -> bpre[nb];
// totals[0] = D, totals[1] = max count, totals[2] = #tiles over GSR_SORT_LDS_CAP, totals[3] = slab counter (0)
void gsr_launch_tile_scan(hipStream_t s, int n_tiles, const uint32_t* tile_count, uint32_t* tile_start,
                          uint32_t* totals /* 8 words, [7] = ticket (zero between launches) */, int n_blocks,
                          const uint32_t* bsum, uint32_t* bpre, const uint32_t* bvis,
                          uint32_t* tier_lists /* [3 * n_tiles], see gsr_launch_tile_sort */,
                          uint32_t* host_mirror /* pinned host, 8 words: totals[0..6] + seq, or NULL */, uint32_t seq,
                          uint32_t* order /* [n_tiles] tile ids by descending list length: launch order of the compositing
                                             workgroups, computed by a third workgroup of the same launch */);
// bin_cap > 0: keys of tile t at bins + t * bin_cap; bin_cap == 0: compact layout, keys of tile t at bins + tile_start[t].
// tier_lists (written by tile_scan): [0, T) tiles with lists > 8192, [T, 2T) lists in (4096, 8192], [2T, 3T) in (1024, 4096]
// passes: GSR_SORT_PASS_MAIN = the T-workgroup pass over lists of up to 1024 keys (also writes `ranges` and re-zeroes the
// counters), GSR_SORT_PASS_TIERS = the launches over the tier lists (their sizes are host-side numbers).  totals != NULL:
// the main pass is launched before the host knows the counts and leaves everything untouched when the view needs more
// than cap_instances instances or bin_cap keys in a bin (the host then launches it again with totals == NULL).
#define GSR_SORT_PASS_MAIN 1
#define GSR_SORT_PASS_TIERS 2
void gsr_launch_tile_sort(hipStream_t s, int passes, int n_tiles, int grid_x, int channels, const uint32_t* tile_start,
                          uint32_t* tile_count /* re-zeroed for the next view */, const uint64_t* bins, uint32_t bin_cap,
                          const uint64_t* overflow_keys /* compact-layout keys of the lists longer than bin_cap, or NULL */,
                          uint32_t n_mid4, uint32_t n_mid8, uint32_t n_big, const uint32_t* tier_lists,
                          uint64_t* big_scratch /* 2 slabs of slab_stride keys per tile over 8192 */, size_t slab_stride, GsrGeom geom,
                          GsrStream stream, uint32_t* values_sorted, uint32_t* ranges, const uint32_t* totals,
                          uint32_t cap_instances, uint32_t first4 = 0, uint32_t first8 = 0 /* leading tiles of the mid tier
                          lists already sorted by gsr_launch_tile_sort_mid */);
// the sorts of the (1024, 4096] / (4096, 8192] tiers launched BEFORE the host has the counts: grids are guesses, every workgroup
// checks `totals` (device) — instances <= cap_instances, longest list <= bin_cap, its slot < the tier's count — else leaves
void gsr_launch_tile_sort_mid(hipStream_t s, int n_tiles, int grid_x, int channels, const uint32_t* tile_start,
                              const uint64_t* bins, uint32_t bin_cap, uint32_t grid4, uint32_t grid8, const uint32_t* tier_lists,
                              GsrGeom geom, GsrStream stream, uint32_t* values_sorted, const uint32_t* totals,
                              uint32_t cap_instances);

// ---- composite.hip ----
// Tiles whose list is longer than split_len (a tier boundary of the scan: 1024, 4096, 8192, or 0xFFFFFFFF for none)
// are left out by gsr_launch_composite_bwd and walked by four waves each (one 16x4 pixel strip per wave) in the
// listed launch, on a second stream.
struct GsrTierLists {  // the scan's tier lists: [0, T) lists > 8192, [T, 2T) (4096, 8192], [2T, 3T) (1024, 4096]
    const uint32_t* lists;
    uint32_t n_tiles, n_big, n_mid8, n_mid4;  // a tier that is not split has count 0 here
    uint32_t split_len;  // (forward launch over the lists: non-zero = its waves run at raised issue priority, beside another launch)
};
void gsr_launch_composite_fwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                              const uint32_t* tile_order /* NULL: only the tiles of *listed */, GsrStream stream,
                              const float* background, float* image, uint32_t* n_contrib, float* final_T,
                              const uint32_t* values_sorted, uint8_t* covis, float* uncert,
                              const GsrTierLists* listed /* or NULL */);
// sort + forward of every tile in one launch (fixed-capacity bins, no list beyond 1024 instances; checked on the
// device against the scan's totals — a view that does not qualify leaves everything untouched)
void gsr_launch_sort_composite_fwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                                   const uint32_t* tile_order, uint32_t* tile_count, const uint64_t* bins, uint32_t bin_cap,
                                   GsrGeom geom, GsrStream stream, const float* background, float* image,
                                   uint32_t* n_contrib, float* final_T, uint32_t* values_sorted, uint32_t* ranges,
                                   uint8_t* covis, float* uncert, const uint32_t* totals, uint32_t cap_instances,
                                   bool keep_backward_state /* false (GSR_FORWARD_ONLY): the sorted stream and ids are not stored */);
void gsr_launch_composite_bwd(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                              const uint32_t* tile_order, GsrStream stream,
                              const float* background, const float* vpixels, const uint32_t* n_contrib,
                              const float* final_T, GsrInst inst,
                              uint32_t split_len /* tiles with a longer list are left to the listed launch */,
                              bool color_only /* channels >= 3 of vpixels are zeros (the loss head's cotangent) */,
                              bool accurate /* libm exp + IEEE division per pixel (gsr_config.grad_precision); the caller then
                                               leaves EVERY tile to this launch (split_len = 0xFFFFFFFF) */);
void gsr_launch_composite_bwd_listed(hipStream_t s, int channels, GsrCam cam, const uint32_t* tile_start,
                                     GsrTierLists tiers, GsrStream stream, const float* background,
                                     const float* vpixels, const uint32_t* n_contrib, const float* final_T, GsrInst inst,
                                     float* long_state /* GSR_BWD_LONG_SEGS x 512 floats per listed tile */);

// ---- trainer.hip ----
#define GSR_ADAM_MAX_GROUPS 8
void gsr_launch_prologue_fwd(hipStream_t s, int n, int k_rest, int scale_dims, const float* sh_color,
                             const float* sh_remainder, const float* opacities, const float* scales, float* shs,
                             float* opacities_act, float* scales_act);
void gsr_launch_prologue_bwd(hipStream_t s, int n, int k_rest, int scale_dims, const float* opacities_act,
                             const float* scales_act, const float* vshs, const float* vopacities_act,
                             const float* vscales_act, float* v_sh_color, float* v_sh_remainder, float* v_opacities,
                             float* v_scales);
void gsr_launch_adam(hipStream_t s, int n_groups, float* const* theta, const float* const* grad, float* const* mu,
                     float* const* nu, const long long* count, const float* lr_t, float beta1, float beta2, float eps);

void gsr_launch_trainer_tail(hipStream_t s, int n, int k_rest, int scale_dims, const float* const* grads,
                             float* const* theta, float* const* mu, float* const* nu, const float* lr_t, float beta1,
                             float beta2, float eps, float* shs, float* opac_act, float* scales_act);
size_t gsr_findall_scratch_bytes(long long n);
void gsr_launch_findall(hipStream_t s, long long n, const uint8_t* mask, uint32_t* indices, uint32_t* count_dev,
                        uint32_t* scratch);
void gsr_launch_gather_rows(hipStream_t s, int n_groups, const void* const* src, void* const* dst, const int* row_words,
                            const uint32_t* idx, long long count);
void gsr_launch_triad(hipStream_t s, size_t n4, float* a, const float* b, const float* c, float q);

// ---- densify.hip (compiled with -ffp-contract=off) ----
#define GSR_COMPOSE_MAX_GROUPS 24
void gsr_launch_grad_mean(hipStream_t s, long long n, const float* accum, const float* denom, float* out);
void gsr_launch_densify_mask(hipStream_t s, int kind, long long n, long long n_grad, const float* grad, const float* scales,
                             int scale_dims, const float* opacities, const int32_t* max_radii, float thr, float gamma,
                             float min_opacity, int max_screen_size, uint8_t* mask);
void gsr_launch_compose_rows(hipStream_t s, int n_groups, const void* const* src, void* const* dst, const int* row_words,
                             const int* new_zero, const uint32_t* keep_idx, long long n_keep, const uint32_t* sel_idx,
                             long long n_sel, int reps);
void gsr_launch_split_transform(hipStream_t s, long long n_new, int scale_dims, float* points, const float* rots,
                                float* scales, uint32_t seed);
void gsr_launch_reset_opacity(hipStream_t s, long long n, float* opacities);
void gsr_launch_morton_codes(hipStream_t s, long long n, const float* points, const float lo[3], const float inv_extent[3],
                             unsigned long long* codes);
void gsr_launch_nonfinite_scan(hipStream_t s, int n_groups, const float* const* src, const int* row_words, long long n_rows,
                               uint32_t* counts, uint32_t* first_bad);
void gsr_launch_ply_rows(hipStream_t s, bool pack, long long n, int kr, float* points, float* dc, float* rest, float* opac,
                         float* scales, float* rots, float* rows);

// ---- ssim.hip (compiled twice: *_exact = -ffp-contract=off + IEEE divisions, bit-exact vs the oracle; *_fast = contracted
// multiply-adds + hardware reciprocals, the default path; gsr_ssim_precision selects) ----
#define GSR_SSIM_DECL(SUF)                                                                                              \
    void gsr_launch_ssim_fwd_##SUF(hipStream_t s, int W, int H, int CH, int B, const float* img, const float* ref, float C1, \
                                   float C2, int train, float* ssim_map, float* d0, float* d1, float* d2);             \
    void gsr_launch_ssim_bwd_##SUF(hipStream_t s, int W, int H, int CH, int B, const float* img, const float* ref,      \
                                   const float* dL_dmap, const float* d0, const float* d1, const float* d2, float* dL_dimg); \
    /* fused loss head: image (C,W,H) vs target (W,H,3); partial: [3T][2] per-workgroup sum|x-y|, sum ssim */          \
    void gsr_launch_loss_fwd_##SUF(hipStream_t s, int W, int H, int C, const float* image, const float* target, float C1, \
                                   float C2, float* d0, float* d1, float* d2, float* partial);                         \
    void gsr_launch_loss_bwd_##SUF(hipStream_t s, int W, int H, int C, const float* image, const float* target, float lambda, \
                                   const float* d0, const float* d1, const float* d2, const float* partial, float* loss_out, \
                                   float* vpixels);
GSR_SSIM_DECL(exact)
GSR_SSIM_DECL(fast)
#undef GSR_SSIM_DECL
