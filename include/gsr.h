/*
 * gsr.h — C ABI of libgsr_hip.so, the MI355X (gfx950) drop-in for the differentiable
 * rasterizer of GaussianSplatting.jl.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference repository, `src/rasterization/` unless stated).  Conventions:
 *   - every pointer marked "device" is a HIP device address owned by the caller; the
 *     library never frees or retains it beyond the call;
 *   - array layouts are the reference's (Julia column-major): means (3,N), scales (3,N),
 *     rotations (4,N) as (w,x,y,z), opacities (1,N), shs (3,K,N), image (C,W,H) channel
 *     fastest, n_contrib / final_T (W,H) x fastest, tile_ranges (2,T);
 *   - all work is enqueued on the `stream` argument (a hipStream_t passed as void*);
 *     the only host synchronisation in steady state is the instance-count read-back inside
 *     gsr_forward (the reference has the same one: rasterizer.jl:337).  Scratch is grow-only
 *     (rasterizer.jl:275-278,340-343): a call that has to GROW a scratch buffer (first view, more
 *     Gaussians, a much denser view) frees and reallocates it with hipFree / hipMalloc, which
 *     synchronise the device — exactly where the reference reallocates;
 *   - every function returns 0 on success or a negative GSR_E_* code; no C++ exception
 *     crosses this boundary; gsr_last_error_string() describes the last failure on the
 *     calling thread;
 *   - a handle supports one outstanding forward -> backward pair (as the reference's
 *     GaussianRasterizer object does, rasterizer.jl:13-15); distinct handles are
 *     independent.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_API __attribute__((visibility("default")))

enum {
    GSR_OK = 0,
    GSR_E_INVALID_ARG = -1, /* reference: @assert / error() in rasterizer.jl:51,66-68,281 */
    GSR_E_OOM = -2,
    GSR_E_HIP = -3,         /* a HIP runtime call failed */
    GSR_E_STATE = -4        /* backward without a matching forward */
};

/* Number of blended feature channels (rasterizer.jl:47-51 `n_color_features`). */
enum { GSR_MODE_RGB = 3, GSR_MODE_RGBD = 5, GSR_MODE_RGBDN = 8 };

/* Replaces the keyword arguments of `GaussianRasterizer(kab; width, height, mode,
 * near_plane, far_plane)` (rasterizer.jl:60-65) and the constants hard-coded in
 * `rasterize` (radius_clip = 3, blur_eps = 0.3: rasterizer.jl:294-295,505).  Unlike the
 * reference (rasterizer.jl:66) width/height need not be multiples of 16: partial tiles
 * are masked. */
typedef struct gsr_config {
    int32_t width, height;
    int32_t mode;        /* GSR_MODE_* */
    float near_plane;    /* 0.2  */
    float far_plane;     /* 1000 */
    int32_t radius_clip; /* 3 px */
    float blur_eps;      /* 0.3  */
    uint32_t flags;      /* GSR_FLAG_* */
    uint64_t bins_budget_bytes; /* 0 = default.  Cap on the fixed-capacity per-tile key bins of the fast binning mode
                          * ((tiles+1) x capacity x 8 B); default max(512 MiB, 160 B x instance count of the last view).
                          * Where bins for the longest list would exceed it — a few very deep tiles — the bins are sized for the
                          * other tiles (4 x the mean list, at least 1024 keys) and only the lists beyond that capacity are
                          * scattered a second time (+ 8 B per instance); a budget below bins of 2 x the mean list (or of 64
                          * keys) selects the compact mode (count -> scan -> scatter, 8 B per instance, no bins).  Same lists,
                          * same results in every mode. */
    /* The behaviour switches are PER HANDLE (ABI 5), as the reference's knobs are constructor keywords (rasterizer.jl:60-65).
     * ABI 6: 0 = GSR_DEFAULT everywhere, so that a zero-initialised (memset) gsr_config means "the defaults" — under ABI 5's
     * encoding (-1 default, 0 / 1 explicit) it silently pinned fast SSIM and direct binning.  GSR_DEFAULT follows the
     * process-wide default (gsr_ssim_precision / gsr_preprocess_form below, themselves started from GSR_SSIM_EXACT /
     * GSR_PREPROCESS_AGG), read at every call; an explicit value pins the handle whatever another thread sets process-wide —
     * e.g. a GUI render worker next to a trainer (gui/worker.jl:47-58).  Other values: GSR_E_INVALID_ARG. */
    int32_t ssim_precision;  /* arithmetic of gsr_loss_l1_ssim on this handle: GSR_DEFAULT, GSR_SSIM_FAST, GSR_SSIM_EXACT */
    int32_t preprocess_form; /* binning form of gsr_forward on this handle: GSR_DEFAULT (by size), GSR_PREPROCESS_DIRECT,
                              * GSR_PREPROCESS_AGGREGATING */
    /* New in ABI 6 */
    int32_t form_tuner;      /* on 4K-class grids, where neither binning form wins everywhere, the handle times four views'
                              * first kernel (two per form) and keeps the faster form (gsr_policy.h; results are bit-identical
                              * either way): GSR_DEFAULT (on; GSR_FORM_TUNER=0 in the environment turns the default off),
                              * GSR_TUNER_OFF (the previous view's skew decides, as before the tuner existed), GSR_TUNER_ON.
                              * An explicit preprocess_form pins the form and leaves the tuner nothing to decide. */
    int32_t grad_precision;  /* arithmetic of gsr_backward where fp32 shortcuts show on needle-shaped splats (2-D axis ratio beyond
                              * ~10 : 1), whose sums cancel by up to the square of that ratio:
                              *   GSR_DEFAULT: (i) ∇render!'s per-pixel exp / reciprocal are the hardware's fast instructions
                              *     (v_exp_f32 on sigma x log2 e, v_rcp_f32); (ii) the chain of ∇scales / ∇rotations (∇inverse ->
                              *     ∇perspective -> ∇covar_world_to_cam -> ∇quat_scale_to_cov -> ∇unnorm_quat2rot, projection.jl:132-257,
                              *     render.jl:302-366) is evaluated in float64 from the raw inputs.  ∇means of a 90 : 1 needle of
                              *     radius 100 px is then ~4e-4 from float64 (the reference's fp32 atomics: 0.2 .. 3e-4), everything
                              *     well conditioned ~1e-6;
                              *   GSR_GRAD_ACCURATE: (i) libm-accurate exp and IEEE division — what the reference's source means by
                              *     `exp` and `/` (render.jl:236-259) — with (ii) unchanged: the closest to float64 the library gets
                              *     (that needle: 2e-5); ∇render! +12 %, and long tile lists are not split along their length;
                              *   GSR_GRAD_FP32_REFERENCE: (i) as GSR_GRAD_ACCURATE, (ii) the reference's own fp32 expression trees,
                              *     operation for operation: reference-parity runs (equal to the fp32 CPU oracle to ~1e-6 on
                              *     well-conditioned Gaussians; on needles ∇rotations carry the fp32 chain's 1e-4 .. 1e-3). */
} gsr_config;
#define GSR_DEFAULT 0
enum { GSR_SSIM_FAST = 1, GSR_SSIM_EXACT = 2 };
enum { GSR_PREPROCESS_DIRECT = 1, GSR_PREPROCESS_AGGREGATING = 2 };
enum { GSR_TUNER_OFF = 1, GSR_TUNER_ON = 2 };
enum { GSR_GRAD_FP32_REFERENCE = 1, GSR_GRAD_ACCURATE = 2 };

/* Tile lists.  DEFAULT (flags = 0): exact footprint culling at binning — a (Gaussian, tile)
 * instance none of whose pixels can reach alpha >= 1/255 is not emitted at all.  The reference
 * emits it and then skips it pixel by pixel (render.jl:95), so the image, final_T, every
 * gradient, radii and ∇means_2d are unchanged (image / final_T bit-identical, tested at 1 M
 * Gaussians); only the INTERNAL lists shrink: gsr_stats.n_rendered, GSR_BUF_TILE_RANGES and
 * GSR_BUF_VALUES_SORTED describe the culled lists (each an order-preserving subsequence of the
 * reference's), and GSR_BUF_N_CONTRIB counts positions in them.  No caller of the reference reads
 * those (bstate / istate are private to rasterize / ∇rasterize).
 * GSR_FLAG_REFERENCE_TILE_LISTS: keep exactly the reference's lists (duplicate_with_keys!,
 * utils.jl:85-120) — for list-level parity checks; ~7 % slower at 1 M Gaussians @1080p. */
#define GSR_FLAG_REFERENCE_TILE_LISTS 2u
/* Bit 1u is RETIRED: ABI 1 used it for GSR_FLAG_EXACT_TILE_CULL (the opposite meaning, when exact culling was opt-in).
 * gsr_create rejects it with GSR_E_INVALID_ARG so that a caller built against the old header fails loudly instead of
 * silently getting the other list mode. */
#define GSR_FLAG_RETIRED_BIT0 1u

/* ABI version of this header: bumped whenever a struct of this file changes size or a flag / enum value changes meaning.
 *   1: round-1 layout (flag bit 1u = exact tile cull, smaller gsr_config / gsr_aux / gsr_stats / gsr_grads)
 *   2: round-2 layout (flag bit 1u = reference tile lists) — never given a number at the time
 *   3: round-3 layout (GSR_FLAG_REFERENCE_TILE_LISTS = 2u, bit 1u rejected, gsr_check_abi)
 *   4: gsr_aux grew by `flags` (GSR_FORWARD_ONLY) + `reserved`; gsr_sh_grad_from_views_tail.
 *   5: gsr_config grew by `ssim_precision` + `preprocess_form` (per-handle switches); gsr_stats.reserved became
 *      `preprocess_form` (the form that ran); gsr_get_ssim_precision / gsr_get_preprocess_form; late: gsr_grads and
 *      gsr_tail_state grew by `flags` + `reserved` (GSR_GRADS_COLOR_COTANGENT).
 *   6: this layout: gsr_config's switches re-encoded with 0 = default (a zero-initialised config is the default config) and
 *      grown by `form_tuner` + `grad_precision`; gsr_stats grew by the handle's view-history counters; gsr_check_abi also
 *      takes sizeof(gsr_tail_state); GSR_GRADS_COLOR_COTANGENT is checked against the cotangent gsr_loss_l1_ssim wrote;
 *      the policies are GPU-free exports of their own (gsr_policy.h). */
#define GSR_ABI_VERSION 6

/* Positional arguments of `rasterize(means_3d, shs, opacities, scales, rotations, ...)`
 * (rasterizer.jl:255-267).  opacities / scales are the ACTIVATED values, as the
 * reference's functor prologue produces them (rasterizer.jl:228-248). */
typedef struct gsr_inputs {
    int32_t n;              /* number of Gaussians */
    int32_t n_coeffs;       /* K: SH coefficients stored per Gaussian (1, 4, 9 or 16) */
    int32_t sh_degree;      /* active degree, (sh_degree+1)^2 <= K */
    const float* means;     /* device (3,N) */
    const float* shs;       /* device (3,K,N) */
    const float* opacities; /* device (1,N) */
    const float* scales;    /* device (3,N) */
    const float* rotations; /* device (4,N), 16-byte aligned (simd.jl:1-11) */
    float background[3];    /* keyword `background` */
} gsr_inputs;

/* The fields `rasterize` reads from `camera::Camera` (rasterizer.jl:285-291,310,321;
 * camera.jl:2-16).  R_dev/t_dev are the optional positional `R_w2c`, `t_w2c` device
 * arrays of the pose-optimisation variant (rasterizer.jl:261, projection.jl:71-75);
 * when non-NULL they override R/t. */
typedef struct gsr_camera {
    float R[9];             /* world->camera rotation, column-major */
    float t[3];
    float focal[2];         /* pixels */
    float principal[2];     /* normalised to [0,1] (projection.jl:268) */
    float camera_center[3];
    const float* R_dev;     /* device (3,3) column-major or NULL */
    const float* t_dev;     /* device (3) or NULL */
} gsr_camera;

/* Optional side outputs of `render!` (render.jl:7-8,109-112,128); keywords
 * `covisibilities`, `uncertainties` of rasterize (rasterizer.jl:265-266). */
typedef struct gsr_aux {
    uint8_t* covisibilities; /* device (N) Bool, set to 1 where T > 0.5, never cleared; or NULL */
    float* uncertainties;    /* device (W,H); or NULL */
    /* `rast.gstate.radii` (states.jl:12, Int32 (N)) — read by the densification strategy right after
     * the step (strategy.jl:85-86).  When non-NULL the forward writes the radii THERE instead of into
     * handle-owned memory, so the caller's own GeometryState stays truthful; the array must stay valid
     * until the matching gsr_backward (which re-reads it).  NULL: handle-owned (GSR_BUF_RADII). */
    int32_t* radii;
    /* GSR_FORWARD_*.  New in ABI 4.  The reference's functor has a branch for rendering outside AD (rasterizer.jl:214-248,
     * `within_gradient`), used by `validate` (training.jl:501-504), the GUI (gui/worker.jl:654-657) and
     * scripts/render-views.jl: the same `rasterize`, whose backward state nobody will read. */
    uint32_t flags;
    uint32_t reserved; /* 0 */
} gsr_aux;
/* gsr_aux.flags — GSR_FORWARD_ONLY: this forward will not be differentiated.  The image, final_T, n_contrib, radii, tile ranges
 * and the geometry records are produced as always (image / final_T bit-identical to a training forward), but the sorted
 * splat stream and the sorted ids — 52-68 bytes per tile instance that only gsr_backward reads, a third of the fused
 * forward's HBM traffic — are not written and no gradient-row storage is reserved: GSR_BUF_VALUES_SORTED and
 * GSR_BUF_INSTANCE_AUX report 0 bytes, and gsr_backward / gsr_backward_trainer_tail / gsr_update_stats after such a forward
 * fail with GSR_E_STATE.  (Tiles with more than 1024 instances, and views binned in compact mode, still go through the
 * stream: they are the rare path.) */
#define GSR_FORWARD_ONLY 1u

typedef struct gsr_stats {
    int64_t n_rendered;         /* D: tile instances (rasterizer.jl:337) */
    int32_t n_visible;          /* V: count(radii > 0) */
    int32_t max_tile_instances; /* longest per-tile list */
    uint64_t generation;        /* ordinal of this forward on the handle (1, 2, ...): pass it to
                                 * gsr_backward (gsr_grads.forward_generation) to have the pairing checked */
    int64_t bins_bytes;         /* bytes of unsorted-key storage this view used (bins: (T+1) x capacity x 8; compact: 8 D;
                                 * bins + overflow tiles: the sum) */
    int32_t compact_binning;    /* 0: fixed-capacity bins; 1: the whole view was binned count -> scan -> scatter (no budget for
                                 * bins, or bins of < 1024 keys overflowed); 2: bins, and the lists beyond their capacity
                                 * scattered a second time (the rest of the view stayed on the fast path) */
    int32_t preprocess_form;    /* the binning form this view's first kernel ran in: 0 direct, 1 aggregating (2 x 32-bit LDS
                                 * words), 2 aggregating (2 x 16-bit words), 3 aggregating in horizontal bands of the tile grid */
    /* New in ABI 6 — the handle's VIEW HISTORY (gsr_policy.h: gsr_policy_state), cumulative since gsr_create.  gsr_forward keeps
     * state from one view to the next (bins capacity, compact mode, the form tuner, the held fused launch); a training run in
     * which N, D and the list skew drift can read here what that state did: in steady state none of the first four moves. */
    uint32_t bins_regrowths;    /* views after which the bins' capacity grew (a hipFree + hipMalloc before the next view) */
    uint32_t compact_fallbacks; /* views whose small bins overflowed: filled in vain, then binned again compactly */
    uint32_t tuner_rearms;      /* times the form tuner started over after a decision (scene size moved by 25 %, or 4096 views) */
    uint32_t scratch_regrowths; /* reallocations of any grow-only scratch buffer of the handle (each synchronises the device) */
    uint32_t fused_relaunches;  /* early fused launches that found the per-instance buffers too small: a no-op + the redo */
    uint32_t held_views;        /* views whose fused launch was held so that the tier walk could run beside it */
    uint32_t bin_capacity;      /* keys per tile of THIS view's bins (0: the view was binned compactly) */
    int32_t tuner_form;         /* the form tuner's decision on this handle: -1 none (yet / not its territory), 0 direct, 1 aggregating */
    float tuner_ms[2];          /* the measurement behind it: faster of two timed views of the direct / the aggregating form */
    uint32_t tier_tiles[3];     /* tiles of THIS view whose list has (1024, 4096], (4096, 8192], > 8192 instances: the lists the
                                 * fused sort + forward launch leaves to the tier sorts and the tier walk */
    uint32_t reserved;
} gsr_stats;

/* Cotangents returned by `∇rasterize` (rasterizer.jl:549): caller-provided device
 * buffers, fully overwritten (culled Gaussians and SH bands above sh_degree get exact
 * zeros, projection.jl:172-176).  Gradients are w.r.t. the ACTIVATED opacity/scale.
 * vR/vt (pose optimisation, projection.jl:243-256) may be NULL. */
typedef struct gsr_grads {
    float* vmeans;     /* (3,N) */
    float* vshs;       /* (3,K,N) */
    float* vopacities; /* (1,N) */
    float* vscales;    /* (3,N) */
    float* vrotations; /* (4,N) */
    float* vR;         /* (3,3) column-major or NULL */
    float* vt;         /* (3) or NULL */
    float* vcolors;    /* (3,N) or NULL.  New: when given, vshs is NOT written (may be NULL)
                        * and the view's SH-coefficient gradient is returned in its factored form — the colour
                        * cotangent after the clamp mask, vc — from which gsr_sh_grad_from_views rebuilds
                        * Σ_views basis(dir_view) x vc_view.  3 floats per Gaussian cross the links, not 3K. */
    float* vmeans2d;   /* (2,N) or NULL: `rast.gstate.∇means_2d` (states.jl:8; rasterizer.jl:440,474) — when given
                        * the screen-space mean gradient the densification statistics read (strategy.jl:85-86) is
                        * written THERE (fully overwritten; zeros for culled Gaussians) instead of into
                        * handle-owned memory (GSR_BUF_GRAD_MEANS2D then returns this pointer). */
    uint64_t forward_generation; /* 0: unchecked.  Otherwise must equal the gsr_stats.generation of the forward
                        * this backward belongs to: an intervening forward on the same handle (e.g. an eval
                        * render between a training forward and its pullback) is reported as GSR_E_STATE instead
                        * of silently producing the gradients of the wrong view. */
    uint32_t flags;    /* GSR_GRADS_* (new in ABI 5, late): 0 = none */
    uint32_t reserved; /* must be 0 */
} gsr_grads;
/* gsr_grads.flags / gsr_tail_state.flags — the caller's promise that channels >= 3 of `vpixels` (depth, alpha, normal) are exact
 * zeros: the cotangent gsr_loss_l1_ssim writes — the photometric loss only sees features[1:3] (training.jl:656,684-685) — not
 * touched since.  ∇render! then runs the :rgb arithmetic on the mode's stream: the :rgbdn backward 0.885 -> 0.702 ms, :rgbd
 * 0.713 -> 0.700 (config 3).  Gradients equal the unflagged call's up to the association of fp32 sums.  Ignored in :rgb mode.
 * ABI 6: the promise is CHECKED as far as the library can see — the flag is honoured only for the very buffer the handle's
 * last gsr_loss_l1_ssim wrote for THIS forward (pointer and forward generation recorded by the loss head); any other vpixels
 * with the flag set is GSR_E_INVALID_ARG instead of silently dropped depth / normal gradients.  What it cannot see is a term
 * added to that buffer in place afterwards: GSR_CHECK_COLOR_COTANGENT=1 in the environment makes every flagged backward
 * reduce |vpixels[3:]| first (one pass over the image + a host wait — debugging only) and fail with GSR_E_INVALID_ARG when it
 * is not exactly zero. */
#define GSR_GRADS_COLOR_COTANGENT 0x1u

typedef struct gsr_handle gsr_handle;

/* GaussianRasterizer(kab; width, height, mode, ...) — rasterizer.jl:60-90. */
GSR_API int gsr_create(const gsr_config* cfg, gsr_handle** out);
/* KA.unsafe_free!(rast) — rasterizer.jl:136-145. */
GSR_API int gsr_destroy(gsr_handle* h);
/* release_scene_buffers!(rast) — rasterizer.jl:111-123. */
GSR_API int gsr_release_scene_buffers(gsr_handle* h);
/* memory_usage(rast) — rasterizer.jl:127-134 (device bytes owned by the handle). */
GSR_API int64_t gsr_memory_usage(const gsr_handle* h);
/* New in ABI 6 (no reference counterpart: the reference reallocates its states whenever the model or the instance count grew,
 * rasterizer.jl:275-278,340-343).  Pre-size the handle's grow-only scratch for views of up to n_gaussians Gaussians and
 * n_instances tile instances, so that the forwards that follow allocate nothing — a reallocation inside gsr_forward is a hipFree
 * + hipMalloc pair, i.e. a device synchronisation in the middle of a training step.  Where a trainer calls it: right after a
 * densification round changed the model (the reference empties its allocation cache at the same place, strategy.jl:92), with
 * headroom for the next rounds; the Python mirror's post_train_step does (1.5 x).  The key bins are not covered: their capacity is
 * a policy of its own (gsr_policy.h).  Never shrinks; 0 / smaller values are no-ops. */
GSR_API int gsr_reserve(gsr_handle* h, int64_t n_gaussians, int64_t n_instances);
/* Memory planning; no GPU needed.  The capacity — keys per tile — of the fixed-capacity key bins gsr_forward chooses for the view
 * AFTER one that rendered n_rendered instances with a longest tile list of max_tile_instances on a width x height image, under
 * bins_budget_bytes (0 = the default budget, see gsr_config) and with bins of current_capacity keys in place (0 = none yet; the
 * capacity only grows).  0 = no bins: compact mode.  The bins take (tiles + 1) x capacity x 8 bytes; a view whose longest list
 * exceeds the capacity scatters those lists a second time (+ 8 bytes per instance: gsr_stats.compact_binning == 2).
 * (No reference counterpart: the reference's BinningState is O(instances), states.jl:66-85.) */
GSR_API uint32_t gsr_bins_capacity_after(int64_t n_rendered, int32_t max_tile_instances, int32_t width, int32_t height,
                                         uint64_t bins_budget_bytes, uint32_t current_capacity);

/* rasterize(...) — rasterizer.jl:255-408: project! + spherical_harmonics! +
 * count_tiles_per_gaussian! + cumsum! + duplicate_with_keys! + sortperm!/_permute! +
 * identify_tile_range! + render!.  Writes image_out (C,W,H); returns an all-zero image
 * when nothing is visible (rasterizer.jl:338).  aux and stats may be NULL.
 * Synchronisation: all device work is ordered on `stream`.  The call returns after the host has read the instance
 * count of this view (the reference's one read-back, rasterizer.jl:337) — it spins on a word of pinned host memory
 * the tile scan writes; by then the sort + forward of the tiles is already queued behind the scan (it checks the
 * scan's totals against the capacity of the handle's buffers on the device), so the GPU does not wait for the host. */
GSR_API int gsr_forward(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam, float* image_out,
                        const gsr_aux* aux, void* stream, gsr_stats* stats);

/* ∇rasterize(vpixels, ...) — rasterizer.jl:416-550 (the pullback of the rrule,
 * rasterizer.jl:552-573): ∇render! + ∇project! + ∇spherical_harmonics!.
 * vpixels: device (C,W,H).  `in`/`cam` must be the ones given to gsr_forward.
 * All work is ordered on `stream`; when the view has a few very long tile lists (> 1024 instances, at most 256 such
 * tiles) their part of ∇render! runs on a stream the handle owns, forked from and joined back into `stream` with
 * events — nothing the caller has to do. */
GSR_API int gsr_backward(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam, const float* vpixels,
                         const gsr_grads* grads, void* stream);

/* How the host thread waits inside gsr_forward for the instance count (the reference blocks in a synchronous
 * device->host copy there, rasterizer.jl:337).  Process-wide; the three values are stored and read atomically (a forward
 * running on another thread sees the old or the new policy, never a mix of fields from both: they are packed in one word).
 *   sleep_us == 0 (default: 30, 0, 0): spin for `spin_us` microseconds, then poll with sched_yield() — other runnable threads
 *   (RCCL's proxy threads on an 8-rank host) get the core at once; no timers; a pure spin's step time.
 *   sleep_us > 0 (opt-in, e.g. 100, 0, 50): sleep through the expected wait (a running average per handle; the host runs
 *   ahead of the GPU, so the wait is about the same every step) except for its last `spin_us` microseconds, then
 *   sched_yield() for `yield_us`, then sleeps of `sleep_us` — a fraction of the CPU time, at the mercy of the host's timer
 *   wake-up latency.
 *   (1000000, 0, 0) is a pure spin. */
GSR_API int gsr_host_wait_policy(int spin_us, int yield_us, int sleep_us);

/* Views into the state the reference keeps in rast.gstate / bstate / istate
 * (states.jl:2-111); valid until the next gsr_forward / release on this handle.
 * gsr_buffer() copies nothing: it returns the device address and byte size. */
enum {
    GSR_BUF_RADII = 0,         /* int32 (N)        gstate.radii — read by densification (strategy.jl:85-86); the caller's
                                *                   own array when the forward was given gsr_aux.radii */
    GSR_BUF_GRAD_MEANS2D = 1,  /* float (2,N)      gstate.∇means_2d, valid after gsr_backward (the caller's own array when
                                *                   the backward was given gsr_grads.vmeans2d) */
    GSR_BUF_N_CONTRIB = 2,     /* uint32 (W,H)     istate.n_contrib */
    GSR_BUF_FINAL_T = 3,       /* float (W,H)      istate.accum_α */
    GSR_BUF_TILE_RANGES = 4,   /* uint32 (2,T)     istate.ranges */
    GSR_BUF_VALUES_SORTED = 5, /* uint32 (D)       bstate.gaussian_values_sorted (0-based ids) */
    GSR_BUF_GEOM = 6,          /* 16 x 32-bit (N): one 64-byte record per Gaussian (stale where radii == 0):
                                *   [0..1] mean2d, [2..4] conic a,b,c, [5] opacity, [6..8] rgb, [9] clamped bits (u32,
                                *   bit c = channel c), [10] depth, [11] u32 instance-slot prefix inside the 256-block,
                                *   [12] u32 rect xmin|ymin<<16, [13] u32 rect xmax|ymax<<16 (utils.jl:18-29), [14] u32 blend-test
                                *   threshold X of the Gaussian (see GSR_BUF_INSTANCE_AUX), [15] u32 bit mask of the rect's tiles (row-major) the Gaussian
                                *   was binned into (rects of at most 32 tiles) */
    GSR_BUF_NORMALS = 7,       /* float4 (N): camera-space normal (mode RGBDN only) */
    GSR_BUF_GRAD_ROWS = 8,     /* 12 (:rgb) or 16 x float per slot, Gaussian-major: the per-instance gradient rows of the last
                                *   gsr_backward.  A Gaussian whose tile rect has at most 32 tiles owns one slot per tile it
                                *   was binned into (in row-major tile order); larger rects one per tile of the rect */
    GSR_BUF_INSTANCE_AUX = 9   /* 4 x 32-bit (D), sorted instance order (plane s2 of the splat stream):
                                *   [0] blue, [1] u32 Gaussian-major slot, [2] depth (:rgbd / :rgbdn) or, in :rgb mode, the u32 blend-test
                                *   threshold X (bits(sigma) < X is the reference's blend test; 0: never blends), [3] u32 footprint masks
                                *   (bits 0..15 tile rows, 16..19 8x8 quadrants the instance can touch; bit 20: the splat is a
                                *   needle — 2-D axis ratio beyond 10 : 1 —, evaluated in the backward's accurate arithmetic) */
};
GSR_API int gsr_buffer(const gsr_handle* h, int which, const void** dev_ptr, size_t* bytes);
/* Copy of one of those buffers into caller memory, enqueued on `stream` by the library's own HIP
 * runtime (hosts that cannot wrap foreign device pointers — and must not dlopen a second HIP
 * runtime to copy them — use this).  dst: device, at least `bytes` bytes; bytes must not exceed
 * what gsr_buffer reports. */
GSR_API int gsr_copy_buffer(const gsr_handle* h, int which, void* dst, size_t bytes, void* stream);

/* update_stats!(strategy, rast.gstate.radii, rast.gstate.∇means_2d, resolution) —
 * src/strategy.jl:107-136 (`_update_stats!`): for every Gaussian visible in the last
 * forward, max_radii = max(max_radii, radius), accum += ‖∇mean_2d · resolution · 0.5‖,
 * denom += 1.  All three are caller-owned device arrays of N elements (the DefaultStrategy's
 * densification state); needs a completed gsr_forward/gsr_backward pair on the handle. */
GSR_API int gsr_update_stats(gsr_handle* h, int32_t* max_radii, float* accum_grad_means2d, float* denom, void* stream);

/* _fused_ssim / fused_ssim_bwd — src/fused_ssim.jl:373-408.  Arrays are (W,H,CH,B),
 * x fastest.  With train == 0 the three partial-derivative maps may be NULL. */
GSR_API int gsr_ssim_forward(int W, int H, int CH, int B, const float* img, const float* ref, float C1, float C2,
                             int train, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                             void* stream);
GSR_API int gsr_ssim_backward(int W, int H, int CH, int B, const float* img, const float* ref, const float* dL_dmap,
                              const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                              float* dL_dimg, void* stream);

/* Arithmetic of the three SSIM entry points (gsr_ssim_forward / gsr_ssim_backward / gsr_loss_l1_ssim): the PROCESS-WIDE
 * DEFAULT — what gsr_ssim_forward / gsr_ssim_backward (no handle) use, and gsr_loss_l1_ssim on a handle whose
 * gsr_config.ssim_precision is GSR_DEFAULT; a handle created with GSR_SSIM_FAST / GSR_SSIM_EXACT there is not affected.  Stored and read
 * atomically; gsr_get_ssim_precision returns the current value (so that a scoped override can restore what it found).
 *   0 (default): multiply-adds contracted to FMAs and the six divisions of the SSIM formula (fused_ssim.jl:219-233) taken over
 *      two hardware reciprocals — what a GPU compiler makes of the reference's own source; results agree with the
 *      fp32-as-written evaluation to ~1e-6 relative (the loss to 1e-6 absolute, its pullback to 1e-5 relative L2);
 *   1: every fp32 operation as written, IEEE divisions: bit-identical to the CPU oracle (the parity tests' twin), ~35 us
 *      slower per 1080p loss evaluation.
 * GSR_SSIM_EXACT=1 in the environment starts the process in mode 1. */
GSR_API int gsr_ssim_precision(int exact);
GSR_API int gsr_get_ssim_precision(void);

/* Form of the binning inside gsr_forward's first kernel (preprocess: projection.jl:69-129 + spherical_harmonics! +
 * utils.jl:85-142 fused): the PROCESS-WIDE DEFAULT, used by handles whose gsr_config.preprocess_form is GSR_DEFAULT (a
 * handle created with GSR_PREPROCESS_DIRECT / GSR_PREPROCESS_AGGREGATING there is pinned).  Outputs are identical in every form (only the arbitrary order of the unsorted
 * keys inside a tile's bin differs), this is a performance switch and the tests' handle on both code paths.
 *  -1 (default): chosen per call — the aggregating form for scenes of >= 250 000 Gaussians on grids whose counter words
 *      fit the LDS three times per CU (up to ~10 700 tiles with 2 x 32-bit words: 1080p; up to ~21 500 with 2 x 16-bit words
 *      while the bins' capacity is below 65 024: 1440p; 4K no), else the direct form;
 *   0: always the direct form (one returning global atomic per instance pair; the rect walk is spread evenly over the
 *      lanes of each wave, as in the aggregating form);
 *   1: the aggregating form wherever its LDS fits (a workgroup adds its requests up per counter word in LDS and issues
 *      one global atomic per word, in address order; both rect walks are spread evenly over the lanes of each wave).
 * GSR_PREPROCESS_AGG=0/1 in the environment starts the process in mode 0 / 1. */
GSR_API int gsr_preprocess_form(int form);
GSR_API int gsr_get_preprocess_form(void);

/* The photometric loss head of Trainer.step! — src/training.jl:656,684-694:
 *   image = features[1:3,:,:]; permute to (W,H,3,1);
 *   L = (1-lambda)*mean|image-target| + lambda*(1-mean(fused_ssim(image; ref=target)))
 * and its pullback to the rasterizer output.  image: device (C,W,H) as written by
 * gsr_forward; target: device (W,H,3); loss_out: device scalar; vpixels: device (C,W,H)
 * (channels >= 3 zeroed).  Scratch is owned by the handle. */
GSR_API int gsr_loss_l1_ssim(gsr_handle* h, const float* image, const float* target, float lambda_dssim,
                             float* loss_out, float* vpixels, void* stream);

/* The functor prologue `(rast::GaussianRasterizer)(means_3d, opacities, scales, rotations,
 * sh_color, sh_remainder, ...)` (rasterizer.jl:200-253) up to its call of rasterize():
 *   shs = hcat(sh_color (3,1,N), sh_remainder (3,k_rest,N))          (rasterizer.jl:218-228)
 *   opacities_act = NU.sigmoid.(opacities (1,N))                     (rasterizer.jl:229-234)
 *   scales_act = exp.(scales), an isotropic (1,N) scale tiled x3     (rasterizer.jl:235-247)
 * scale_dims is 1 (isotropic) or 3.  k_rest may be 0 (sh_remainder NULL).  All device. */
GSR_API int gsr_prologue_forward(int32_t n, int32_t k_rest, int32_t scale_dims, const float* sh_color,
                                 const float* sh_remainder, const float* opacities, const float* scales, float* shs,
                                 float* opacities_act, float* scales_act, void* stream);
/* Pullback of the prologue — what Zygote derives around the rrule of rasterize
 * (rasterizer.jl:552-573) for the raw parameters the trainer optimises
 * (training.jl:646-656): vshs is split, v_opacities = v_act * a(1-a), v_scales = v_act * exp(s)
 * (summed over the three tiled rows when isotropic). */
GSR_API int gsr_prologue_backward(int32_t n, int32_t k_rest, int32_t scale_dims, const float* opacities_act,
                                  const float* scales_act, const float* vshs, const float* vopacities_act,
                                  const float* vscales_act, float* v_sh_color, float* v_sh_remainder,
                                  float* v_opacities, float* v_scales, void* stream);

/* `NU.step!(opt, θ, ∇)` of NerfUtils 0.2's Adam (external dependency, Project.toml:76; call
 * sites training.jl:234-239,778), for up to GSR_ADAM_MAX_GROUPS parameter arrays in ONE
 * launch.  Each group is one `NU.Adam` (its own lr; μ, ν of the parameter's length).
 * `current_step` is the optimizer's counter AFTER its increment (1 on the first step).
 *   μ = β1 μ + (1-β1) g;  ν = β2 ν + (1-β2) g²;
 *   θ -= lr · sqrt(1-β2^t)/(1-β1^t) · μ / (sqrt(ν) + ϵ)            (ϵ = 1f-15: training.jl:229)
 * theta, mu, nu are updated in place; grad is read only.  All device. */
#define GSR_ADAM_MAX_GROUPS 8
typedef struct gsr_adam_group {
    float* theta;
    const float* grad;
    float* mu;
    float* nu;
    int64_t count; /* elements */
    float lr;
    uint32_t current_step;
} gsr_adam_group;
GSR_API int gsr_adam_step(const gsr_adam_group* groups, int32_t n_groups, float beta1, float beta2, float eps,
                          void* stream);

/* The whole trainer tail of `step!` in one pass over the parameters (SURVEY.md §8f rank 1:
 * "fusing it removes 3 more round-trips"): pullback of the functor prologue
 * (rasterizer.jl:218-247) + `NU.step!` on the six optimizers (training.jl:768-779) + the
 * prologue of the NEXT forward.  Group order everywhere: points, features_dc, features_rest,
 * opacities, scales, rotations (training.jl:415-416).
 *   grads: the gradients gsr_backward wrote, w.r.t. the ACTIVATED values
 *   theta/mu/nu[6], lr[6], current_step[6] (counters AFTER increment): the raw parameters and
 *          the NU.Adam states, updated in place
 *   shs (3,K,N), opacities_act (1,N), scales_act (3,N): on entry the activated copies the
 *          forward of THIS step used (σ' and exp' are taken from them), on exit those of the
 *          updated parameters, ready for the next gsr_forward.
 * Bit-identical θ, μ, ν to gsr_prologue_backward + gsr_adam_step; features_rest may be empty
 * (k_rest = 0). */
typedef struct gsr_tail_grads {
    const float* vmeans;        /* (3,N) */
    const float* vshs;          /* (3,K,N) */
    const float* vopacities;    /* (1,N)  w.r.t. sigmoid(opacities) */
    const float* vscales;       /* (3,N)  w.r.t. exp(scales) */
    const float* vrotations;    /* (4,N) */
} gsr_tail_grads;
GSR_API int gsr_trainer_tail_step(int32_t n, int32_t k_rest, int32_t scale_dims, const gsr_tail_grads* grads,
                                  float* const theta[6], float* const mu[6], float* const nu[6], const float lr[6],
                                  const uint32_t current_step[6], float beta1, float beta2, float eps, float* shs,
                                  float* opacities_act, float* scales_act, void* stream);

/* Single-GPU trainer step: gsr_backward with gsr_trainer_tail_step applied in its epilogue
 * ("fuse the trainer tail into the per-Gaussian backward").  Equivalent to
 *     gsr_backward(h, in, cam, vpixels, &grads, stream);
 *     gsr_trainer_tail_step(n, K-1, scale_dims, &grads, theta, mu, nu, ..., stream);
 * with bit-identical θ, μ, ν and activated copies, but the 59·N gradient floats are never written:
 * they live in registers between ∇project / ∇spherical_harmonics and the Adam update.  Not for the
 * multi-GPU step (the gradients must be exchanged before the update) and not with pose gradients.
 *   in : the inputs of the matching gsr_forward.  They must BE the trainer's arrays —
 *        in->means == theta[0], in->rotations == theta[5], in->shs == shs,
 *        in->opacities == opacities_act, in->scales == scales_act (checked; GSR_E_INVALID_ARG) —
 *        because they are updated in place.  K = in->n_coeffs; features_rest holds K-1 bands.
 *   vmeans2d : (2,N) gstate.∇means_2d of this backward, or NULL (handle-owned GSR_BUF_GRAD_MEANS2D)
 *   forward_generation : gsr_stats.generation of the forward being differentiated, or 0 */
typedef struct gsr_tail_state {
    float* theta[6];            /* points, features_dc, features_rest, opacities, scales, rotations */
    float* mu[6];
    float* nu[6];
    float lr[6];
    uint32_t current_step[6];   /* counters AFTER increment (NU.Adam counts from 1) */
    float beta1, beta2, eps;
    int32_t scale_dims;         /* 3, or 1 for isotropic scenes */
    float* shs;                 /* (3,K,N) */
    float* opacities_act;       /* (1,N) */
    float* scales_act;          /* (3,N) */
    float* vmeans2d;
    uint64_t forward_generation;
    uint32_t flags;             /* GSR_GRADS_* (gsr_grads.flags) */
    uint32_t reserved;          /* must be 0 */
} gsr_tail_state;
GSR_API int gsr_backward_trainer_tail(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam,
                                      const float* vpixels, const gsr_tail_state* st, void* stream);

/* Boolean-mask compaction of per-Gaussian arrays — the `x[:, mask]` / `x[:, :, mask]` /
 * `x[mask]` logical indexing that `prune_points!`, `densify_clone!`, `densify_split!` and
 * `_prune_optimizer!` apply to every parameter, both Adam moments and the densification
 * statistics (src/densification.jl:29-62,64-121,138-191,279-288).
 *   gsr_mask_findall : indices (0-based, ascending) of the non-zero bytes of mask[n] — Julia's
 *                      `findall(mask)`; *count_out (device) receives their number.  `scratch`
 *                      is gsr_mask_findall_scratch_bytes(n) bytes of device memory.
 *   gsr_gather_rows  : dst[g][r, :] = src[g][indices[r], :] for up to GSR_ADAM_MAX_GROUPS arrays
 *                      in one launch; a row is row_words 4-byte words (float / int32 / uint32).
 * All pointers device; order-preserving, bit-exact. */
typedef struct gsr_gather_group {
    const void* src;
    void* dst;
    int32_t row_words;
} gsr_gather_group;
GSR_API size_t gsr_mask_findall_scratch_bytes(int64_t n);
GSR_API int gsr_mask_findall(const uint8_t* mask, int64_t n, uint32_t* indices, uint32_t* count_out, void* scratch,
                             void* stream);
GSR_API int gsr_gather_rows(const gsr_gather_group* groups, int32_t n_groups, const uint32_t* indices, int64_t count,
                            void* stream);

/* Adaptive density control of the reference's DefaultStrategy on the device (SURVEY.md §8f rank 3;
 * src/densification.jl:1-297, src/gaussians.jl:119-137).  The host keeps the reference's control flow
 * (densify_and_prune! -> densify_clone! -> densify_split! -> prune_points!, strategy.jl:78-105); every
 * per-Gaussian pass is one of these launches.  All pointers device, caller-owned.
 *
 * gsr_densify_grad_mean : ∇means_2d = accum ./ denom with NaN -> 0                 (densification.jl:7-10)
 * gsr_densify_mask      : kind GSR_DENSIFY_CLONE  mask = grad >  thr && max(exp(scales)) < gamma   (:34-38)
 *                              GSR_DENSIFY_SPLIT  mask = grad >= thr && max(exp(scales)) > gamma   (:73-80; `grad`
 *                                                 holds n_grad <= n entries, rows beyond it count as 0: padded_grad)
 *                              GSR_DENSIFY_PRUNE  mask = sigmoid(opacity) > min_opacity, and when max_screen_size > 0
 *                                                 && max_radii < max_screen_size && max(exp(scales)) < gamma  (:18-25)
 *                         scales are the RAW log-scales (scale_dims = 3, or 1 for an isotropic model).
 * gsr_compose_rows      : the row surgery of append_gaussians! / _append_optimizer! / prune_points! /
 *                         _prune_optimizer! (:138-297) for up to GSR_COMPOSE_MAX_GROUPS arrays in one launch:
 *                           dst[r]                      = src[keep_idx[r]]                 r < n_keep (keep_idx NULL: r)
 *                           dst[n_keep + k*n_sel + j]   = new_zero ? 0 : src[sel_idx[j]]   j < n_sel, k < reps
 *                         clone: keep = identity, sel = findall(mask), reps 1; split: keep = findall(!mask),
 *                         sel = findall(mask), reps 2 (Julia's block `repeat`); prune: keep = findall(valid), n_sel 0.
 *                         new_zero = 1 for the Adam moments (their new rows are zeros).
 * gsr_split_transform   : on the 2m rows a split appended: sigma = exp(scale); point += R(q)·(sigma .* randn3);
 *                         scale = log(sigma / 1.6)   (:81-104, _add_split_noise! :121-135).  The normals come from a
 *                         counter-based generator keyed by (seed, row) — the reference uses the backend's device RNG.
 * gsr_reset_opacity     : opacity = inverse_sigmoid(min(0.1, sigmoid(opacity)))  (gaussians.jl:119-126,137). */
enum { GSR_DENSIFY_CLONE = 0, GSR_DENSIFY_SPLIT = 1, GSR_DENSIFY_PRUNE = 2 };
#define GSR_COMPOSE_MAX_GROUPS 24
typedef struct gsr_compose_group {
    const void* src;
    void* dst;
    int32_t row_words; /* 4-byte words per row */
    int32_t new_zero;  /* appended rows are zeros (Adam moments) instead of copies */
} gsr_compose_group;
GSR_API int gsr_densify_grad_mean(int64_t n, const float* accum_grad_means2d, const float* denom, float* grad_out, void* stream);
GSR_API int gsr_densify_mask(int32_t kind, int64_t n, int64_t n_grad, const float* grad, const float* scales,
                             int32_t scale_dims, const float* opacities, const int32_t* max_radii, float grad_threshold,
                             float gamma, float min_opacity, int32_t max_screen_size, uint8_t* mask, void* stream);
GSR_API int gsr_compose_rows(const gsr_compose_group* groups, int32_t n_groups, const uint32_t* keep_idx, int64_t n_keep,
                             const uint32_t* sel_idx, int64_t n_sel, int32_t reps, void* stream);
GSR_API int gsr_split_transform(int64_t n_new, int32_t scale_dims, float* points, const float* rotations, float* scales,
                                uint32_t seed, void* stream);
GSR_API int gsr_reset_opacity(int64_t n, float* opacities, void* stream);
/* Not in the reference: 63-bit Morton (Z-order) codes of the Gaussians' positions inside the host-given box (21 bits per
 * axis) — the sort key of the OPTIONAL spatial re-sort a trainer may run after a densification (the arrays are composed
 * anyway there: gsr_compose_rows with keep_idx = the sorting permutation, `gs.ids` keeps the identities).  Spatially ordered
 * Gaussians make the binning's counter traffic and the tile sort's record gathers coherent: config 3 steps in 1.41 ms
 * instead of 1.43.  box_lo / box_hi: host pointers to 3 floats. */
GSR_API int gsr_morton_codes(int64_t n, const float* points, const float* box_lo, const float* box_hi, uint64_t* codes,
                             void* stream);

/* The non-finite gradient guard of `step!` under GSP_DEBUG (src/training.jl:772-777: `isfinite(sum(∇ᵢ))` per parameter)
 * and the per-parameter counts of `nonfinite_gradient_report` (:534-552), in one pass over up to GSR_ADAM_MAX_GROUPS
 * gradient arrays of n_rows Gaussians each (rows of row_words floats; row_words = 0 skips an empty array):
 *   counts[g]    = number of Gaussians whose row holds a NaN or ±Inf          (device, n_groups)
 *   first_bad[g] = smallest such Gaussian index, 0xFFFFFFFF if none           (device, n_groups) */
GSR_API int gsr_count_nonfinite(const float* const* arrays, const int32_t* row_words, int32_t n_groups, int64_t n_rows,
                                uint32_t* counts, uint32_t* first_bad, void* stream);

/* The vertex rows of a 3DGS .ply scene (`export_ply` / `import_ply`, src/gaussians.jl:140-247):
 * per Gaussian  x y z | nx ny nz (0) | f_dc_0..2 | f_rest_0..3·k_rest-1 (channel-major) | opacity | scale_0..2 | rot_0..3,
 * i.e. 17 + 3·k_rest floats of RAW parameters.  gsr_ply_pack_rows gathers the model's arrays (device, the reference's
 * layouts: points (3,N), features_dc (3,1,N), features_rest (3,k_rest,N), opacities (1,N), scales (3,N), rotations (4,N))
 * into the row matrix `rows` (device, N x (17 + 3·k_rest)); gsr_ply_unpack_rows scatters a row matrix back (normals are
 * ignored).  Header and disk I/O stay with the host. */
GSR_API int gsr_ply_pack_rows(int64_t n, int32_t k_rest, const float* points, const float* features_dc,
                              const float* features_rest, const float* opacities, const float* scales, const float* rotations,
                              float* rows, void* stream);
GSR_API int gsr_ply_unpack_rows(int64_t n, int32_t k_rest, const float* rows, float* points, float* features_dc,
                                float* features_rest, float* opacities, float* scales, float* rotations, void* stream);

/* New: the SH-coefficient gradient of a batch of views from the factored
 * per-view colour cotangents written by gsr_backward (gsr_grads.vcolors):
 *   vshs[:, k, i] = Σ_v basis_k(normalize(means[:, i] - camera_centers[:, v])) * vcolors_all[:, i, v]
 * — ∇spherical_harmonics! (spherical_harmonics.jl:32-37) of every view, summed in ascending view
 * order.  camera_centers: device (3,V); vcolors_all: device (3,N,V) (view slowest), e.g. the output
 * of an all-gather over the ranks; vshs: device (3,K,N), fully overwritten (bands above
 * sh_degree get zeros).  With V = 1 the result is bit-identical to gsr_backward's own vshs. */
GSR_API int gsr_sh_grad_from_views(int32_t n, int32_t n_coeffs, int32_t sh_degree, int32_t n_views,
                                   const float* camera_centers, const float* means, const float* vcolors_all,
                                   float* vshs, void* stream);

/* New in ABI 4 (SURVEY.md §8f-1 for the multi-GPU step; training.jl:768-779): gsr_sh_grad_from_views with the trainer tail in
 * place of the ∇shs store.  After the factored exchange — the 11·N small gradients all-reduced, the per-view colour
 * cotangents all-gathered — ONE pass rebuilds Σ_v basis(dir_v) x vc_v per Gaussian, keeps it on chip, and applies the
 * pullback of the functor prologue + the six NU.Adam updates + the activated copies of the next forward.  Equivalent to
 *     gsr_sh_grad_from_views(n, K, deg, V, centers, st->theta[0], vcolors_all, vshs, stream);
 *     gsr_trainer_tail_step(n, K-1, st->scale_dims, {small..., vshs}, st->theta, st->mu, st->nu, ..., stream);
 * with bit-identical θ, μ, ν and activated copies, without the 3K·N-float ∇shs ever being written or read.
 *   small : vmeans, vopacities, vscales, vrotations of the all-reduced block (w.r.t. the ACTIVATED opacity / scale);
 *           small->vshs is not read (may be NULL)
 *   st    : as for gsr_backward_trainer_tail (theta[0] = the points the view directions are taken from; vmeans2d and
 *           forward_generation are not used). */
GSR_API int gsr_sh_grad_from_views_tail(int32_t n, int32_t n_coeffs, int32_t sh_degree, int32_t n_views,
                                        const float* camera_centers, const float* vcolors_all,
                                        const gsr_tail_grads* small, const gsr_tail_state* st, void* stream);

/* New (no reference counterpart): sum the per-view gradient arena over
 * the ranks of an RCCL communicator (ncclComm_t passed as void*).  librccl is resolved
 * lazily with dlopen, so single-GPU users need not have it. */
GSR_API int gsr_allreduce_grads(void* nccl_comm, float* arena, size_t count, void* stream);

/* New (measurement): STREAM triad a = b + q*c over `count` floats — the
 * device-to-device bandwidth bench.py measures in the same run as the hot path and reports
 * next to the 8 TB/s data-sheet peak (12 bytes of HBM traffic per element). */
GSR_API int gsr_stream_triad(float* a, const float* b, const float* c, size_t count, float q, void* stream);

/* New (measurement; the reference has no profiling hooks): per-stage kernel
 * timing with HIP events recorded on the caller's stream around each launch.  While
 * enabled, every gsr_forward / gsr_backward / gsr_loss_l1_ssim appends one event pair per
 * stage; gsr_profile_read() waits for them and returns, per stage, the summed
 * milliseconds and the number of launches (arrays of gsr_profile_stage_count() entries). */
/* `on` is a bit set: 1 = HIP-event stage timing (above), 2 = one roctx range "gsr:<stage>" per stage around its
 * launches (lets `rocprofv3 --marker-trace --kernel-trace` slice a trace by stage; the roctx library
 * is resolved lazily with dlopen; GSR_ROCTX=1 in the environment enables the ranges on every handle). */
GSR_API int gsr_profile_enable(gsr_handle* h, int on);
/* Restrict the event pairs to the stages whose bit is set (bit i = stage i of gsr_profile_stage_name; default all).
 * An event record is a marker packet between two kernels: eight pairs per step cost ≈ 0.06 ms of a 1.8 ms step on
 * MI355X, so a benchmark times its dominant stage alone inside the timed region and surveys the rest outside it. */
GSR_API int gsr_profile_stages(gsr_handle* h, uint32_t stage_mask);
GSR_API int gsr_profile_stage_count(void);
GSR_API const char* gsr_profile_stage_name(int stage);
GSR_API int gsr_profile_read(gsr_handle* h, double* ms_sum, int* launches, int reset);
/* Time from each recorded launch of `stage` to the next one (begin event to begin event), in milliseconds, oldest
 * first: with one launch of the stage per step these are the per-step times of a run, from the event pairs that are
 * on the stream anyway (a separate per-step marker would be one more bubble).  Writes min(count-1, max_n) values,
 * *n_out = count-1 (0 when fewer than two launches are recorded).  Does not reset. */
GSR_API int gsr_profile_read_intervals(gsr_handle* h, int stage, double* ms_out, int max_n, int* n_out);

GSR_API const char* gsr_last_error_string(void);
GSR_API const char* gsr_version(void);   /* "gsr-hip <release> abi <GSR_ABI_VERSION> (gfx950)" */
GSR_API int gsr_abi_version(void);        /* GSR_ABI_VERSION the library was built from */
/* Binding self-check, to be called once by every binding right after it loads the library (the Python mirror and the
 * Julia binding do): the caller's GSR_ABI_VERSION and its sizeof of the six structs that cross gsr_create / gsr_forward
 * / gsr_backward, and (ABI 6) of gsr_tail_state, which grew late in ABI 5.  Any mismatch — a caller built against another header — is GSR_E_INVALID_ARG with a message naming the
 * struct, instead of mis-sized structs being read silently. */
GSR_API int gsr_check_abi(int abi_version, size_t sizeof_config, size_t sizeof_inputs, size_t sizeof_camera,
                          size_t sizeof_aux, size_t sizeof_stats, size_t sizeof_grads, size_t sizeof_tail_state);

#ifdef __cplusplus
}
#endif

/* The library's policies as GPU-free functions (bins capacity, binning mode and form, the form tuner's rule, the held
 * fused launch, the backward's list split) and the view-history state gsr_stats reports from. */
#include "gsr_policy.h"

#endif /* GSR_H */
