/*
 * gsr_policy.h — the POLICIES of libgsr_hip.so as pure, GPU-free functions (round-5 verdict "next #8").
 *
 * gsr_forward carries state keyed on "the previous view": the capacity of the fixed-capacity key bins, whether the handle has
 * given up on bins (compact mode), which binning form the first kernel runs in (and, on grids where two forms are candidates,
 * a four-view timing experiment that decides it), whether the fused sort + forward launch is held back for the tier tiles,
 * and which tiers the backward splits along their lists.  Every one of those DECISIONS is made here, by functions of plain
 * numbers — no HIP call, no clock, no environment read after the first call — and gsr_forward / gsr_backward only execute
 * what they return.  The library calls exactly these functions (csrc/gsr_api.cpp), so a host-side test can replay a recorded
 * history of views — (N, instances, longest list, tier tiles) per view — through gsr_policy_begin_view / gsr_policy_end_view
 * and assert what a training run will do: how often the bins are regrown, whether a view ever falls back to the compact
 * mode, how often the form tuner is re-armed (tests/test_policy.py; the histories are recorded by tools/train_harness.py).
 *
 * The reference has no counterpart: its BinningState is O(instances) and rebuilt per view (states.jl:66-85,
 * rasterizer.jl:340-378); its knobs are constructor keywords (rasterizer.jl:60-65), as gsr_policy_config's are here.
 * Everything in this header works without a GPU.
 */
#ifndef GSR_POLICY_H
#define GSR_POLICY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef GSR_API
#define GSR_API __attribute__((visibility("default")))
#endif

/* Binning form of the forward's first kernel (gsr_stats.preprocess_form reports the same numbers). */
enum {
    GSR_FORM_DIRECT = 0,       /* one returning global atomic per instance pair */
    GSR_FORM_AGG_W64 = 1,      /* aggregating: the grid's counter words in LDS as 2 x 32-bit pairs */
    GSR_FORM_AGG_W32 = 2,      /* aggregating: 2 x 16-bit pairs */
    GSR_FORM_AGG_BANDED = 3    /* aggregating over horizontal bands of the tile grid */
};

/* How a view was binned (gsr_stats.compact_binning reports the same numbers). */
enum {
    GSR_BINNING_BINS = 0,      /* fixed-capacity bins held every list */
    GSR_BINNING_COMPACT = 1,   /* count -> scan -> scatter for the whole view */
    GSR_BINNING_OVERFLOW = 2   /* bins, and the lists beyond their capacity scattered a second time */
};

/* The four-view timing experiment that picks the binning form where two are candidates (4K grids): views are timed in the
 * order direct, aggregating, direct, aggregating; the faster view of each form counts; the aggregating form must win by 3 %.
 * Re-armed when the scene grew or shrank by a quarter, or after 4096 views. */
typedef struct gsr_form_tuner {
    int32_t phase;      /* 0..3: the view to time next (form = phase & 1); 4: waiting for the events; 5: decided */
    int32_t form;       /* the decision (0 direct / 1 aggregating), -1 before it */
    int32_t n_ref;      /* Gaussians at decision time */
    uint32_t age;       /* views since the decision */
    float ms[2];        /* the measurement the decision was taken from: faster view of each form */
} gsr_form_tuner;
#define GSR_TUNER_TIMED_VIEWS 4

/* What a handle was configured with (the policy-relevant part of gsr_config + the process defaults, resolved by the caller). */
typedef struct gsr_policy_config {
    int32_t grid_x, grid_y;        /* tiles */
    uint64_t bins_budget_bytes;    /* gsr_config.bins_budget_bytes (0 = default: max(512 MiB, 160 B x last instance count)) */
    int32_t preprocess_form;       /* requested form: -1 by scene and grid, 0 direct, 1 aggregating */
    int32_t form_tuner;            /* 1: measure where two forms are candidates; 0: the previous view's skew decides */
    uint32_t beside_max_tiles;     /* hold the fused launch only when the previous view had at most this many tier tiles */
    uint32_t bwd_split_max_tiles;  /* the backward splits long lists only while at most this many tiles have one (256) */
    int32_t agg_max_bands;         /* bands the default form choice accepts without a skew hint / measurement (1) */
    int32_t reserved;
} gsr_policy_config;

/* The view-history state of one handle.  Zero-initialise (then tuner.form = -1: gsr_policy_state_init does both). */
typedef struct gsr_policy_state {
    uint32_t bin_cap;            /* capacity (keys per tile) the NEXT view's bins will have; 0 = none chosen yet / none */
    uint32_t compact_sticky;     /* the last view showed that no bins fit the budget: stay in compact mode */
    int32_t last_n;              /* previous view: Gaussians */
    uint32_t last_max_tile;      /* ... longest tile list */
    int64_t last_n_rendered;     /* ... tile instances D */
    uint32_t tier_n[3];          /* ... tiles with lists in (1024, 4096], (4096, 8192], > 8192 */
    uint32_t bin_cap_view;       /* capacity the CURRENT view's bins were filled with (0: count only) */
    uint64_t views;              /* forwards begun on the handle */
    gsr_form_tuner tuner;
    /* history counters, cumulative since gsr_create (gsr_stats reports them) */
    uint32_t bins_regrowths;     /* views after which the bins' capacity grew while bins were in use (a reallocation) */
    uint32_t compact_fallbacks;  /* views whose bins (of < 1024 keys) overflowed: filled in vain, binned again compactly */
    uint32_t compact_views;      /* views binned compactly because no bins fit the budget */
    uint32_t overflow_views;     /* views that kept their bins and scattered only the lists beyond them */
    uint32_t tuner_rearms;       /* times the form tuner was started again after a decision */
    uint32_t fused_relaunches;   /* early fused launches that found the per-instance buffers too small (a no-op + a redo) */
    uint32_t held_views;         /* views whose fused launch was held for the tier tiles */
    uint32_t reserved;
} gsr_policy_state;

/* Decisions BEFORE a view's first kernel. */
typedef struct gsr_view_plan {
    uint32_t bin_cap_view;   /* keys per tile of this view's bins; 0 = preprocess only counts (compact mode) */
    int32_t form_request;    /* what the first kernel is asked for: -1 default rule (with `skewed`), 0 direct, 1 aggregating */
    int32_t form;            /* the form that will run: GSR_FORM_* */
    int32_t timed_slot;      /* 0..3: this view's first kernel is one of the tuner's timed views; -1: not timed */
    int32_t skewed;          /* the previous view's longest list was > 6 x its mean list (hot counter words) */
    int32_t hold_fused;      /* the previous view had tier tiles: hold the fused launch until their sorts have run */
    int32_t tuner_decided;   /* this call took the tuner's decision (from `timed_ms`) */
    /* Speculative sorts of the two mid tiers (lists of 1024 ... 4096 / ... 8192 keys), queued behind the scan BEFORE the host has the
     * counts when the fused launch is held: grid sizes guessed from the previous view (count x 1.25 + 16, at most every tile);
     * 0 = none (no hold, or no such tiles last view).  Lists beyond 8192 are always sorted after the read-back.  The
     * caller zeroes them when it does not launch (forward-only render) — gsr_policy_end_view reads them back. */
    uint32_t spec_mid4, spec_mid8;
    int32_t reserved;
} gsr_view_plan;

/* Decisions AFTER the scan has published a view's counts. */
typedef struct gsr_view_outcome {
    int32_t binning;         /* GSR_BINNING_* */
    int32_t fused_done;      /* the early / held fused launch covers every list of up to 1024 instances */
    int32_t long_tiles;      /* the view has tier tiles */
    int32_t beside;          /* the tier walk runs beside the (held) fused launch */
    int32_t launch_fused_now;/* a held fused launch with no reason to be held any longer: launch it before the tier work */
    int32_t reserved;
    /* leading tiles of the two mid tier lists that the speculative sorts have covered (their device-side guard passed: the view
     * fits the buffers and no list exceeds the bins' capacity): the host sorts the rest */
    uint32_t sorted_mid4, sorted_mid8;
    uint32_t bin_cap_next;   /* capacity chosen for the next view (0: compact) */
    uint32_t bins_regrown;   /* this view made the bins grow */
} gsr_view_outcome;

/* Which tiers gsr_backward takes out of the one-wave-per-tile launch and walks in list segments. */
typedef struct gsr_bwd_split {
    uint32_t n_big, n_mid8, n_mid4;  /* tiles taken from each tier (0: the tier stays in the main launch) */
    uint32_t split_len;              /* lists longer than this are split (8192 / 4096 / 1024); 0xFFFFFFFF: none */
} gsr_bwd_split;

/* Defaults: tuner on, hold for any number of tier tiles, split up to 256 tiles, one band.  Environment overrides of the
 * library (GSR_FORM_TUNER, GSR_TIERS_BESIDE_MAX, GSR_BWD_SPLIT_TILES, GSR_AGG_MAX_BANDS: A/B runs) are applied by gsr_create,
 * not here. */
GSR_API void gsr_policy_config_init(gsr_policy_config* cfg, int32_t width, int32_t height, uint64_t bins_budget_bytes,
                                    int32_t preprocess_form);
GSR_API void gsr_policy_state_init(gsr_policy_state* st);

/* The aggregating form's LDS plan for a grid and a largest position (bins capacity, or longest list in the scatter pass):
 * number of bands, tile rows per band, dynamic LDS bytes, 16-bit words or not. */
GSR_API void gsr_policy_agg_plan(int32_t grid_x, int32_t grid_y, uint32_t max_pos, int32_t* n_bands, int32_t* band_rows,
                                 size_t* lds_bytes, int32_t* words16);
/* The form the first kernel runs in for a request (-1 / 0 / 1), a scene size, a grid, a bins capacity and the skew hint. */
GSR_API int32_t gsr_policy_preprocess_form(const gsr_policy_config* cfg, int32_t form_request, int32_t n, uint32_t bin_cap_view,
                                           int32_t skewed);
/* 1 where the default rule has two candidates (scenes of >= 250 k Gaussians on grids of 2..8 bands): the tuner's territory. */
GSR_API int32_t gsr_policy_form_is_open(const gsr_policy_config* cfg, int32_t n, uint32_t bin_cap_view);

/* One view: begin (before the first kernel), then end (after the scan has published the counts).
 *   timed_ms      : NULL, or — once st->tuner.phase == GSR_TUNER_TIMED_VIEWS and all four timed views have completed — their
 *                   first kernels' milliseconds (view 0..3 = direct, aggregating, direct, aggregating): the decision is taken
 *                   in this call.  The caller owns the clock; the rule (faster view of each form, 3 % margin) is here.
 *   cap_instances : how many instances the handle's per-instance buffers hold (the early fused launch checks the same
 *                   number on the device).
 *   fused_allowed : the fused sort + forward launch applies to this view at all (bins in use, buffers exist, not disabled). */
GSR_API void gsr_policy_begin_view(const gsr_policy_config* cfg, gsr_policy_state* st, int32_t n,
                                   const float* timed_ms /* GSR_TUNER_TIMED_VIEWS values or NULL */, gsr_view_plan* plan);
GSR_API void gsr_policy_end_view(const gsr_policy_config* cfg, gsr_policy_state* st, const gsr_view_plan* plan,
                                 int64_t n_rendered, uint32_t max_tile_instances, uint32_t n_mid4, uint32_t n_mid8,
                                 uint32_t n_big, uint64_t cap_instances, int32_t fused_allowed, gsr_view_outcome* out);

GSR_API void gsr_policy_bwd_split(const gsr_policy_config* cfg, uint32_t n_mid4, uint32_t n_mid8, uint32_t n_big,
                                  gsr_bwd_split* out);

#ifdef __cplusplus
}
#endif
#endif /* GSR_POLICY_H */
