// The policies of libgsr_hip.so (include/gsr_policy.h): pure functions of plain numbers — no HIP call, no clock, no
// environment.  gsr_api.cpp executes what they decide; tests/test_policy.py replays recorded view histories through them.
// Compiles with any C++17 compiler (the CPU tests also build it with g++ alone: tests/test_policy.py).
#include "../../include/gsr.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "agg_plan.h"

namespace {

constexpr int kTile = 16;  // GaussianSplatting.jl:55 BLOCK (gsr_kernels.h GSR_TILE)
// default budget of the fast binning mode's fixed-capacity bins (gsr_config.bins_budget_bytes = 0)
constexpr uint64_t kBinsBudgetMin = 512ull << 20, kBinsBudgetPerInstance = 160ull;
constexpr uint32_t kHybridMinCap = 1024u;  // the overflow path needs bins that hold every list of the fused launch
constexpr uint32_t kTunerMaxAge = 4096u;

inline uint64_t round64_slack(uint64_t v) { return (v + v / 4 + 63) & ~63ull; }
inline uint64_t budget_of(uint64_t configured, uint64_t instances) {
    return configured ? configured : std::max<uint64_t>(kBinsBudgetMin, kBinsBudgetPerInstance * instances);
}

// Capacity (keys per tile) of the fixed-capacity bins for the view after one with `instances` instances and a longest list of
// `longest` on a grid of `tiles` tiles: the longest list + 25 % where the budget allows it; where it does not, the deep tiles are
// OUTLIERS for the overflow path and the bins are sized for the rest (4 x the mean list, at least 1024, + 25 %).  A budget below
// twice the mean list (or below 64 keys) is no budget for bins at all: 0 = compact mode.  So is a budget that keeps the bins
// below the overflow path's 1024 keys while a list is longer than them (round 6): such bins would overflow on every
// view — filled, the fused launch bailing out, the view binned again compactly — where the compact mode repeats nothing.
uint32_t bins_capacity_after(uint64_t instances, uint64_t longest, uint64_t tiles, uint64_t budget_bytes) {
    const uint64_t mean_list = instances / tiles + 1;
    const uint64_t budget = budget_of(budget_bytes, instances);
    const uint64_t cap = (budget / (8ull * (tiles + 1))) & ~63ull;
    uint64_t want = std::max<uint64_t>(64ull, round64_slack(longest));  // (an empty view keeps the minimum)
    if (want > cap) {
        want = std::min<uint64_t>(cap, round64_slack(std::max<uint64_t>(kHybridMinCap, 4ull * mean_list)));
        if (want < kHybridMinCap && longest > want) return 0u;
    }
    if (want < 64 || cap < 2 * mean_list) return 0u;
    return (uint32_t)std::min<uint64_t>(want, 1u << 20);
}

inline uint64_t tiles_of(const gsr_policy_config* c) { return (uint64_t)c->grid_x * (uint64_t)c->grid_y; }

int form_for(const gsr_policy_config* cfg, int request, int n, uint32_t bin_cap_view, bool skewed) {
    const gsr_agg::Plan pl = gsr_agg::plan(cfg->grid_x, cfg->grid_y, bin_cap_view);
    // Default (request = -1): the aggregating form for scenes of >= 250 k Gaussians where ONE band holds the grid (measured faster
    // on every scene: 1080p, 1440p); on larger grids (4K: two bands) only for SKEWED views — the previous view's longest tile list
    // was several times its mean, i.e. some counter words are hot and their global atomics serialise (dense 4K scene: 2.0 ->
    // 1.1 ms) — because a uniform 4K scene is faster in the direct form (config 5: 0.67 against 0.83 ms).
    const bool agg = request >= 0 ? request != 0
                                  : (n >= gsr_agg::kMinGaussians &&
                                     (pl.n_bands <= cfg->agg_max_bands || (skewed && pl.n_bands <= gsr_agg::kMaxBandsOpen)));
    return gsr_agg::form_code(agg, pl);
}

bool form_is_open(const gsr_policy_config* cfg, int n, uint32_t bin_cap_view) {
    if (n < gsr_agg::kMinGaussians) return false;
    const gsr_agg::Plan pl = gsr_agg::plan(cfg->grid_x, cfg->grid_y, bin_cap_view);
    return pl.n_bands > cfg->agg_max_bands && pl.n_bands <= gsr_agg::kMaxBandsOpen;
}

}  // namespace

extern "C" {

void gsr_policy_config_init(gsr_policy_config* cfg, int32_t width, int32_t height, uint64_t bins_budget_bytes,
                            int32_t preprocess_form) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->grid_x = (width + kTile - 1) / kTile;
    cfg->grid_y = (height + kTile - 1) / kTile;
    cfg->bins_budget_bytes = bins_budget_bytes;
    cfg->preprocess_form = preprocess_form < 0 ? -1 : (preprocess_form != 0 ? 1 : 0);
    cfg->form_tuner = 1;
    cfg->beside_max_tiles = 0xFFFFFFFFu;
    cfg->bwd_split_max_tiles = 256u;
    cfg->agg_max_bands = gsr_agg::kMaxBandsDefault;
}

void gsr_policy_state_init(gsr_policy_state* st) {
    if (!st) return;
    memset(st, 0, sizeof *st);
    st->tuner.form = -1;
}

uint32_t gsr_bins_capacity_after(int64_t n_rendered, int32_t max_tile_instances, int32_t width, int32_t height,
                                 uint64_t bins_budget_bytes, uint32_t current_capacity) {
    if (n_rendered < 0 || max_tile_instances < 0 || width <= 0 || height <= 0) return 0u;
    const uint64_t tiles = (uint64_t)((width + kTile - 1) / kTile) * (uint64_t)((height + kTile - 1) / kTile);
    const uint32_t want = bins_capacity_after((uint64_t)n_rendered, (uint64_t)max_tile_instances, tiles, bins_budget_bytes);
    return want == 0u ? 0u : std::max(want, current_capacity);
}

void gsr_policy_agg_plan(int32_t grid_x, int32_t grid_y, uint32_t max_pos, int32_t* n_bands, int32_t* band_rows,
                         size_t* lds_bytes, int32_t* words16) {
    if (grid_x <= 0 || grid_y <= 0) return;
    const gsr_agg::Plan p = gsr_agg::plan(grid_x, grid_y, max_pos);
    if (n_bands) *n_bands = p.n_bands;
    if (band_rows) *band_rows = p.band_rows;
    if (lds_bytes) *lds_bytes = p.lds;
    if (words16) *words16 = p.w32 ? 1 : 0;
}

int32_t gsr_policy_preprocess_form(const gsr_policy_config* cfg, int32_t form_request, int32_t n, uint32_t bin_cap_view,
                                   int32_t skewed) {
    if (!cfg || n <= 0) return GSR_FORM_DIRECT;
    return form_for(cfg, form_request, n, bin_cap_view, skewed != 0);
}

int32_t gsr_policy_form_is_open(const gsr_policy_config* cfg, int32_t n, uint32_t bin_cap_view) {
    return cfg && form_is_open(cfg, n, bin_cap_view) ? 1 : 0;
}

void gsr_policy_begin_view(const gsr_policy_config* cfg, gsr_policy_state* st, int32_t n, const float* timed_ms,
                           gsr_view_plan* plan) {
    memset(plan, 0, sizeof *plan);
    plan->timed_slot = -1;
    st->views++;
    const uint64_t T = tiles_of(cfg);
    const uint64_t nn = n > 0 ? (uint64_t)n : 1ull;
    // FAST mode: every tile owns a fixed-capacity key bin (capacity = the longest list seen on this handle + 25 %; before the
    // first view an estimate from N / T: ~6 tiles per Gaussian, + 35 %) and preprocess drops the keys straight into them.  Its
    // memory is (T+1) x capacity x 8 B, so it is only used while that stays within the bins budget.
    if (st->bin_cap == 0 && !st->compact_sticky) {
        const uint64_t est = 8ull * nn / T + 64;
        st->bin_cap = (uint32_t)((est < (1u << 20) ? est : (1u << 20)) + 63) & ~63u;
    }
    const uint64_t budget = budget_of(cfg->bins_budget_bytes, (uint64_t)st->last_n_rendered);
    // (the capacity the budget allows: bins smaller than one cache line of keys per tile are not worth having)
    const uint64_t cap_budget = (budget / (8ull * (T + 1))) & ~63ull;
    if (st->bin_cap > cap_budget) st->bin_cap = (uint32_t)cap_budget;
    const bool use_bins = st->bin_cap >= 64u;
    st->bin_cap_view = plan->bin_cap_view = use_bins ? st->bin_cap : 0u;

    // the binning form: the configured request, else by scene and grid — and where that leaves two candidates, the one this
    // handle has measured to be faster
    int request = cfg->preprocess_form;
    if (request < 0 && cfg->form_tuner && form_is_open(cfg, n, plan->bin_cap_view)) {
        gsr_form_tuner& t = st->tuner;
        const bool decided = t.phase == GSR_TUNER_TIMED_VIEWS + 1;
        if (decided && (++t.age > kTunerMaxAge || std::abs(n - t.n_ref) > t.n_ref / 4)) {
            t.phase = 0; t.form = -1; t.age = 0;
            st->tuner_rearms++;
        }
        if (t.phase == GSR_TUNER_TIMED_VIEWS + 1) {
            request = t.form;
        } else if (st->views > 2) {  // (the first views of a handle grow buffers and touch memory for the first time)
            if (t.phase == GSR_TUNER_TIMED_VIEWS) {
                if (timed_ms) {
                    t.ms[0] = std::min(timed_ms[0], timed_ms[2]);
                    t.ms[1] = std::min(timed_ms[1], timed_ms[3]);
                    t.form = t.ms[1] < 0.97f * t.ms[0] ? 1 : 0;  // (a tie stays with the direct form)
                    t.phase = GSR_TUNER_TIMED_VIEWS + 1; t.n_ref = n; t.age = 0;
                    request = t.form;
                    plan->tuner_decided = 1;
                }
            } else {
                plan->timed_slot = t.phase;
                request = t.phase & 1;
                t.phase++;
            }
        }
    }
    plan->skewed = st->last_n_rendered > 0 && (uint64_t)st->last_max_tile * T > 6ull * (uint64_t)st->last_n_rendered;
    plan->form_request = request;
    plan->form = n > 0 ? form_for(cfg, request, n, plan->bin_cap_view, plan->skewed != 0) : GSR_FORM_DIRECT;
    // TIER TILES (lists beyond the fused launch's 1024 instances) are walked by their own launch of single-wave workgroups, which
    // belongs BESIDE the fused launch: when the previous view had tier tiles the fused launch is held until their sorts have run
    const uint64_t prev_tiers = (uint64_t)st->tier_n[0] + st->tier_n[1] + st->tier_n[2];
    plan->hold_fused = prev_tiers > 0 && prev_tiers <= cfg->beside_max_tiles;
    // ... and the mid tiers' sorts need not wait for the host either: grids from the previous view's counts, checked on the device
    const auto guess = [&](uint32_t prev) -> uint32_t {
        if (!prev) return 0u;
        const uint64_t g = (uint64_t)prev + prev / 4 + 16u, T = tiles_of(cfg);
        return (uint32_t)(g < T ? g : T);
    };
    // (lists beyond 8192, if the view has any, are sorted after the read-back as before: their chain needs host-sized scratch)
    const bool spec_sorts = plan->hold_fused && plan->bin_cap_view > 0;
    plan->spec_mid4 = spec_sorts ? guess(st->tier_n[0]) : 0u;
    plan->spec_mid8 = spec_sorts ? guess(st->tier_n[1]) : 0u;
}

void gsr_policy_end_view(const gsr_policy_config* cfg, gsr_policy_state* st, const gsr_view_plan* plan, int64_t n_rendered,
                         uint32_t max_tile_instances, uint32_t n_mid4, uint32_t n_mid8, uint32_t n_big,
                         uint64_t cap_instances, int32_t fused_allowed, gsr_view_outcome* out) {
    memset(out, 0, sizeof *out);
    const uint64_t T = tiles_of(cfg);
    const uint64_t D = n_rendered > 0 ? (uint64_t)n_rendered : 0ull;
    const bool use_bins = plan->bin_cap_view > 0;
    // OVERFLOW TILES: lists longer than the bins' capacity.  Their bins hold the first arrivals only; their complete key lists
    // come from a scatter pass restricted to them.  Bins of fewer than 1024 keys could also cut a list of the fused launch's:
    // such a view is finished in the compact mode.
    const bool overflow = use_bins && max_tile_instances > plan->bin_cap_view;
    const bool hybrid = overflow && plan->bin_cap_view >= kHybridMinCap;
    const bool compact = !use_bins || (overflow && !hybrid);
    out->binning = compact ? GSR_BINNING_COMPACT : (hybrid ? GSR_BINNING_OVERFLOW : GSR_BINNING_BINS);
    // capacity for the NEXT view: grow-only while bins are in use
    uint32_t want = bins_capacity_after(D, max_tile_instances, T, cfg->bins_budget_bytes);
    if (use_bins && want > st->bin_cap) {
        // the bins GROW: by at least a quarter of what is there (within the budget).  A multi-view batch brings a slightly longer
        // list every few views (round 6, 16 views at reduced size: 2112 -> 2176 -> 2368 keys within one densification round —
        // three reallocations of the bins for 3 % each); geometric growth bounds their number by the logarithm of the growth.
        const uint64_t cap = (budget_of(cfg->bins_budget_bytes, D) / (8ull * (T + 1))) & ~63ull;
        const uint64_t grown = ((uint64_t)st->bin_cap + st->bin_cap / 4 + 63) & ~63ull;
        want = (uint32_t)std::max<uint64_t>(want, std::min<uint64_t>(std::min<uint64_t>(grown, cap), 1u << 20));
    }
    if (want == 0u) {
        st->bin_cap = 0; st->compact_sticky = 1;
    } else if (want > st->bin_cap || !use_bins) {
        out->bins_regrown = use_bins && want > st->bin_cap;
        st->bin_cap = want; st->compact_sticky = 0;
    }
    out->bin_cap_next = st->bin_cap;
    const bool spec = fused_allowed != 0 && use_bins;
    const bool held = spec && plan->hold_fused != 0;
    // did (will) the fused launch run?  (the same two comparisons as in the kernel, on the same numbers; overflow views: every
    // list of up to 1024 sat complete in its bin)
    out->fused_done = spec && D <= cap_instances && !(overflow && !hybrid);
    out->long_tiles = (n_mid4 | n_mid8 | n_big) != 0u;
    const uint64_t tiers = (uint64_t)n_mid4 + n_mid8 + n_big;
    out->beside = held && out->fused_done && out->long_tiles && tiers <= cfg->beside_max_tiles;
    // (a held fused launch that this view gives no reason to hold any longer: no tier tiles after all, or too many)
    out->launch_fused_now = held && !out->beside && out->fused_done;
    // what the speculative mid-tier sorts covered: the kernels' own test (tile_sort_runs_kernel), on the same numbers
    const bool spec_sorted = (plan->spec_mid4 | plan->spec_mid8) != 0u && D <= cap_instances && max_tile_instances <= plan->bin_cap_view;
    out->sorted_mid4 = spec_sorted ? (plan->spec_mid4 < n_mid4 ? plan->spec_mid4 : n_mid4) : 0u;
    out->sorted_mid8 = spec_sorted ? (plan->spec_mid8 < n_mid8 ? plan->spec_mid8 : n_mid8) : 0u;
    // history
    if (!use_bins) st->compact_views++;
    else if (overflow && !hybrid) st->compact_fallbacks++;
    else if (hybrid) st->overflow_views++;
    if (out->bins_regrown) st->bins_regrowths++;
    if (spec && !held && D > cap_instances) st->fused_relaunches++;
    if (held) st->held_views++;
    st->last_n_rendered = (int64_t)D;
    st->last_max_tile = max_tile_instances;
    st->tier_n[0] = n_mid4; st->tier_n[1] = n_mid8; st->tier_n[2] = n_big;
}

// As long as tiles with long lists are too few to fill the GPU by themselves (at most bwd_split_max_tiles: one per CU), they are
// taken out of the main launch and walked in list segments.  The cut is a tier boundary of the scan (> 8192, > 4096 or > 1024
// instances): the deepest tiers first, as many as fit the limit.
void gsr_policy_bwd_split(const gsr_policy_config* cfg, uint32_t n_mid4, uint32_t n_mid8, uint32_t n_big, gsr_bwd_split* out) {
    memset(out, 0, sizeof *out);
    const uint32_t cut[3] = {8192u, 4096u, 1024u};  // deepest tier first (GSR_SORT_LDS_CAP, ...)
    const uint32_t have[3] = {n_big, n_mid8, n_mid4};
    uint32_t* take[3] = {&out->n_big, &out->n_mid8, &out->n_mid4};
    uint64_t n = 0;
    out->split_len = 0xFFFFFFFFu;
    for (int t = 0; t < 3 && n + have[t] <= cfg->bwd_split_max_tiles; t++) {
        n += *take[t] = have[t];
        out->split_len = cut[t];
    }
    if (n == 0) out->split_len = 0xFFFFFFFFu;  // nothing to split (or the deepest tier alone is already plentiful)
}

}  // extern "C"
