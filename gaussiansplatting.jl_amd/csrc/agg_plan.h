// The aggregating binning form's LDS plan — shared by the kernels' launcher (pergauss.hip) and the GPU-free policy layer
// (gsr_policy.cpp), so that the form the policy announces is the form the launcher runs.  Plain C++, no HIP.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace gsr_agg {

constexpr int kThreads = 512;            // workgroup of preprocess_kernel's aggregating form
constexpr size_t kLdsMax = 42 * 1024;    // ... its dynamic LDS (the counter words; + 10 KB of static tables): three workgroups per CU
constexpr int kMaxBandsDefault = 1;      // ... and the number of bands the default choice accepts (measured: DESIGN.md §4)
constexpr int kMinGaussians = 250000;    // ... and the scene size from which it is the default (measured, both forms flattened: DESIGN.md §4)
constexpr int kMaxBandsOpen = 8;         // grids of up to this many bands are candidates for the banded form (skew hint / tuner)

// The counter words of ONE BAND of the tile grid must fit kLdsMax (three workgroups per CU).  One band = the whole grid where
// that fits (1080p with 2 x 32-bit words, 1440p with 2 x 16-bit words); else the grid is cut into the fewest equal bands of
// whole tile rows.  2 x 16-bit words need every position handed out to stay below 65 535 - 512 (`max_pos`: the bins'
// capacity, or the longest list in the scatter pass).
struct Plan { bool w32; int n_bands, band_rows; size_t lds; };
inline Plan plan(int grid_x, int grid_y, uint32_t max_pos) {
    Plan p;
    const size_t n_words = ((size_t)grid_x * grid_y + 2) / 2;
    const bool small_pos = max_pos < 0xFFFFu - (uint32_t)kThreads;
    p.w32 = small_pos && n_words * 8 > kLdsMax;  // 64-bit words where the whole grid fits with them (as round 4)
    const size_t wbytes = p.w32 ? 4 : 8;
    const int words_max = (int)(kLdsMax / wbytes);
    int rows = (int)((2 * (size_t)(words_max - 2)) / (size_t)grid_x);  // a band of r rows spans at most r * grid_x / 2 + 2 words
    rows = rows < 1 ? 1 : (rows > grid_y ? grid_y : rows);
    p.n_bands = (grid_y + rows - 1) / rows;
    p.band_rows = (grid_y + p.n_bands - 1) / p.n_bands;  // equal bands
    p.lds = ((size_t)p.band_rows * grid_x / 2 + 2) * wbytes;
    return p;
}
// gsr_stats.preprocess_form of a launch in the aggregating form under plan p
inline int form_code(bool agg, const Plan& p) { return !agg ? 0 : (p.n_bands > 1 ? 3 : (p.w32 ? 2 : 1)); }

}  // namespace gsr_agg
