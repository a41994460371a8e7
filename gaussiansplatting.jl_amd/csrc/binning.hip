// Tile binning: replaces cumsum! + duplicate_with_keys! + sortperm!/_permute! +
// identify_tile_range! (reference: rasterizer.jl:333-378, utils.jl:56-120) with
// fixed-capacity per-tile bins filled by the per-Gaussian kernel and an independent
// per-tile LDS sort.
//
//   preprocess     : each visible Gaussian drops (depth_bits<<32 | id) into the bins of its
//                    tiles (returning atomics on the tile counters; pergauss.hip)
//   tile_scan      : exclusive scan over the T tile counts  -> tile ranges, D, max count
//   tile_sort      : one workgroup per tile sorts its bin in LDS by (depth, id) and writes
//                    the sorted ids plus the packed, sorted splat stream the composite
//                    kernels consume linearly (compact: tile_start[tile] + rank)
//
// The result equals a stable ascending sort of the reference's 64-bit (tile<<32 | depth)
// keys with emit order by Gaussian id: within a tile, ascending depth
// bits, ties by ascending id.  Unlike a global 64-bit radix sort this moves each instance
// through HBM twice (8 B key out, 8 B key in) instead of ~8 passes x 12 B.  The bins are
// sized for the part's 288 GB of HBM: capacity = the longest list seen so far with slack,
// unused slots are never read.
#include <cstdlib>
#include "gsr_kernels.h"
#include "tile_mask.h"
#include "tile_sort_device.h"

namespace {

using gsr_sort::emit_instance;
using gsr_sort::wave_sort_and_emit;

// ---- single-workgroup scans (T = 8160 tiles at 1080p, 32400 at 4K; N/256 Gaussian blocks) ----
// Each thread owns a CONTIGUOUS chunk of ceil(n/1024) elements: serial sum, one block-wide
// exclusive scan of the 1024 chunk sums (wave scan + 16 wave totals), serial write-back — two
// barriers per array instead of four per 1024 elements (this kernel sits before the host sync).
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* wave_sums, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    __syncthreads();  // wave_sums free for reuse
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t ws = wave_sums[w];
        woff += w < wave ? ws : 0u;
        tot += ws;
    }
    total = tot;
    return woff + x - v;
}

// ---- launch order of the compositing workgroups ----
// Tiles in descending order of list length (longest-processing-time-first): the compositing
// kernels hand the heavy tiles out first and the light ones fill the gaps, so the makespan over
// the 1024 SIMDs is ~1 % above the mean load instead of 5-8 % (synthetic) or far more (real
// scenes with a few very deep tiles).  Counting sort over 1024 length classes in one workgroup;
// the order inside a class is arbitrary — tiles are independent, outputs do not depend on it.
__device__ __forceinline__ void tile_order_body(int n_tiles, const uint32_t* __restrict__ tile_count,
                                                uint32_t* __restrict__ order, uint32_t* hist /* [1024] LDS */,
                                                uint32_t* wave_sums /* [16] LDS */) {
    constexpr int NB = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    hist[tid] = 0;
    uint32_t vmax = 0;  // the longest list (this workgroup does not wait for the scan's)
    for (int i = tid; i < n_tiles; i += NB) vmax = max(vmax, tile_count[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor(vmax, off));
    if (lane == 0) wave_sums[wave] = vmax;
    __syncthreads();
    vmax = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) vmax = max(vmax, wave_sums[w]);
    const uint64_t maxc = vmax > 0u ? vmax : 1u;
    __syncthreads();  // wave_sums is reused below
    for (int i = tid; i < n_tiles; i += NB) {
        const uint32_t b = (NB - 1) - (uint32_t)(((uint64_t)tile_count[i] * (NB - 1)) / maxc);
        atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    const uint32_t v = hist[tid];
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += wave_sums[w];
    hist[tid] = wave_off + x - v;
    __syncthreads();
    for (int i = tid; i < n_tiles; i += NB) {
        const uint32_t b = (NB - 1) - (uint32_t)(((uint64_t)tile_count[i] * (NB - 1)) / maxc);
        order[atomicAdd(&hist[b], 1u)] = (uint32_t)i;
    }
}

// Three single-workgroup jobs between the per-Gaussian pass and the sort, all of them chains of a few
// memory round trips and barriers — run as THREE workgroups of one launch instead of one after the other
// (the whole GPU waits for them: 26 us as two launches, scan then order):
//   block 0: exclusive scan of the tile counts -> tile_start, D, longest list, tier lists
//   block 1: exclusive scan of the per-block rect-area sums -> bpre, slot total, visible count
//   block 2: tile launch order (order == NULL: skipped)
// Whichever of blocks 0 / 1 finishes second (ticket in totals[7], left at zero) hands the totals to the host.
__global__ __launch_bounds__(1024) void tile_scan_kernel(int n_tiles, const uint32_t* __restrict__ tile_count,
                                                         uint32_t* __restrict__ tile_start,
                                                         uint32_t* __restrict__ totals, int n_blocks,
                                                         const uint32_t* __restrict__ bsum,
                                                         uint32_t* __restrict__ bpre,
                                                         const uint32_t* __restrict__ bvis,
                                                         uint32_t* __restrict__ big_list,
                                                         uint32_t* __restrict__ host_mirror, uint32_t seq,
                                                         uint32_t* __restrict__ order) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t red[3][16];
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t big_fill, mid8_fill, mid4_fill;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 2) {
        if (order) tile_order_body(n_tiles, tile_count, order, hist, wave_sums);
        return;
    }
    if (blockIdx.x == 0) {
        // tiles: exclusive scan of the counts, longest list, lists beyond the LDS sort
        if (tid == 0) { big_fill = 0; mid8_fill = 0; mid4_fill = 0; }
        __syncthreads();
        constexpr int PER = 8;  // a wave reads 2 KB contiguous per round; rounds of 8192 elements carry a running total
        uint32_t vmax = 0, big = 0, total = 0;
        for (int base = 0; base < n_tiles; base += 1024 * PER) {
            const int lo = base + tid * PER, hi = min(lo + PER, n_tiles);
            uint32_t v[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                v[k] = lo + k < hi ? tile_count[lo + k] : 0u;
                sum += v[k];
                vmax = v[k] > vmax ? v[k] : vmax;
                // rare: tiles whose list exceeds the 1024-key LDS sort are listed per tier (any order: tiles are
                // independent), so the larger sorts launch one workgroup per LISTED tile, sized for their capacity:
                // big_list[0, T) lists > 8192 (merge sort), [T, 2T) lists in (4096, 8192], [2T, 3T) lists in (1024, 4096]
                if (v[k] > GSR_SORT_LDS_CAP) {
                    big += 1u;
                    big_list[atomicAdd(&big_fill, 1u)] = (uint32_t)(lo + k);
                } else if (v[k] > 4096u) {
                    big_list[n_tiles + atomicAdd(&mid8_fill, 1u)] = (uint32_t)(lo + k);
                } else if (v[k] > 1024u) {
                    big_list[2 * n_tiles + atomicAdd(&mid4_fill, 1u)] = (uint32_t)(lo + k);
                }
            }
            uint32_t round_total;
            uint32_t run = total + block_exclusive_scan_1024(sum, wave_sums, round_total);
#pragma unroll
            for (int k = 0; k < PER; k++) {
                if (lo + k < hi) tile_start[lo + k] = run;
                run += v[k];
            }
            total += round_total;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t y = __shfl_xor(vmax, off);
            vmax = y > vmax ? y : vmax;
            big += __shfl_xor(big, off);
        }
        if (lane == 0) { red[0][wave] = vmax; red[1][wave] = big; }
        __syncthreads();
        if (tid == 0) {
            uint32_t m = 0, b = 0;
            for (int w = 0; w < 16; w++) { m = red[0][w] > m ? red[0][w] : m; b += red[1][w]; }
            tile_start[n_tiles] = total;
            totals[0] = total;  // D
            totals[1] = m;      // longest list
            totals[2] = b;      // #tiles over GSR_SORT_LDS_CAP (listed in big_list; they get two global-scratch slabs each)
            totals[3] = mid4_fill;  // #tiles with a list in (1024, 4096]
            totals[6] = mid8_fill;  // #tiles with a list in (4096, 8192]
        }
    } else {
        // Gaussian blocks: per-block sums of tile-rect areas -> bpre (Gaussian-major instance-slot
        // offsets), and the visible count (per-block counts written by preprocess)
        constexpr int PER = 8;
        uint32_t vis = 0, total = 0, any_rect = 0;
        for (int base = 0; base < n_blocks; base += 1024 * PER) {
            const int lo = base + tid * PER, hi = min(lo + PER, n_blocks);
            uint32_t v[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                v[k] = lo + k < hi ? bsum[lo + k] : 0u;
                sum += v[k];
                const uint32_t bv = lo + k < hi ? bvis[lo + k] : 0u;  // visible | (visible with a non-empty rect) << 16, per block
                vis += bv & 0xFFFFu;
                any_rect |= bv >> 16;
            }
            uint32_t round_total;
            uint32_t run = total + block_exclusive_scan_1024(sum, wave_sums, round_total);
#pragma unroll
            for (int k = 0; k < PER; k++) {
                if (lo + k < hi) bpre[lo + k] = run;
                run += v[k];
            }
            total += round_total;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { vis += __shfl_xor(vis, off); any_rect |= __shfl_xor(any_rect, off); }
        if (lane == 0) red[2][wave] = vis | (any_rect ? 0x80000000u : 0u);
        __syncthreads();
        if (tid == 0) {
            uint32_t v = 0, any = 0;
            for (int w = 0; w < 16; w++) { v += red[2][w] & 0x7FFFFFFFu; any |= red[2][w] & 0x80000000u; }
            totals[4] = v | any;  // visible Gaussians; bit 31: one of them has a non-empty tile rect (the reference's D > 0)
            totals[5] = total;  // sum of tile-rect areas = number of Gaussian-major instance slots (gradient rows)
        }
    }
    if (tid == 0) {
        __threadfence();  // this block's totals before its ticket
        if (atomicAdd(&totals[7], 1u) == 1u) {
            totals[7] = 0u;  // ready for the next view
            if (host_mirror) {
                // The host's copy, written straight into its pinned (fine-grained) memory: the seven totals, then the
                // sequence number of this forward with system-scope release — the host spins on that word.  A D2H copy
                // packet + an event record here cost an 8 us bubble on the stream (rocprofv3 kernel trace).
#pragma unroll
                for (int k = 0; k < 7; k++) host_mirror[k] = __hip_atomic_load(&totals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&host_mirror[7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---- per-tile sort ----
// Bitonic network over `m` (power of two) keys held in `buf` (LDS or global scratch).
__device__ __forceinline__ void bitonic_sort(uint64_t* buf, uint32_t m, int tid, int nthreads) {
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (m >> 1); t += nthreads) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t ixj = i | j;
                const uint64_t a = buf[i], b = buf[ixj];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    buf[i] = b;
                    buf[ixj] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <int CH, int NT>
__device__ __forceinline__ void sort_and_emit(uint64_t* buf, uint32_t m, uint32_t n, uint32_t start, int tid,
                                              int X0, int Y0,
                                              const uint64_t* __restrict__ keys, const GsrGeom& geom,
                                              const GsrStream& stream, uint32_t* __restrict__ values_sorted) {
    for (uint32_t i = tid; i < m; i += NT) buf[i] = i < n ? keys[i] : ~0ull;  // keys = this tile's bin / compact segment
    __syncthreads();
    if (m > 1) bitonic_sort(buf, m, tid, NT);
    for (uint32_t i = tid; i < n; i += NT) emit_instance<CH>(buf[i], start + i, X0, Y0, geom, stream, values_sorted);
}

// Where a tile's unsorted keys are.  bin_cap == 0: compact layout, the keys of tile t sit at bins[tile_start[t] ...) (count ->
// scan -> scatter; memory O(D) whatever the skew).  bin_cap > 0: the tile's fixed-capacity bin — unless the list is longer than
// the capacity (round 5: the bin then holds only its first arrivals): the restricted scatter pass has put the complete list at
// overflow[tile_start[t] ...).
__device__ __forceinline__ const uint64_t* tile_keys(const uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                     const uint64_t* __restrict__ overflow, int tile, uint32_t start, uint32_t n) {
    if (bin_cap == 0u) return bins + start;
    return n <= bin_cap ? bins + (size_t)tile * bin_cap : overflow + start;
}

// Tier launches over the tile lists the scan wrote, for the lists the main pass (tile_sort_wave_kernel, below) leaves:
// CAP = 4096 keys with 512 threads (32 KB of LDS), CAP = 8192 with 1024 threads (64 KB); lists beyond 8192 belong to
// tile_sort_big_kernel.  Tiers that are empty (the host knows the counts) are not launched.
// The keys come from tile_keys() above.
template <int CH, int CAP, int NT>
__global__ __launch_bounds__(NT) void tile_sort_kernel(const uint32_t* __restrict__ tile_start,
                                                       const uint32_t* __restrict__ tier_list,
                                                       const uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                       const uint64_t* __restrict__ overflow, int grid_x,
                                                       GsrGeom geom, GsrStream stream,
                                                       uint32_t* __restrict__ values_sorted) {
    __shared__ uint64_t skeys[CAP];
    const int tile = (int)tier_list[blockIdx.x], tid = threadIdx.x;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    const uint32_t n = end - start;
    if (n == 0 || n > (uint32_t)CAP) return;
    const int X0 = (tile % grid_x) * GSR_TILE, Y0 = (tile / grid_x) * GSR_TILE;
    const uint64_t* __restrict__ keys = tile_keys(bins, bin_cap, overflow, tile, start, n);
    uint32_t m = 1;
    while (m < n) m <<= 1;
    sort_and_emit<CH, NT>(skeys, m, n, start, tid, X0, Y0, keys, geom, stream, values_sorted);
}

// Round 5: the tier kernels above run a bitonic NETWORK in LDS — 78 barrier steps for 4096 keys — and a trained-like scene with
// longer lists has hundreds of such tiles (in-plane splat size 12 px at 1 M / 1080p: tile_sort 0.18 ms, as much as the fused
// forward).  Here every wave sorts a run of 1024 keys IN REGISTERS (the main pass's network: no LDS traffic for the keys, no
// barrier) and the RUNS runs are merged through LDS: log2(RUNS) passes in which every thread finds its 16 outputs by merge path
// (a binary search over two sorted runs) and merges them sequentially.  Runs are padded to 1024 with +inf, so every merge is of
// two full runs; passes stop as soon as one run holds every real key.  Keys are unique: the same total order as everywhere.
template <int CH, int RUNS>
__global__ __launch_bounds__(64 * RUNS) void tile_sort_runs_kernel(const uint32_t* __restrict__ tile_start,
                                                                   const uint32_t* __restrict__ tier_list,
                                                                   const uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                                   const uint64_t* __restrict__ overflow, int grid_x,
                                                                   GsrGeom geom, GsrStream stream,
                                                                   uint32_t* __restrict__ values_sorted, uint32_t first,
                                                                   const uint32_t* __restrict__ totals, uint32_t cap_instances,
                                                                   int count_slot) {
    constexpr int NT = 64 * RUNS, CAP = 1024 * RUNS;
    __shared__ uint64_t buf[CAP];
    const uint32_t slot = first + blockIdx.x;
    // totals != NULL: launched behind the scan BEFORE the host has the counts (gsr_launch_tile_sort_mid): the grid is a guess.
    // A view that needs more instances than the buffers hold, or whose bins overflowed, is left alone (the host, which sees
    // the same totals, sorts it after growing / scattering); a workgroup beyond the tier's real count leaves.
    if (totals && (totals[0] > cap_instances || totals[1] > bin_cap || slot >= totals[count_slot])) return;
    const int tile = (int)tier_list[slot], tid = threadIdx.x;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    const uint32_t n = end - start;
    if (n == 0 || n > (uint32_t)CAP) return;
    const int X0 = (tile % grid_x) * GSR_TILE, Y0 = (tile / grid_x) * GSR_TILE;
    const uint64_t* __restrict__ keys = tile_keys(bins, bin_cap, overflow, tile, start, n);
    gsr_sort::sort_runs_lds<RUNS>(buf, n, tid, keys);
    for (uint32_t i = tid; i < n; i += NT) emit_instance<CH>(buf[i], start + i, X0, Y0, geom, stream, values_sorted);
}

// ---- the main pass: ONE wave64 per tile, keys in registers (tile_sort_device.h) ----
template <int CH>
__global__ __launch_bounds__(64) void tile_sort_wave_kernel(const uint32_t* __restrict__ tile_start,
                                                            uint32_t* __restrict__ tile_count,
                                                            const uint64_t* __restrict__ bins, uint32_t bin_cap, int grid_x,
                                                            GsrGeom geom, GsrStream stream,
                                                            uint32_t* __restrict__ values_sorted,
                                                            uint32_t* __restrict__ ranges,
                                                            const uint32_t* __restrict__ totals, uint32_t cap_instances,
                                                            int n_tiles) {
    __shared__ uint32_t ids[1024];
    // Launched BEFORE the host has read the instance count (totals != NULL): the output buffers hold cap_instances
    // instances and the bins bin_cap keys — if this view needs more, every workgroup leaves without touching
    // anything and the host, which sees the same totals, launches the pass again after growing them.
    if (totals && (totals[0] > cap_instances || (totals[1] > bin_cap && bin_cap < 1024u))) return;
    // Workgroup id -> tile, XCD-aware (grid = 8 * ceil(T / 8)): workgroups are dealt round-robin to the 8 XCDs, each
    // with its own L2; XCD x sorts the x-th contiguous eighth of the tiles in raster order, so the record gathers of
    // neighbouring tiles (a Gaussian touches 3.6 on average) meet in one L2.
    const int per = (n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles || (int)(blockIdx.x >> 3) >= per) return;
    const int lane = threadIdx.x;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    const uint32_t n = end - start;
    if (lane == 0) {
        tile_count[tile] = 0u;  // counter ready for the next view
        // identify_tile_range! (utils.jl:56-78): empty tiles keep the (0,0) of the prior fill!
        ranges[2 * tile] = n ? start : 0u;
        ranges[2 * tile + 1] = n ? end : 0u;
    }
    if (n == 0 || n > 1024u) return;  // a longer list: a tier launch's tile
    const int X0 = (tile % grid_x) * GSR_TILE, Y0 = (tile / grid_x) * GSR_TILE;
    const uint64_t* __restrict__ keys = bin_cap ? bins + (size_t)tile * bin_cap : bins + start;
    wave_sort_and_emit<CH>(ids, n, start, lane, X0, Y0, keys, geom, stream, values_sorted);
}

// ---- lists beyond the LDS capacity: chunked LDS sort + merge passes (one 1024-thread workgroup per listed tile) ----
// Phase 1 sorts runs of GSR_SORT_LDS_CAP keys in LDS (the same bitonic network) into slab A; phase 2 merges runs
// pairwise, ping-ponging between the tile's two global slabs: the merge-path split of every 4096-key output block is
// found by a parallel binary search, then each block's two input pieces are staged in LDS (coalesced), every thread
// merges its 4 outputs from LDS and the block is stored coalesced — ceil(log2(n / 8192)) passes of n keys instead of
// the log^2(n)/2 (~100-150) global passes of a bitonic network.  Keys are unique (depth bits << 32 | id), so the
// result is the same total order as every other tier's.
constexpr int BIG_THREADS = 1024, BIG_OUT = 4096;  // outputs per merge block (4 per thread)
__device__ __forceinline__ uint32_t merge_path(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb, uint32_t diag) {
    // number of elements taken from `a` among the first `diag` outputs of merge(a, b)
    uint32_t lo = diag > nb ? diag - nb : 0u, hi = diag < na ? diag : na;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < b[diag - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Round 5: the same three phases as SEPARATE launches over (tile, chunk) / (tile, output block) work items, so that a list of
// tens of thousands of keys is sorted by as many workgroups as it has 8192-key chunks / 4096-key blocks instead of by one (the
// hot-tile scene's 32 k list + its neighbours: 0.49 ms in one workgroup per tile, five workgroups on the whole chip).
//   big_plan_kernel  : prefix sums of the chunk counts and of the block counts over the listed tiles (one workgroup)
//   big_chunk_kernel : sorted runs of GSR_SORT_LDS_CAP keys -> slab 0
//   big_merge_kernel : one merge pass (run length L -> 2L), slab (pass & 1) -> the other; every tile runs every pass — a tile
//                      that is already one run is copied — so that the result of ALL tiles sits in slab (n_pass & 1)
//   big_emit_kernel  : ids + stream entries of one block of the sorted list
// plan[0 .. n_big] = chunk prefix, plan[n_big + 1 .. 2 n_big + 1] = block prefix.
__global__ __launch_bounds__(1024) void big_plan_kernel(const uint32_t* __restrict__ tile_start,
                                                        const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                        uint32_t* __restrict__ plan) {
    __shared__ uint32_t sc[1024], sb[1024];
    const int tid = threadIdx.x;
    uint32_t carry_c = 0, carry_b = 0;
    uint32_t* plan_c = plan, *plan_b = plan + n_big + 1;
    for (uint32_t base = 0; base < n_big; base += 1024) {
        const uint32_t t = base + tid;
        uint32_t n = 0;
        if (t < n_big) { const uint32_t tile = big_list[t]; n = tile_start[tile + 1] - tile_start[tile]; }
        sc[tid] = (n + GSR_SORT_LDS_CAP - 1) / GSR_SORT_LDS_CAP;
        sb[tid] = (n + BIG_OUT - 1) / BIG_OUT;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t vc = tid >= off ? sc[tid - off] : 0u, vb = tid >= off ? sb[tid - off] : 0u;
            __syncthreads();
            sc[tid] += vc; sb[tid] += vb;
            __syncthreads();
        }
        if (t < n_big) { plan_c[t + 1] = carry_c + sc[tid]; plan_b[t + 1] = carry_b + sb[tid]; }
        carry_c += sc[1023]; carry_b += sb[1023];
        __syncthreads();
    }
    if (tid == 0) { plan_c[0] = 0u; plan_b[0] = 0u; }
}

// work item `id` -> (listed tile b, item k inside it); false when id is beyond the plan's total
__device__ __forceinline__ bool big_item(const uint32_t* __restrict__ prefix, uint32_t n_big, uint32_t id, uint32_t& b, uint32_t& k) {
    if (id >= prefix[n_big]) return false;
    uint32_t lo = 0, hi = n_big;  // prefix[lo] <= id < prefix[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= id) lo = mid; else hi = mid;
    }
    b = lo; k = id - prefix[lo];
    return true;
}

// NET = false (default): the chunk is sorted as eight register runs of 1024 keys + three merge-path passes through LDS
// (tile_sort_device.h: sort_runs_lds, 512 threads) — the bitonic network's 91 barrier steps took 77 us per chunk, on the critical
// path of a view whose longest list is tens of thousands of keys; NET = true: that network (GSR_SORT_TIERS_NETWORK=1, A/B runs).
template <bool NET>
__global__ __launch_bounds__(NET ? BIG_THREADS : 512) void big_chunk_kernel(const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                                const uint32_t* __restrict__ plan,
                                                                const uint64_t* __restrict__ bins, uint32_t bin_cap,
                                                                const uint64_t* __restrict__ overflow,
                                                                uint64_t* __restrict__ scratch, size_t slab_stride) {
    constexpr uint32_t NT = NET ? BIG_THREADS : 512;
    __shared__ uint64_t skeys[GSR_SORT_LDS_CAP];
    uint32_t b, c;
    if (!big_item(plan, n_big, blockIdx.x, b, c)) return;
    const int tile = (int)big_list[b], tid = threadIdx.x;
    const uint32_t start = tile_start[tile], n = tile_start[tile + 1] - start;
    const uint64_t* __restrict__ keys = tile_keys(bins, bin_cap, overflow, tile, start, n);
    uint64_t* slab0 = scratch + (size_t)(2 * b) * slab_stride;
    const uint32_t c0 = c * GSR_SORT_LDS_CAP, cn = min((uint32_t)GSR_SORT_LDS_CAP, n - c0);
    if (NET) {
        uint32_t m = 1;
        while (m < cn) m <<= 1;
        for (uint32_t i = tid; i < m; i += NT) skeys[i] = i < cn ? keys[c0 + i] : ~0ull;
        __syncthreads();
        if (m > 1) bitonic_sort(skeys, m, tid, NT);
    } else {
        static_assert(GSR_SORT_LDS_CAP == 8192, "eight runs of 1024 keys");
        gsr_sort::sort_runs_lds<8>(skeys, cn, tid, keys + c0);
    }
    for (uint32_t i = tid; i < cn; i += NT) slab0[c0 + i] = skeys[i];
}

__global__ __launch_bounds__(BIG_THREADS) void big_merge_kernel(const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                                const uint32_t* __restrict__ plan,
                                                                uint64_t* __restrict__ scratch, size_t slab_stride,
                                                                uint32_t L, int cur) {
    __shared__ uint64_t skeys[BIG_OUT];
    __shared__ uint32_t split_a[2];
    uint32_t b, kb;
    if (!big_item(plan + n_big + 1, n_big, blockIdx.x, b, kb)) return;
    const int tile = (int)big_list[b], tid = threadIdx.x;
    const uint32_t n = tile_start[tile + 1] - tile_start[tile];
    const uint64_t* __restrict__ src = scratch + (size_t)(2 * b + cur) * slab_stride;
    uint64_t* __restrict__ dst = scratch + (size_t)(2 * b + (cur ^ 1)) * slab_stride;
    const uint32_t o0 = kb * (uint32_t)BIG_OUT, cnt = min((uint32_t)BIG_OUT, n - o0);
    const uint32_t pair = o0 / (2 * L) * (2 * L);  // a block never straddles a pair: 2L is a multiple of BIG_OUT
    const uint32_t na = min(L, n - pair), nbb = min(L, n - pair - na);
    if (tid < 2) {
        const uint32_t o = o0 + (tid ? cnt : 0u);  // first output of this block / of the next one
        split_a[tid] = o - pair >= na + nbb ? na : merge_path(src + pair, na, src + pair + na, nbb, o - pair);
    }
    __syncthreads();
    const uint32_t a0 = split_a[0], b0 = (o0 - pair) - a0, a1 = split_a[1];
    const uint32_t la = a1 - a0, lb = cnt - la;
    for (uint32_t i = tid; i < cnt; i += BIG_THREADS)
        skeys[i] = i < la ? src[pair + a0 + i] : src[pair + na + b0 + (i - la)];
    __syncthreads();
    // each thread merges 4 consecutive outputs from the staged pieces [0, la) and [la, la + lb)
    const uint32_t d0 = min((uint32_t)tid * 4u, cnt);
    if (d0 < cnt) {
        uint32_t ia = merge_path(skeys, la, skeys + la, lb, d0), ib = d0 - ia;
        uint64_t out[4];
        const uint32_t cnt_t = min(4u, cnt - d0);
        for (uint32_t k = 0; k < cnt_t; k++) {
            const bool take_a = ib >= lb || (ia < la && skeys[ia] < skeys[la + ib]);
            out[k] = take_a ? skeys[ia] : skeys[la + ib];
            ia += take_a ? 1u : 0u;
            ib += take_a ? 0u : 1u;
        }
        for (uint32_t k = 0; k < cnt_t; k++) dst[o0 + d0 + k] = out[k];
    }
}

template <int CH>
__global__ __launch_bounds__(BIG_THREADS) void big_emit_kernel(const uint32_t* __restrict__ tile_start,
                                                               const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                               const uint32_t* __restrict__ plan,
                                                               const uint64_t* __restrict__ scratch, size_t slab_stride, int cur,
                                                               int grid_x, GsrGeom geom, GsrStream stream,
                                                               uint32_t* __restrict__ values_sorted) {
    uint32_t b, kb;
    if (!big_item(plan + n_big + 1, n_big, blockIdx.x, b, kb)) return;
    const int tile = (int)big_list[b];
    const uint32_t start = tile_start[tile], n = tile_start[tile + 1] - start;
    const uint64_t* __restrict__ sorted = scratch + (size_t)(2 * b + cur) * slab_stride;
    const int X0 = (tile % grid_x) * GSR_TILE, Y0 = (tile / grid_x) * GSR_TILE;
    const uint32_t o0 = kb * (uint32_t)BIG_OUT, o1 = min(n, o0 + (uint32_t)BIG_OUT);
    for (uint32_t i = o0 + threadIdx.x; i < o1; i += BIG_THREADS)
        emit_instance<CH>(sorted[i], start + i, X0, Y0, geom, stream, values_sorted);
}

}  // namespace

// The image of a view whose lists are all empty although the reference would have rendered instances (exact-cull mode: every
// instance dropped as invisible): what the compositing writes for a pixel nothing blends into — background, T = 1, no contributor
// (render.jl:118-129 with an empty range).  The reference's all-zero image is for D == 0 only (rasterizer.jl:283,338).
__global__ __launch_bounds__(256) void fill_background_kernel(size_t n_pixels, int channels, GsrBg8 bg, float* __restrict__ image,
                                                              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pixels) return;
    for (int c = 0; c < channels; c++) image[(size_t)channels * p + c] = bg.v[c];
    final_T[p] = 1.0f;
    n_contrib[p] = 0u;
}

__global__ void tile_order_identity_kernel(int n_tiles, uint32_t* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) order[i] = (uint32_t)i;
}

void gsr_launch_tile_scan(hipStream_t s, int n_tiles, const uint32_t* tile_count, uint32_t* tile_start,
                          uint32_t* totals, int n_blocks, const uint32_t* bsum, uint32_t* bpre,
                          const uint32_t* bvis, uint32_t* big_list, uint32_t* host_mirror, uint32_t seq,
                          uint32_t* order) {
    // GSR_TILE_ORDER=raster: row-major launch order (A/B measurements only; outputs are the same)
    static const bool raster = [] { const char* e = getenv("GSR_TILE_ORDER"); return e && e[0] == 'r'; }();
    if (raster)
        hipLaunchKernelGGL(tile_order_identity_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, s, n_tiles, order);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(3), dim3(1024), 0, s, n_tiles, tile_count, tile_start, totals,
                       n_blocks, bsum, bpre, bvis, big_list, host_mirror, seq, raster ? nullptr : order);
}

void gsr_launch_tile_sort(hipStream_t s, int passes, int n_tiles, int grid_x, int channels, const uint32_t* tile_start,
                          uint32_t* tile_count, const uint64_t* bins, uint32_t bin_cap, const uint64_t* overflow_keys,
                          uint32_t n_mid4, uint32_t n_mid8,
                          uint32_t n_big, const uint32_t* tier_lists, uint64_t* big_scratch, size_t slab_stride, GsrGeom geom,
                          GsrStream stream, uint32_t* values_sorted, uint32_t* ranges, const uint32_t* totals,
                          uint32_t cap_instances, uint32_t first4, uint32_t first8) {
    // (first4 / first8: the leading tiles of the two mid tier lists that gsr_launch_tile_sort_mid has already sorted)
    // GSR_SORT_TIERS_NETWORK=1: round 2's LDS bitonic network for the (1024, 8192] tiers (A/B runs); default: register runs + merges
    static const bool net = [] { const char* e = getenv("GSR_SORT_TIERS_NETWORK"); return e && e[0] == '1'; }();
#define LAUNCH_RUNS(CC, RUNSV, GRID, LIST)                                                                        \
    hipLaunchKernelGGL((tile_sort_runs_kernel<CC, RUNSV>), dim3(GRID), dim3(64 * RUNSV), 0, s, tile_start, LIST, bins, \
                       bin_cap, overflow_keys, grid_x, geom, stream, values_sorted, 0u, (const uint32_t*)nullptr, 0u, 0)
#define LAUNCH(CC, CAPV, NTV, GRID, LIST)                                                                        \
    hipLaunchKernelGGL((tile_sort_kernel<CC, CAPV, NTV>), dim3(GRID), dim3(NTV), 0, s, tile_start, LIST, bins,    \
                       bin_cap, overflow_keys, grid_x, geom, stream, values_sorted)
    // lists beyond the LDS sort: plan -> chunk sorts -> merge passes -> emit, one workgroup per chunk / 4096-key block
    // (the plan lives behind the 2 n_big slabs; grids are upper bounds from the longest list, surplus workgroups leave at once)
    uint32_t* const plan = reinterpret_cast<uint32_t*>(big_scratch + (size_t)2 * n_big * slab_stride);
    const uint32_t chunks_ub = n_big * (uint32_t)((slab_stride + GSR_SORT_LDS_CAP - 1) / GSR_SORT_LDS_CAP);
    const uint32_t blocks_ub = n_big * (uint32_t)((slab_stride + BIG_OUT - 1) / BIG_OUT);
#define LAUNCH_BIG(CC)                                                                                            \
    do {                                                                                                          \
        hipLaunchKernelGGL(big_plan_kernel, dim3(1), dim3(1024), 0, s, tile_start, tier_lists, n_big, plan);      \
        if (net) hipLaunchKernelGGL(big_chunk_kernel<true>, dim3(chunks_ub), dim3(BIG_THREADS), 0, s, tile_start, tier_lists, \
                                    n_big, plan, bins, bin_cap, overflow_keys, big_scratch, slab_stride);           \
        else hipLaunchKernelGGL(big_chunk_kernel<false>, dim3(chunks_ub), dim3(512), 0, s, tile_start, tier_lists, n_big, \
                                plan, bins, bin_cap, overflow_keys, big_scratch, slab_stride);                      \
        int cur = 0;                                                                                              \
        for (uint32_t L = GSR_SORT_LDS_CAP; L < slab_stride; L <<= 1, cur ^= 1)                                   \
            hipLaunchKernelGGL(big_merge_kernel, dim3(blocks_ub), dim3(BIG_THREADS), 0, s, tile_start, tier_lists, n_big, \
                               plan, big_scratch, slab_stride, L, cur);                                           \
        hipLaunchKernelGGL((big_emit_kernel<CC>), dim3(blocks_ub), dim3(BIG_THREADS), 0, s, tile_start, tier_lists, n_big, \
                           plan, big_scratch, slab_stride, cur, grid_x, geom, stream, values_sorted);             \
    } while (0)
#define ALL(CC)                                                                                                   \
    if (passes & GSR_SORT_PASS_MAIN)                                                                              \
        hipLaunchKernelGGL((tile_sort_wave_kernel<CC>), dim3(8 * ((n_tiles + 7) / 8)), dim3(64), 0, s, tile_start,  \
                           tile_count, bins, bin_cap, grid_x, geom, stream, values_sorted, ranges, totals,          \
                           cap_instances, n_tiles);                                                                 \
    if (passes & GSR_SORT_PASS_TIERS) {                                                                           \
        if (n_mid4 > first4) { if (net) LAUNCH(CC, 4096, 512, n_mid4 - first4, tier_lists + 2 * (size_t)n_tiles + first4); \
                               else LAUNCH_RUNS(CC, 4, n_mid4 - first4, tier_lists + 2 * (size_t)n_tiles + first4); }        \
        if (n_mid8 > first8) { if (net) LAUNCH(CC, GSR_SORT_LDS_CAP, 1024, n_mid8 - first8, tier_lists + (size_t)n_tiles + first8); \
                               else LAUNCH_RUNS(CC, 8, n_mid8 - first8, tier_lists + (size_t)n_tiles + first8); }            \
        if (n_big > 0) LAUNCH_BIG(CC);                                                                            \
    }
    if (channels > 5) { ALL(8) } else if (channels > 3) { ALL(5) } else { ALL(3) }
#undef ALL
#undef LAUNCH
#undef LAUNCH_RUNS
#undef LAUNCH_BIG
}

// The two mid tiers' sorts queued BEHIND the scan, before the host has read the counts (round 6): with a held fused launch the
// GPU used to idle 22-34 us per view between tile_scan and these sorts — the host's reaction plus a launch
// (profiles/r06/experiments/training_step_idle_time.txt).  The grids are the caller's guesses from the previous view; every
// workgroup checks the scan's totals (tile_sort_runs_kernel).  Tiles the guess did not cover, lists beyond 8192 and views the
// guard turned down are sorted by gsr_launch_tile_sort after the read-back (first4 / first8).  Sorting a tile twice is harmless:
// same keys, same entries.
void gsr_launch_tile_sort_mid(hipStream_t s, int n_tiles, int grid_x, int channels, const uint32_t* tile_start,
                              const uint64_t* bins, uint32_t bin_cap, uint32_t grid4, uint32_t grid8, const uint32_t* tier_lists,
                              GsrGeom geom, GsrStream stream, uint32_t* values_sorted, const uint32_t* totals,
                              uint32_t cap_instances) {
#define SPEC(CC)                                                                                                          \
    do {                                                                                                                  \
        if (grid4) hipLaunchKernelGGL((tile_sort_runs_kernel<CC, 4>), dim3(grid4), dim3(256), 0, s, tile_start,              \
                                      tier_lists + 2 * (size_t)n_tiles, bins, bin_cap, (const uint64_t*)nullptr, grid_x, geom, \
                                      stream, values_sorted, 0u, totals, cap_instances, 3 /* totals[3] = n_mid4 */);          \
        if (grid8) hipLaunchKernelGGL((tile_sort_runs_kernel<CC, 8>), dim3(grid8), dim3(512), 0, s, tile_start,              \
                                      tier_lists + (size_t)n_tiles, bins, bin_cap, (const uint64_t*)nullptr, grid_x, geom,    \
                                      stream, values_sorted, 0u, totals, cap_instances, 6 /* totals[6] = n_mid8 */);          \
    } while (0)
    if (channels > 5) SPEC(8); else if (channels > 3) SPEC(5); else SPEC(3);
#undef SPEC
}

void gsr_launch_fill_background(hipStream_t s, size_t n_pixels, int channels, const float* background, float* image,
                                float* final_T, uint32_t* n_contrib) {
    if (n_pixels == 0) return;
    GsrBg8 bg{};
    for (int c = 0; c < 3 && c < channels; c++) bg.v[c] = background ? background[c] : 0.0f;  // the colour channels (rasterizer.jl:411-414)
    hipLaunchKernelGGL(fill_background_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, n_pixels, channels, bg,
                       image, final_T, n_contrib);
}
