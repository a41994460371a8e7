// Device code shared by the sort kernels (binning.hip) and the fused sort + forward kernel (composite.hip): the emit of
// one sorted instance (record gather -> packed splat stream + footprint masks) and the single-wave register sort of a
// tile's key list.
#pragma once
#include "gsr_kernels.h"
#include "tile_mask.h"

namespace gsr_sort {

// The packed stream entry of one sorted instance (record gather -> three / four float4).
struct InstanceVals { float4 v0, v1, v2, v3; };
// What instance_vals reads from HBM for one instance (split from the arithmetic so that a kernel can have the gathers of
// its next instances in flight while it does something else).
struct InstanceRaw { GsrGeoRec rec; uint32_t bpre; float4 normal; };
template <int CH>
__device__ __forceinline__ InstanceRaw instance_load(uint32_t id, const GsrGeom& geom) {
    InstanceRaw r;
    r.rec = geom.rec[id];  // one 64-byte line per gather
    r.bpre = geom.bpre[id >> 8];
    r.normal = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (CH > 5) r.normal = geom.normal[id];
    return r;
}
// The blend-test threshold X of an instance (tile_mask.h blend_threshold_bits: one unsigned compare, bits(sigma) < X, is
// the reference's whole test) was computed by preprocess and sits in the Gaussian's record.  It travels in the stream's
// depth slot in :rgb mode (which does not use the depth) and in the w component of the fourth plane in :rgbd / :rgbdn
// mode (the plane that carries the normal in :rgbdn; :rgbd gets it for X alone: 16 bytes per instance for one compare
// and, in the backward, exp / alpha only on the lanes that pass).  Forward and backward compare the same sigma against
// the same X: identical contributor sets by construction.
template <int CH>
__device__ __forceinline__ InstanceVals instance_vals_of(const InstanceRaw& r, int X0, int Y0) {
    InstanceVals o;
    const GsrGeoRec& rec = r.rec;
    // the compositing kernels evaluate sigma = b·dx·dy + (a/2)·dx² + (c/2)·dy²: the stream carries the halves
    o.v0 = make_float4(rec.q0.x, rec.q0.y, 0.5f * rec.q0.z, rec.q0.w);
    o.v1 = make_float4(0.5f * rec.q1.x, rec.q1.y, rec.q1.z, rec.q1.w);
    // Gaussian-major slot of this instance: offset of the Gaussian's rect (cumsum of
    // tiles_touched, rasterizer.jl:333-335) + row-major rank of the tile inside the rect (the
    // emit order of duplicate_with_keys!, utils.jl:112).  The backward writes the instance's
    // gradient row there, so a Gaussian's rows are contiguous for the per-Gaussian sum.
    const uint32_t lo = __float_as_uint(rec.q3.x), hi = __float_as_uint(rec.q3.y);
    const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, x1 = hi & 0xFFFFu, y1 = hi >> 16;
    // rank of this tile among the Gaussian's slots: small rects (<= GSR_DENSE_RECT tiles) have one slot per
    // EMITTED tile — the popcount of the record's emitted-tile mask below this tile's bit —, larger ones one per tile
    const uint32_t k = ((uint32_t)(Y0 / GSR_TILE) - y0) * (x1 - x0) + ((uint32_t)(X0 / GSR_TILE) - x0);
    const uint32_t dense = (x1 - x0) * (y1 - y0) <= GSR_DENSE_RECT;
    const uint32_t rank = dense ? (uint32_t)__popc(__float_as_uint(rec.q3.w) & ((1u << (k & 31u)) - 1u)) : k;
    const uint32_t slot = r.bpre + __float_as_uint(rec.q2.w) + rank;
    const uint32_t mask_bits = instance_row_mask(rec.q0, rec.q1, X0, Y0);
    // (:rgb, CH == 3: the blend-test threshold in place of the depth, which only the :rgbd / :rgbdn features use;
    //  otherwise in v3.w)
    const float X = rec.q3.z;  // (bit pattern)
    const float z = CH == 3 ? X : rec.q2.z;
    o.v2 = make_float4(rec.q2.x, __uint_as_float(slot), z, __uint_as_float(mask_bits));
    o.v3 = make_float4(r.normal.x, r.normal.y, r.normal.z, X);
    return o;
}
template <int CH>
__device__ __forceinline__ InstanceVals instance_vals(uint32_t id, int X0, int Y0, const GsrGeom& geom) {
    return instance_vals_of<CH>(instance_load<CH>(id, geom), X0, Y0);
}

// Emit one sorted instance: its id, and its entry of the packed splat stream.
template <int CH>
__device__ __forceinline__ void emit_instance(uint64_t k, uint32_t pos, int X0, int Y0, const GsrGeom& geom,
                                              const GsrStream& stream, uint32_t* __restrict__ values_sorted) {
    const uint32_t id = (uint32_t)k;
    values_sorted[pos] = id;
    const InstanceVals v = instance_vals<CH>(id, X0, Y0, geom);
    stream.s0[pos] = v.v0;
    stream.s1[pos] = v.v1;
    stream.s2[pos] = v.v2;
    if (CH > 3) stream.s3[pos] = v.v3;
}

// ---- the main pass: ONE wave64 per tile, keys in registers ----
// A list of up to 1024 keys (nearly every tile) is sorted by a single wave with KPT = m / 64 keys per lane
// (element e = lane * KPT + r).  Of the bitonic network's stages, those with stride j < KPT are compare-exchanges
// between two registers of one lane; the others exchange with lane ^ (j / KPT) through ds_bpermute — no LDS traffic
// for the keys and not a single workgroup barrier (the 256-thread LDS network paid one per stage: 45 for m = 512).
// Per tile at m = 512: ~1.3 k wave instructions instead of ~3.6 k.  Same total order (unique keys).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int s) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, s), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), s);
    return ((uint64_t)hi << 32) | lo;
}
template <int KPT>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&v)[KPT], uint32_t m, int lane) {
    for (uint32_t k = 2; k <= m; k <<= 1) {
        // stages across lanes: j = s * KPT, partner lane ^ s; all KPT keys of a lane play the same role
        const bool up = ((uint32_t)(lane * KPT) & k) == 0u;  // k >= 2 KPT here
        for (uint32_t j = k >> 1; j >= (uint32_t)KPT; j >>= 1) {
            const int s = (int)(j / KPT);
            const bool keep_min = ((lane & s) == 0) == up;
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const uint64_t p = shfl_xor_u64(v[r], s);
                v[r] = ((v[r] < p) == keep_min) ? v[r] : p;
            }
        }
        // stages inside a lane: j = KPT/2 ... 1 (those not larger than k/2)
#pragma unroll
        for (int jj = KPT >> 1; jj >= 1; jj >>= 1) {
            if ((uint32_t)jj <= (k >> 1)) {
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    if ((r & jj) == 0) {
                        const bool upr = (((uint32_t)(lane * KPT + r)) & k) == 0u;
                        const uint64_t a = v[r], b = v[r | jj];
                        const bool sw = (a > b) == upr;
                        v[r] = sw ? b : a;
                        v[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
}

// One wave sorts the n <= 64 KPT keys of a tile and leaves the ids, in order, in LDS (only the id of a sorted key is
// needed afterwards; through LDS so that the emit is lane-contiguous whatever the number of emitting threads).
template <int KPT>
__device__ __forceinline__ void wave_sort_ids(uint32_t* ids /* LDS [64 KPT] */, uint32_t n, int lane,
                                              const uint64_t* __restrict__ keys) {
    uint64_t v[KPT];
#pragma unroll
    for (int r = 0; r < KPT; r++) {
        const uint32_t e = (uint32_t)(lane * KPT + r);
        v[r] = e < n ? keys[e] : ~0ull;  // padded with +inf to m = 64 KPT
    }
    wave_bitonic_sort<KPT>(v, 64u * KPT, lane);
#pragma unroll
    for (int r = 0; r < KPT; r++) ids[lane * KPT + r] = (uint32_t)v[r];
}
// KPT by list length (n <= 1024)
__device__ __forceinline__ void wave_sort_ids_any(uint32_t* ids /* LDS [1024] */, uint32_t n, int lane,
                                                  const uint64_t* __restrict__ keys) {
    if (n <= 64u) wave_sort_ids<1>(ids, n, lane, keys);
    else if (n <= 128u) wave_sort_ids<2>(ids, n, lane, keys);
    else if (n <= 256u) wave_sort_ids<4>(ids, n, lane, keys);
    else if (n <= 512u) wave_sort_ids<8>(ids, n, lane, keys);
    else wave_sort_ids<16>(ids, n, lane, keys);
}

// ---- lists of up to 1024 RUNS keys: every wave sorts a run of 1024 keys in registers, the runs are merged through LDS ----
// (round 5; used by the tier sort tile_sort_runs_kernel and by the mid-list fused forward sort_composite_fwd_mid_kernel.)
// log2(RUNS) passes in which every thread finds its 16 outputs by merge path (a binary search over two sorted runs) and merges
// them sequentially.  Runs are padded to 1024 with +inf, so every merge is of two full runs; passes stop as soon as one run holds
// every real key.  Keys are unique: the same total order as everywhere.  Ends with a workgroup barrier: buf[0, n) is sorted.
__device__ __forceinline__ uint32_t merge_path_lds(const uint64_t* a, const uint64_t* b, uint32_t len, uint32_t diag) {
    // number of elements taken from `a` among the first `diag` outputs of merge(a[0, len), b[0, len))
    uint32_t lo = diag > len ? diag - len : 0u, hi = diag < len ? diag : len;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < b[diag - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <int RUNS>
__device__ __forceinline__ void sort_runs_lds(uint64_t* buf /* LDS [1024 RUNS] */, uint32_t n, int tid,
                                              const uint64_t* __restrict__ keys) {
    constexpr int NT = 64 * RUNS, CAP = 1024 * RUNS, PER = CAP / NT;  // 16 outputs per thread and pass
    const int lane = tid & 63, wave = tid >> 6;
    {
        uint64_t v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t e = (uint32_t)(wave * 1024 + lane * 16 + r);
            v[r] = e < n ? keys[e] : ~0ull;
        }
        if ((uint32_t)(wave * 1024) < n) wave_bitonic_sort<16>(v, 1024u, lane);  // (a run of +inf only is sorted)
#pragma unroll
        for (int r = 0; r < 16; r++) buf[wave * 1024 + lane * 16 + r] = v[r];
    }
    __syncthreads();
    for (uint32_t L = 1024; L < (uint32_t)CAP && L < n; L <<= 1) {
        const uint32_t o0 = (uint32_t)tid * PER, pair = o0 / (2 * L) * (2 * L);
        const uint64_t* a = buf + pair;
        const uint64_t* b = buf + pair + L;
        uint32_t ia = merge_path_lds(a, b, L, o0 - pair), ib = (o0 - pair) - ia;
        uint64_t out[PER];
        uint64_t ka = ia < L ? a[ia] : ~0ull, kb = ib < L ? b[ib] : ~0ull;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            // (+inf padding compares equal on both sides: taking `a` first keeps ia, ib inside their runs)
            const bool take_a = ib >= L || (ia < L && ka <= kb);
            out[k] = take_a ? ka : kb;
            if (take_a) { ia++; ka = ia < L ? a[ia] : ~0ull; }
            else { ib++; kb = ib < L ? b[ib] : ~0ull; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++) buf[o0 + k] = out[k];
        __syncthreads();
    }
}

template <int CH>
__device__ __forceinline__ void wave_sort_and_emit(uint32_t* ids /* LDS [1024] */, uint32_t n, uint32_t start, int lane,
                                                   int X0, int Y0, const uint64_t* __restrict__ keys,
                                                   const GsrGeom& geom, const GsrStream& stream,
                                                   uint32_t* __restrict__ values_sorted) {
    wave_sort_ids_any(ids, n, lane, keys);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n; i += 64) emit_instance<CH>((uint64_t)ids[i], start + i, X0, Y0, geom, stream, values_sorted);
}

}  // namespace gsr_sort
