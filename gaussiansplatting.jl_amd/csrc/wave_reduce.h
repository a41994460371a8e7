// Transposed multi-value wave64 reduction for gfx950.
//
// Reduces N per-lane values (N <= 16) across the 64 lanes of a wave with ~3N/2 + 6 VALU
// ops instead of 6N: at every butterfly level two values are *paired* — one half of the
// lanes carries on with the first, the other half with the second — so the number of live
// registers halves per level.  Levels, in order:
//   bit 5  v_permlane32_swap (lanes 32-63 <-> 0-31)          2 ops / pair
//   bit 4  v_permlane16_swap (odd rows <-> even rows)         2 ops / pair
//   bit 3  DPP row_ror:8                                      2 v_cndmask + 1 v_add_dpp / pair
//   bit 0  DPP quad_perm [1,0,3,2]                            2 v_cndmask + 1 v_add_dpp / pair
//   bit 1  DPP quad_perm [2,3,0,1]   (plain add, 1 value left)
//   bit 2  DPP row_shr:4             (plain add; no xor-4 DPP exists, so only lanes with
//                                     bit 2 set end up with both halves)
// Afterwards every lane with (lane & 4) != 0 holds the complete 64-lane sum of one of the N
// inputs; which one is given by wave_reduce_index<N>(lane).  The caller stores with one
// ds_write per wave from the lanes wave_reduce_writer(lane) selects.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shr4(float v) {  // lane i <- lane i-4 within the row, 0 for i < 4
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
}

// sum over the lane pair (i, i^32): lanes 0-31 get a's, lanes 32-63 get b's
__device__ __forceinline__ float pair_swap32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over (i, i^16): even rows get a's, odd rows get b's
__device__ __forceinline__ float pair_swap16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct LaneBits {
    bool b0, b3, hi, odd_row;
    __device__ __forceinline__ explicit LaneBits(int lane)
        : b0(lane & 1), b3(lane & 8), hi(lane & 32), odd_row(lane & 16) {}
};

// One pairing level over n live values -> (n+1)/2 live values.  KIND: 0 swap32, 1 swap16, 2 ror8, 3 xor1.
template <int KIND>
__device__ __forceinline__ float pair_level(float a, float b, const LaneBits& L) {
    if (KIND == 0) return pair_swap32(a, b);
    if (KIND == 1) return pair_swap16(a, b);
    const bool bit = KIND == 2 ? L.b3 : L.b0;
    const float send = bit ? a : b, keep = bit ? b : a;
    return keep + (KIND == 2 ? dpp_ror8(send) : dpp_xor1(send));
}
template <int KIND>
__device__ __forceinline__ float single_level(float a, const LaneBits& L) {
    if (KIND == 0) return pair_swap32(a, a);
    if (KIND == 1) return pair_swap16(a, a);
    return a + (KIND == 2 ? dpp_ror8(a) : dpp_xor1(a));
}

template <int N, int KIND>
__device__ __forceinline__ void net_level(float (&v)[16], const LaneBits& L) {
    constexpr int H = N / 2;
#pragma unroll
    for (int p = 0; p < H; p++) v[p] = pair_level<KIND>(v[2 * p], v[2 * p + 1], L);
    if (N & 1) v[H] = single_level<KIND>(v[N - 1], L);
}

// In: v[0..N-1] per-lane partials.  Out: return value = the full wave sum of input
// number wave_reduce_index<N>(lane).
template <int N>
__device__ __forceinline__ float wave_reduce_transposed(float (&v)[16], const LaneBits& L) {
    static_assert(N >= 1 && N <= 16, "at most 16 values per network");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2;
    net_level<N, 0>(v, L);
    net_level<N1, 1>(v, L);
    net_level<N2, 2>(v, L);
    net_level<N3, 3>(v, L);
    float r = v[0];
    r = r + dpp_xor2(r);
    r = r + dpp_shr4(r);
    return r;
}

// Lanes whose result is complete and which are pairwise distinct per input (bit 1 clear, bit 2 set).
__device__ __forceinline__ bool wave_reduce_writer(int lane) { return (lane & 6) == 4; }

// ---- row-then-column reduction of the backward's partials (mode :rgb) ----
// The wave is 4 rows of 16 lanes; lanes l, l^16, l^32, l^48 sit in the same pixel COLUMN, and
// the three geometric moments the gradient row needs are P, dx*P, dx^2*P and U1, dx*U1 with dx a
// function of the column only.  So the six per-lane sums {P, U1, c0, c1, U2, c2} are first
// reduced over the 4 rows (two transposed swap levels: 5 swaps instead of 8 for nine values),
// each row of lanes then expands what it holds with its dx weights into at most 3 outputs, and
// the 16-lane tail of the network finishes.  Row r (= lane >> 4) ends up with:
//   r = 0: {P, dx*P, dx^2*P}   r = 1: {c0, U2, -}   r = 2: {U1, dx*U1, c2}   r = 3: {c1, -, -}
// Output j of a row lands on its lanes with (b3, b0) = (0,0): j = 0, (1,0): j = 1, (0,1): j = 2.
struct RowColConsts {  // loop-invariant per lane
    float e1, e3, k2, k4;  // o1 = b0*(dx*e1) + b1*k2;  o2 = b0*(dx*dx*e3) + b1*k4
    int slot;              // index into the 9-float accumulator row this lane's total goes to, or -1
    __device__ __forceinline__ explicit RowColConsts(int lane) {
        const int r = lane >> 4;
        e1 = (r == 0 || r == 2) ? 1.0f : 0.0f;
        e3 = r == 0 ? 1.0f : 0.0f;
        k2 = r == 1 ? 1.0f : 0.0f;
        k4 = r == 2 ? 1.0f : 0.0f;
        // accumulator row layout (composite.hip): [0..2] rgb, [3] P, [4] dx^2*P, [5] dx*U1, [6] U2, [7] dx*P, [8] U1
        const int j = (lane & 1) ? 2 : ((lane & 8) ? 1 : 0);
        const int table[4][3] = {{3, 7, 4}, {0, 6, -1}, {8, 5, 2}, {1, -1, -1}};
        int t = -1;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
                if (rr == r && jj == j) t = table[rr][jj];
        const bool writer = (lane & 6) == 4 && !((lane & 1) && (lane & 8));  // (b3,b0) = (1,1) duplicates j = 2
        slot = writer ? t : -1;
    }
};

__device__ __forceinline__ float wave_reduce_rowcol_rgb(float P, float U1, float U2, float c0, float c1, float c2,
                                                        float dx, const LaneBits& L, const RowColConsts& K) {
    // rows: pairs (P,U1), (c0,c1), (U2,c2) over lane^32, then pair + single over lane^16
    const float a0 = pair_swap32(P, U1), a1 = pair_swap32(c0, c1), a2 = pair_swap32(U2, c2);
    const float b0 = pair_swap16(a0, a1), b1 = pair_swap16(a2, a2);
    // column weights
    const float o0 = b0;
    const float o1 = b0 * (dx * K.e1) + b1 * K.k2;
    const float o2 = b0 * ((dx * dx) * K.e3) + b1 * K.k4;
    // 16 lanes of the row: pair (o0,o1) by b3, o2 alongside; then pair by b0; then the two plain levels
    const float x0 = pair_level<2>(o0, o1, L), x1 = single_level<2>(o2, L);
    float r = pair_level<3>(x0, x1, L);
    r = r + dpp_xor2(r);
    r = r + dpp_shr4(r);
    return r;
}

// ---- the row stage on the matrix pipe (round 5 experiment, GSR_BWD_MFMA) ----
// v_mfma_f32_16x16x4_f32 computes D[i][j] += sum_k A[i][k] * B[k][j] with lane l holding A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; lane (g, j) gets D[4g + r][j] in register r.  With a per-lane value as B, k runs over the four
// lanes l, l^16, l^32, l^48 of a pixel column: exactly the two swap levels above.  A is a 0/1 selector of OUTPUT rows (the
// same for every k): chaining the six values through one accumulator with six selectors lands value v's column sums in the
// rows its selector names, i.e. on lane row g = i / 4 in register r = i % 4 — the transposed layout the 16-lane tail wants,
// with no swap, no add and no select on the VALU:
//   g = 0: r0 = P   r1 = P  (x dx)  r2 = P (x dx^2)
//   g = 1: r0 = U1  r1 = U1 (x dx)
//   g = 2: r0 = U2  r1 = c0
//   g = 3: r0 = c1  r1 = c2
// The sums are exact-fp32 fmaf chains in k order (0 + v_row0 + v_row1 + v_row2 + v_row3), unselected rows add 0 * v.
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));
struct RowColConstsM {
    float sP, sU1, sU2, s0, s1, s2;  // output-row selectors (A operands): 1.0 where (lane & 15) is one of the value's rows
    bool low;                        // o1 = r1 * (low ? dx : 1): dx on lane rows 0, 1 (a lane-constant SGPR mask, no VGPR)
    int slot;
    __device__ __forceinline__ explicit RowColConstsM(int lane) {
        const int i = lane & 15, g = lane >> 4;
        sP = i <= 2 ? 1.0f : 0.0f;
        sU1 = (i == 4 || i == 5) ? 1.0f : 0.0f;
        sU2 = i == 8 ? 1.0f : 0.0f;
        s0 = i == 9 ? 1.0f : 0.0f;
        s1 = i == 12 ? 1.0f : 0.0f;
        s2 = i == 13 ? 1.0f : 0.0f;
        low = g < 2;
        // accumulator row layout (composite.hip): [0..2] rgb, [3] P, [4] dx^2*P, [5] dx*U1, [6] U2, [7] dx*P, [8] U1
        const int j = (lane & 1) ? 2 : ((lane & 8) ? 1 : 0);
        const int table[4][3] = {{3, 7, 4}, {8, 5, -1}, {6, 0, -1}, {1, 2, -1}};
        int t = -1;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
                if (rr == g && jj == j) t = table[rr][jj];
        const bool writer = (lane & 6) == 4 && !((lane & 1) && (lane & 8));
        slot = writer ? t : -1;
    }
};

__device__ __forceinline__ float wave_reduce_rowcol_rgb_mfma(float P, float U1, float U2, float c0, float c1, float c2,
                                                             float dx, const LaneBits& L, const RowColConstsM& K) {
    mfma_f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.sP, P, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.sU1, U1, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.sU2, U2, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.s0, c0, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.s1, c1, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(K.s2, c2, d, 0, 0, 0);
    const float o0 = d[0];
    const float o1 = d[1] * (K.low ? dx : 1.0f);
    const float o2 = d[2] * (dx * dx);  // rows g > 0 hold an exact 0 there
    const float x0 = pair_level<2>(o0, o1, L), x1 = single_level<2>(o2, L);
    float r = pair_level<3>(x0, x1, L);
    r = r + dpp_xor2(r);
    r = r + dpp_shr4(r);
    return r;
}

// ---- the same for mode :rgbd (seven per-lane sums: P, U1, U2, r, g, b, depth -> ten outputs) ----
// swap32 pairs (P,r) (U1,g) (U2,b) (depth,depth), swap16 pairs (a0,a1) and (a3,a2): six swaps instead of the eight the
// generic ten-value network needs.  Row r (= lane >> 4) ends up with b0 / b1 =
//   r = 0: P / depth (unused)   r = 1: U1 / U2   r = 2: r / depth   r = 3: g / b
// and outputs  r = 0: {P, dx*P, dx^2*P}   r = 1: {U1, dx*U1, U2}   r = 2: {r, depth, -}   r = 3: {g, b, -}.
struct RowColConstsD {
    float e1, e3, k2, k4;
    int slot;
    __device__ __forceinline__ explicit RowColConstsD(int lane) {
        const int r = lane >> 4;
        e1 = r < 2 ? 1.0f : 0.0f;        // o1 = dx * b0 on rows 0, 1
        e3 = r == 0 ? 1.0f : 0.0f;       // o2 = dx^2 * b0 on row 0
        k2 = r >= 2 ? 1.0f : 0.0f;       // o1 = b1 on rows 2, 3
        k4 = r == 1 ? 1.0f : 0.0f;       // o2 = b1 on row 1
        // accumulator row layout (composite.hip, NA = 10): [0..2] rgb, [3] P, [4] dx^2*P, [5] dx*U1, [6] U2, [7] dx*P, [8] U1, [9] depth
        const int j = (lane & 1) ? 2 : ((lane & 8) ? 1 : 0);
        const int table[4][3] = {{3, 7, 4}, {8, 5, 6}, {0, 9, -1}, {1, 2, -1}};
        int t = -1;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
                if (rr == r && jj == j) t = table[rr][jj];
        const bool writer = (lane & 6) == 4 && !((lane & 1) && (lane & 8));
        slot = writer ? t : -1;
    }
};

__device__ __forceinline__ float wave_reduce_rowcol_rgbd(float P, float U1, float U2, float c0, float c1, float c2, float c3,
                                                         float dx, const LaneBits& L, const RowColConstsD& K) {
    const float a0 = pair_swap32(P, c0), a1 = pair_swap32(U1, c1), a2 = pair_swap32(U2, c2), a3 = pair_swap32(c3, c3);
    const float b0 = pair_swap16(a0, a1), b1 = pair_swap16(a3, a2);
    const float o0 = b0;
    const float o1 = b0 * (dx * K.e1) + b1 * K.k2;
    const float o2 = b0 * ((dx * dx) * K.e3) + b1 * K.k4;
    const float x0 = pair_level<2>(o0, o1, L), x1 = single_level<2>(o2, L);
    float r = pair_level<3>(x0, x1, L);
    r = r + dpp_xor2(r);
    r = r + dpp_shr4(r);
    return r;
}

// Which input a lane ends up holding (the same network run on indices).
template <int N>
__device__ __forceinline__ void index_level(int (&idx)[16], bool bit) {
    constexpr int H = N / 2;
#pragma unroll
    for (int p = 0; p < H; p++) idx[p] = bit ? idx[2 * p + 1] : idx[2 * p];
    if (N & 1) idx[H] = idx[N - 1];
}
template <int N>
__device__ __forceinline__ int wave_reduce_index(int lane) {
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2;
    int idx[16];
#pragma unroll
    for (int i = 0; i < 16; i++) idx[i] = i;
    index_level<N>(idx, (lane & 32) != 0);
    index_level<N1>(idx, (lane & 16) != 0);
    index_level<N2>(idx, (lane & 8) != 0);
    index_level<N3>(idx, (lane & 1) != 0);
    return idx[0];
}

}  // namespace gsr
