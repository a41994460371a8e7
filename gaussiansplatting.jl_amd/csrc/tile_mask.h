// Opacity-aware footprint test shared by the binning kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Conservative footprint masks of an instance inside its 16x16 tile.
//   bits 0..15  row mask: bit r is set unless NO pixel of row r of the tile can reach
//               alpha >= 1/255 (render.jl:95), i.e. unless the ellipse {sigma <= ln(255*opacity)}
//               misses the row's pixel centres;
//   bits 16..19 quadrant mask: bit 16 + 2*qy + qx for the 8x8 quadrant (qx, qy), from the same
//               per-row x-intervals tested against the quadrant's columns.
// The composite kernels use them only to skip work; every surviving (pixel, splat) pair still
// runs the exact test, so the slack below never changes a result.
__device__ __forceinline__ uint32_t instance_row_mask(const float4 g0, const float4 g1, int X0, int Y0) {
    const float mx = g0.x, my = g0.y, a = g0.z, b = g0.w, c = g1.x, o = g1.y;
    const float tau = __logf(255.0f * o) + 2e-3f;  // sigma <= tau  <=>  alpha >= 1/255 (with slack)
    if (!(tau >= 0.0f)) return 0u;                  // opacity < 1/255: never blended
    if (!(a > 0.0f)) return 0xFFFFFu;               // degenerate conic: no culling
    // dx = mx - px: whole tile [X0, X0+15], left half [X0, X0+7], right half [X0+8, X0+15].
    // A row's x-interval is centre(dy) +- hw(dy) with centre = -b*dy/a, hw = sqrt(disc)/a,
    // disc = (b^2 - a*c)*dy^2 + 2*a*tau.  One slack for all rows of the instance — 0.01 px + 1e-4 of
    // the largest |centre| + hw any row can have, on top of tau's 2e-3; fp32 rounding of these
    // expressions is ~1e-5 px — is folded into the thresholds.
    const float inv_a = 1.0f / a;
    const float k0 = 2.0f * a * tau, k2 = b * b - a * c;
    const float dy0 = my - (float)Y0;
    const float dymax = fmaxf(fabsf(dy0), fabsf(dy0 - 15.0f));
    const float slack = 0.01f + 1e-4f * (fabsf(b) * inv_a * dymax + __fsqrt_rn(k0) * inv_a);
    const float dx_hi = mx - (float)X0;
    const float t_lo = dx_hi - 15.0f - slack;   // hi >= t_lo  <=> reaches the tile's last column
    const float t_hi = dx_hi + slack;           // lo <= t_hi  <=> reaches the tile's first column
    const float t_left = dx_hi - 7.0f - slack;  // hi >= t_left  (with lo <= t_hi): left half
    const float t_right = dx_hi - 8.0f + slack; // lo <= t_right (with hi >= t_lo): right half
    const float thr = -1e-3f * (k0 + (b * b + fabsf(a * c)) * dymax * dymax);  // numerically on the boundary: keep
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float dy = dy0 - (float)r;
        const float disc = k2 * (dy * dy) + k0;
        if (disc > thr) {
            const float hw = __fsqrt_rn(fmaxf(disc, 0.0f)) * inv_a;
            const float centre = -(b * dy) * inv_a;
            const float lo = centre - hw, hi = centre + hw;
            if (hi >= t_lo && lo <= t_hi) {
                const uint32_t qrow = r < 8 ? 16u : 18u;
                m |= 1u << r;
                if (hi >= t_left) m |= 1u << qrow;
                if (lo <= t_right) m |= 1u << (qrow + 1);
            }
        }
    }
    return m;
}


// Cheap per-tile version for the count / scatter loops: does the ellipse {sigma <= tau}
// reach the box of the tile's pixel centres?  sigma is a convex quadratic, so its minimum
// over the box is 0 if the mean lies inside and otherwise sits on one of the four edges
// (a clamped 1-D parabola each).  Conservative (the box also contains the points between
// pixel centres); tau = ln(255*opacity) + slack is hoisted per Gaussian by the caller.
// The blend test of a (pixel, splat) pair — sigma >= 0 && min(0.99, o·exp(-sigma)) >= 1/255 (render.jl:92-95) — as ONE
// unsigned compare per pair: the predicate is monotone in sigma, non-negative floats order like their bit patterns and any
// negative sigma has the sign bit set, so with  S = the largest sigma for which the reference's own expression passes,
//     bits(sigma) < X,   X = bits(S) + 1   (0 when not even sigma = +0 passes: opacity below 1/255, NaN)
// is the whole test.  S is found ONCE PER GAUSSIAN (preprocess; carried in the geometry record) by bisection over bit
// patterns around ln(255·o), evaluating the reference's expression with a correctly rounded exp (through fp64): the decision
// for every pair is then exactly "fl(o · fl(exp(-sigma))) >= fl(1/255)", not an approximation of it by a rounded logarithm
// or a 1-ulp exp (either flips pairs within an ulp of the boundary against the oracle: 1 scene in 400, then 3 in 1600 of
// tools/fuzz_parity.py had one gradient beyond tolerance).
__device__ __forceinline__ uint32_t blend_threshold_bits(float o) {
    const float amin = 1.0f / 255.0f;
    // (exp through fp64: the correctly rounded fp32 value — the CPU oracle's libm expf is that in 99.6 % of its results)
    auto pass = [&](uint32_t b) { return fminf(0.99f, __fmul_rn(o, (float)exp(-(double)__uint_as_float(b)))) >= amin; };
    if (!pass(0u)) return 0u;
    const float tau = fmaxf(logf(255.0f * o), 0.0f);
    const float w = 4e-7f * fmaxf(tau, 1.0f);  // > the shift a 1-ulp exp and a 1-ulp log can cause, absolute
    uint32_t lo = __float_as_uint(fmaxf(tau - w, 0.0f)), hi = __float_as_uint(tau + w);
    if (!pass(lo)) lo = 0u;
    for (int k = 0; k < 24 && pass(hi); k++) { lo = hi; hi = __float_as_uint(__uint_as_float(hi) + w * (float)(2 << k)); }
    if (pass(hi)) return hi + 1u;  // (unreachable: sigma = tau + 6.7 passes for no opacity <= 1)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pass(mid)) lo = mid; else hi = mid;
    }
    return lo + 1u;
}

__device__ __forceinline__ float footprint_tau(float opacity) { return __logf(255.0f * opacity) + 2e-3f; }

__device__ __forceinline__ bool tile_may_touch(float mx, float my, float a, float b, float c, float tau, int X0,
                                               int Y0) {
    if (!(tau >= 0.0f)) return false;
    if (!(a > 0.0f) || !(c > 0.0f)) return true;
    const float x_lo = mx - (float)(X0 + 15), x_hi = mx - (float)X0;  // dx = mx - px
    const float y_lo = my - (float)(Y0 + 15), y_hi = my - (float)Y0;
    if (x_lo <= 0.0f && x_hi >= 0.0f && y_lo <= 0.0f && y_hi >= 0.0f) return true;
    const float inv_a = 1.0f / a, inv_c = 1.0f / c;
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const float dx = e ? x_hi : x_lo;  // vertical edges: minimise over dy
        float dy = fminf(y_hi, fmaxf(y_lo, -b * dx * inv_c));
        best = fminf(best, b * dx * dy + 0.5f * (a * dx * dx + c * dy * dy));
        const float ey = e ? y_hi : y_lo;  // horizontal edges: minimise over dx
        float ex = fminf(x_hi, fmaxf(x_lo, -b * ey * inv_a));
        best = fminf(best, b * ex * ey + 0.5f * (a * ex * ex + c * ey * ey));
    }
    return best <= tau + 1e-3f * (1.0f + fabsf(best));
}
