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
    const float eps = 0.02f;
    // dx = mx - px: whole tile [X0, X0+15], left half [X0, X0+7], right half [X0+8, X0+15]
    const float dx_hi = mx - (float)X0, dx_lo = dx_hi - 15.0f;
    const float dxl_lo = dx_hi - 7.0f, dxr_hi = dx_hi - 8.0f;
    const float inv_a = 1.0f / a;
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float dy = my - (float)(Y0 + r);
        const float bd = b * dy;
        const float disc = bd * bd - a * (c * dy * dy - 2.0f * tau);
        const uint32_t qrow = r < 8 ? 16u : 18u;
        if (disc >= 0.0f) {
            const float s = __fsqrt_rn(disc);
            const float lo = (-bd - s) * inv_a, hi = (-bd + s) * inv_a;
            const float slack = eps * (1.0f + fabsf(lo) + fabsf(hi));
            const float l = lo - slack, h = hi + slack;
            if (h >= dx_lo && l <= dx_hi) {
                m |= 1u << r;
                if (h >= dxl_lo) m |= 1u << qrow;        // left half: dx in [dx_hi - 7, dx_hi]
                if (l <= dxr_hi) m |= 1u << (qrow + 1);  // right half: dx in [dx_hi - 15, dx_hi - 8]
            }
        } else if (disc > -1e-3f * (bd * bd + fabsf(a * c * dy * dy) + 2.0f * a * tau)) {
            m |= (1u << r) | (3u << qrow);  // numerically on the boundary: keep
        }
    }
    return m;
}


// Cheap per-tile version for the count / scatter loops: does the ellipse {sigma <= tau}
// reach the box of the tile's pixel centres?  sigma is a convex quadratic, so its minimum
// over the box is 0 if the mean lies inside and otherwise sits on one of the four edges
// (a clamped 1-D parabola each).  Conservative (the box also contains the points between
// pixel centres); tau = ln(255*opacity) + slack is hoisted per Gaussian by the caller.
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
