// The NU.Adam update and the per-Gaussian part of the trainer tail, shared by the stand-alone tail
// kernels (trainer.hip) and the backward kernel that applies the tail in its epilogue (pergauss.hip):
// ONE definition of every expression, so the fused and the unfused step give the same bits.
// Both translation units are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

struct AdamHyper { float lr_t, beta1, beta2, omb1, omb2, eps; };  // lr_t = lr · sqrt(1-β2^t) / (1-β1^t) (host)

__device__ __forceinline__ float adam_update(float th, float g, float& m, float& v, const AdamHyper& h) {
    m = h.beta1 * m + h.omb1 * g;
    v = h.beta2 * v + h.omb2 * (g * g);
    return th - h.lr_t * m / (sqrtf(v) + h.eps);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }  // NerfUtils.sigmoid

// The six NU.Adam states of one scene (group order of training.jl:415-416) + the activated copies
// the rasterizer consumes.
struct TailState {
    float *points, *p_mu, *p_nu;
    float *dc, *dc_mu, *dc_nu;
    float *rest, *rest_mu, *rest_nu;
    float *opac, *o_mu, *o_nu;
    float *scales, *s_mu, *s_nu;
    float *rots, *r_mu, *r_nu;
    float *shs, *opac_act, *scales_act;
    AdamHyper h_p, h_dc, h_rest, h_o, h_s, h_r;
    int scale_dims;
};

// points (3), opacity logit (1), log-scales (3 or 1), rotations (4) of Gaussian i.  The gradients
// arrive w.r.t. the ACTIVATED opacity / scales (what ∇project produces); σ' = σ(1-σ) and exp' = exp
// are taken from the activated copies the forward of this step used (rasterizer.jl:218-247), which
// are then overwritten with those of the updated parameters.
__device__ __forceinline__ void tail_gauss_apply(const TailState& S, int i, const float vmean[3], float vopac_act,
                                                 const float vscales_act[3], const float vrot[4]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const size_t k = 3 * (size_t)i + c;
        float m = S.p_mu[k], v = S.p_nu[k];
        S.points[k] = adam_update(S.points[k], vmean[c], m, v, S.h_p);
        S.p_mu[k] = m; S.p_nu[k] = v;
    }
    {
        const float a = S.opac_act[i];
        const float g = vopac_act * (a * (1.0f - a));
        float m = S.o_mu[i], v = S.o_nu[i];
        const float t = adam_update(S.opac[i], g, m, v, S.h_o);
        S.opac[i] = t; S.o_mu[i] = m; S.o_nu[i] = v;
        S.opac_act[i] = sigmoidf_(t);
    }
    {
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; c++) g[c] = vscales_act[c] * S.scales_act[3 * (size_t)i + c];
        if (S.scale_dims == 1) {
            float m = S.s_mu[i], v = S.s_nu[i];
            const float t = adam_update(S.scales[i], (g[0] + g[1]) + g[2], m, v, S.h_s);
            S.scales[i] = t; S.s_mu[i] = m; S.s_nu[i] = v;
            const float e = expf(t);
            S.scales_act[3 * (size_t)i] = e; S.scales_act[3 * (size_t)i + 1] = e; S.scales_act[3 * (size_t)i + 2] = e;
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const size_t k = 3 * (size_t)i + c;
                float m = S.s_mu[k], v = S.s_nu[k];
                const float t = adam_update(S.scales[k], g[c], m, v, S.h_s);
                S.scales[k] = t; S.s_mu[k] = m; S.s_nu[k] = v;
                S.scales_act[k] = expf(t);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const size_t k = 4 * (size_t)i + c;
        float m = S.r_mu[k], v = S.r_nu[k];
        S.rots[k] = adam_update(S.rots[k], vrot[c], m, v, S.h_r);
        S.r_mu[k] = m; S.r_nu[k] = v;
    }
}

}  // namespace gsr

// host side: theta/mu/nu in group order points, features_dc, features_rest, opacities, scales, rotations
static inline gsr::TailState gsr_make_tail_state(float* const* theta, float* const* mu, float* const* nu,
                                                 const float* lr_t, float beta1, float beta2, float eps, int scale_dims,
                                                 float* shs, float* opac_act, float* scales_act) {
    gsr::TailState S;
    S.points = theta[0]; S.p_mu = mu[0]; S.p_nu = nu[0];
    S.dc = theta[1]; S.dc_mu = mu[1]; S.dc_nu = nu[1];
    S.rest = theta[2]; S.rest_mu = mu[2]; S.rest_nu = nu[2];
    S.opac = theta[3]; S.o_mu = mu[3]; S.o_nu = nu[3];
    S.scales = theta[4]; S.s_mu = mu[4]; S.s_nu = nu[4];
    S.rots = theta[5]; S.r_mu = mu[5]; S.r_nu = nu[5];
    S.shs = shs; S.opac_act = opac_act; S.scales_act = scales_act;
    gsr::AdamHyper* h[6] = {&S.h_p, &S.h_dc, &S.h_rest, &S.h_o, &S.h_s, &S.h_r};
    for (int g = 0; g < 6; g++) *h[g] = gsr::AdamHyper{lr_t[g], beta1, beta2, 1.0f - beta1, 1.0f - beta2, eps};
    S.scale_dims = scale_dims;
    return S;
}
