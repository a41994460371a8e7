/*
 * gsr_oracle.c — CPU restatement of the GaussianSplatting.jl rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library, its
 * Python host mirror, the C ABI) may import, call, link or execute this file.
 * It is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / the reported host baseline, never as the thing shipped.
 *
 * PINNING STATUS: the reference is GPU-only Julia (every kernel is
 * `@kernel cpu=false`, test/runtests.jl:9-23 aborts without a GPU) and Julia is
 * absent from the build image, so the reference cannot be executed here and it
 * holds no golden vectors on disk.  This oracle is pinned by the reference's own
 * known-answer and property tests re-expressed in tests/test_oracle_*.py
 * (SURVEY.md §8c K1-K16), by finite differences, and by an independent float64
 * autograd model.  Absolute image values and sort tie order remain
 * "parity unpinned" against a live run of the reference.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Matrices are row-major `m[r][c]` here; the
 * reference builds them column-major (render.jl:329-332).  All arithmetic is
 * IEEE fp32, evaluated left-to-right exactly as the Julia expressions are
 * written; build with -ffp-contract=off so no FMA is introduced.
 *
 * Index convention: 0-based everywhere (the reference is 1-based); Gaussian ids
 * in `values` are 0-based.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* FLOAT64 REPLAY (oracle/Makefile builds this file a second time with -DORC_REAL_DOUBLE into libgsr_oracle_f64.so): the same
 * source text with every `float` a double — the reference's formulas evaluated in float64 on the same inputs.  Used by
 * oracle.py for ONE purpose: the "truth" per-Gaussian backward (orc_project_bwd below), to tell what part of a gradient
 * difference is the fp32 arithmetic of ∇project itself.  For needle-shaped splats (scale ratios of 100 and more, as a trained
 * scene is full of) the reference's fp32 chain conic -> ∇inverse -> ∇perspective_projection -> ∇quat_scale_to_cov loses the
 * thin eigen-direction of the 2x2 covariance to rounding: the last bits of vconic change ∇rotations by 1e-3 relative, whoever
 * computed them.  Only the per-Gaussian functions are meaningful in this build (the binning functions move
 * float BITS around and are not used from it). */
#ifdef ORC_REAL_DOUBLE
#define float double
#define sqrtf sqrt
#define fabsf fabs
#define floorf floor
#define ceilf ceil
#define fmaxf fmax
#define fminf fmin
#define expf exp
#define powf pow
#endif

/* GaussianSplatting.jl:55-56 */
#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE 256

/* utils.jl:33-48 */
static const float SH0 = 0.28209479177387814f;
static const float SH1 = 0.4886025119029199f;
static const float SH2C1 = 1.0925484305920792f;
static const float SH2C2 = -1.0925484305920792f;
static const float SH2C3 = 0.31539156525252005f;
static const float SH2C4 = -1.0925484305920792f;
static const float SH2C5 = 0.5462742152960396f;
static const float SH3C1 = -0.5900435899266435f;
static const float SH3C2 = 2.890611442640554f;
static const float SH3C3 = -0.4570457994644658f;
static const float SH3C4 = 0.3731763325901154f;
static const float SH3C5 = -0.4570457994644658f;
static const float SH3C6 = 1.445305721320277f;
static const float SH3C7 = -0.5900435899266435f;

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* small fixed-size linear algebra (StaticArrays semantics: plain sums,       */
/* left-to-right, no FMA)                                                     */
/* ------------------------------------------------------------------------- */
typedef struct { float m[3][3]; } m33;
typedef struct { float m[2][2]; } m22;
typedef struct { float m[2][3]; } m23;

static m33 m33_mul(m33 a, m33 b) {
    m33 o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return o;
}
static m33 m33_t(m33 a) {
    m33 o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o.m[i][j] = a.m[j][i];
    return o;
}
static m33 m33_add(m33 a, m33 b) {
    m33 o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o.m[i][j] = a.m[i][j] + b.m[i][j];
    return o;
}
static m33 m33_zero(void) { m33 o; memset(&o, 0, sizeof o); return o; }
static m22 m22_mul(m22 a, m22 b) {
    m22 o;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) o.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j];
    return o;
}
/* column-major 9 floats (Julia SMatrix memory order) -> row-major m33 */
static m33 m33_from_colmajor(const float *p) {
    m33 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) o.m[r][c] = p[c * 3 + r];
    return o;
}
static void m33_to_colmajor(m33 a, float *p) {
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) p[c * 3 + r] = a.m[r][c];
}

/* ------------------------------------------------------------------------- */
/* math helpers of render.jl:288-420 and projection.jl:259-393                */
/* ------------------------------------------------------------------------- */

/* render.jl:322-333 `unnorm_quat2rot`; q = (w,x,y,z); normalize = inv(norm)*q */
static m33 unnorm_quat2rot(const float q_in[4]) {
    float n2 = q_in[0] * q_in[0] + q_in[1] * q_in[1] + q_in[2] * q_in[2] + q_in[3] * q_in[3];
    float inv = 1.0f / sqrtf(n2);
    float w = q_in[0] * inv, x = q_in[1] * inv, y = q_in[2] * inv, z = q_in[3] * inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    m33 R;
    R.m[0][0] = 1.0f - 2.0f * (y2 + z2); R.m[1][0] = 2.0f * (xy + wz); R.m[2][0] = 2.0f * (xz - wy);
    R.m[0][1] = 2.0f * (xy - wz); R.m[1][1] = 1.0f - 2.0f * (x2 + z2); R.m[2][1] = 2.0f * (yz + wx);
    R.m[0][2] = 2.0f * (xz + wy); R.m[1][2] = 2.0f * (yz - wx); R.m[2][2] = 1.0f - 2.0f * (x2 + y2);
    return R;
}

/* render.jl:335-366 `∇unnorm_quat2rot` */
static void grad_unnorm_quat2rot(const float q_in[4], m33 vR, float vq[4]) {
    float n2 = q_in[0] * q_in[0] + q_in[1] * q_in[1] + q_in[2] * q_in[2] + q_in[3] * q_in[3];
    float inv_norm = 1.0f / sqrtf(n2);
    float w = q_in[0] * inv_norm, x = q_in[1] * inv_norm, y = q_in[2] * inv_norm, z = q_in[3] * inv_norm;
#define V(i, j) vR.m[(i) - 1][(j) - 1]
    float vqn[4];
    vqn[0] = 2.0f * (x * (V(3, 2) - V(2, 3)) + y * (V(1, 3) - V(3, 1)) + z * (V(2, 1) - V(1, 2)));
    vqn[1] = 2.0f * (-2.0f * x * (V(2, 2) + V(3, 3)) + y * (V(2, 1) + V(1, 2)) + z * (V(3, 1) + V(1, 3)) +
                     w * (V(3, 2) - V(2, 3)));
    vqn[2] = 2.0f * (x * (V(2, 1) + V(1, 2)) - 2.0f * y * (V(1, 1) + V(3, 3)) + z * (V(3, 2) + V(2, 3)) +
                     w * (V(1, 3) - V(3, 1)));
    vqn[3] = 2.0f * (x * (V(3, 1) + V(1, 3)) + y * (V(3, 2) + V(2, 3)) - 2.0f * z * (V(1, 1) + V(2, 2)) +
                     w * (V(2, 1) - V(1, 2)));
#undef V
    float qn[4] = {w, x, y, z};
    float d = vqn[0] * qn[0] + vqn[1] * qn[1] + vqn[2] * qn[2] + vqn[3] * qn[3];
    for (int k = 0; k < 4; k++) vq[k] = (vqn[k] - d * qn[k]) * inv_norm;
}

/* render.jl:291-294 `quat_scale_to_cov(R, scale)`: M = R*diag(s); Σ = M*M' */
static m33 quat_scale_to_cov(m33 R, const float s[3]) {
    m33 M;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M.m[r][c] = R.m[r][c] * s[c];
    return m33_mul(M, m33_t(M));
}

/* render.jl:302-320 `∇quat_scale_to_cov` */
static void grad_quat_scale_to_cov(const float q[4], const float s[3], m33 R, m33 vSigma, m33 vR_extra,
                                   float vq[4], float vscale[3]) {
    m33 M, S = m33_zero();
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M.m[r][c] = R.m[r][c] * s[c];
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    m33 vM = m33_mul(m33_add(vSigma, m33_t(vSigma)), M);
    m33 vR = m33_add(m33_mul(vM, S), vR_extra);
    grad_unnorm_quat2rot(q, vR, vq);
    for (int c = 0; c < 3; c++)
        vscale[c] = R.m[0][c] * vM.m[0][c] + R.m[1][c] * vM.m[1][c] + R.m[2][c] * vM.m[2][c];
}

/* projection.jl:355-361 */
static void pos_world_to_cam(m33 R, const float t[3], const float p[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = (R.m[r][0] * p[0] + R.m[r][1] * p[1] + R.m[r][2] * p[2]) + t[r];
}
/* projection.jl:363-373 */
static void grad_pos_world_to_cam(m33 R, const float p[3], const float v[3], m33 *vR, float vt[3], float vp[3]) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) vR->m[r][c] = v[r] * p[c];
    for (int r = 0; r < 3; r++) vt[r] = v[r];
    for (int c = 0; c < 3; c++) vp[c] = R.m[0][c] * v[0] + R.m[1][c] * v[1] + R.m[2][c] * v[2];
}
/* projection.jl:375-380 */
static m33 covar_world_to_cam(m33 R, m33 Sigma) { return m33_mul(m33_mul(R, Sigma), m33_t(R)); }
/* projection.jl:382-393 */
static void grad_covar_world_to_cam(m33 R, m33 Sigma, m33 vSc, m33 vR_in, m33 *vR, m33 *vSigma) {
    m33 a = m33_mul(m33_mul(vSc, R), m33_t(Sigma));
    m33 b = m33_mul(m33_mul(m33_t(vSc), R), Sigma);
    *vR = m33_add(m33_add(vR_in, a), b);
    *vSigma = m33_mul(m33_mul(m33_t(R), vSc), R);
}

typedef struct {
    float focal[2];
    int res[2];
    float principal[2]; /* normalised [0,1] */
} intr_t;

static void persp_common(const float mean[3], const intr_t *K, float lim[2], float lim_neg[2], float txy[2],
                         m23 *J, float *rz_out) {
    float pp[2], stf[2];
    for (int k = 0; k < 2; k++) {
        float tan_fov = (0.5f * (float)K->res[k]) / K->focal[k];
        stf[k] = 0.3f * tan_fov;
        pp[k] = K->principal[k] * (float)K->res[k];
    }
    float rz = 1.0f / mean[2];
    float rz2 = rz * rz;
    for (int k = 0; k < 2; k++) {
        lim[k] = ((float)K->res[k] - pp[k]) / K->focal[k] + stf[k];
        lim_neg[k] = pp[k] / K->focal[k] + stf[k];
        float v = mean[k] * rz;
        float c = fmaxf(-lim_neg[k], v);
        c = fminf(lim[k], c);
        txy[k] = mean[2] * c;
    }
    J->m[0][0] = K->focal[0] * rz; J->m[1][0] = 0.0f;
    J->m[0][1] = 0.0f;             J->m[1][1] = K->focal[1] * rz;
    J->m[0][2] = -K->focal[0] * txy[0] * rz2;
    J->m[1][2] = -K->focal[1] * txy[1] * rz2;
    *rz_out = rz;
}

/* projection.jl:259-287 `perspective_projection` */
static void perspective_projection(const float mean[3], m33 Sigma, const intr_t *K, m22 *S2, float mean2d[2]) {
    float lim[2], lim_neg[2], txy[2], rz;
    m23 J;
    persp_common(mean, K, lim, lim_neg, txy, &J, &rz);
    for (int k = 0; k < 2; k++) {
        float pp = K->principal[k] * (float)K->res[k];
        mean2d[k] = rz * K->focal[k] * mean[k] + pp;
    }
    /* Σ_2D = (J*Σ)*J' */
    float JS[2][3];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            JS[i][j] = J.m[i][0] * Sigma.m[0][j] + J.m[i][1] * Sigma.m[1][j] + J.m[i][2] * Sigma.m[2][j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            S2->m[i][j] = JS[i][0] * J.m[j][0] + JS[i][1] * J.m[j][1] + JS[i][2] * J.m[j][2];
}

/* projection.jl:289-353 `∇perspective_projection` */
static void grad_perspective_projection(const float mean[3], m33 Sigma, const intr_t *K, m22 vS2,
                                        const float vmean2d[2], m33 *vSigma, float vmean[3]) {
    float lim[2], lim_neg[2], txy[2], rz;
    m23 J;
    persp_common(mean, K, lim, lim_neg, txy, &J, &rz);
    float rz2 = rz * rz, rz3 = rz2 * rz;
    const float *f = K->focal;
    /* vΣ = (J' * vΣ2) * J  (3x2 * 2x2 * 2x3) */
    float JtV[3][2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) JtV[i][j] = J.m[0][i] * vS2.m[0][j] + J.m[1][i] * vS2.m[1][j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) vSigma->m[i][j] = JtV[i][0] * J.m[0][j] + JtV[i][1] * J.m[1][j];
    /* vJ = (vΣ2 * J) * Σ' + (vΣ2' * J) * Σ */
    float A[2][3], B[2][3], vJ[2][3];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            A[i][j] = vS2.m[i][0] * J.m[0][j] + vS2.m[i][1] * J.m[1][j];
            B[i][j] = vS2.m[0][i] * J.m[0][j] + vS2.m[1][i] * J.m[1][j];
        }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            float a = A[i][0] * Sigma.m[j][0] + A[i][1] * Sigma.m[j][1] + A[i][2] * Sigma.m[j][2];
            float b = B[i][0] * Sigma.m[0][j] + B[i][1] * Sigma.m[1][j] + B[i][2] * Sigma.m[2][j];
            vJ[i][j] = a + b;
        }
    float vx = f[0] * rz * vmean2d[0];
    float vy = f[1] * rz * vmean2d[1];
    float vz = -rz2 * (f[0] * mean[0] * vmean2d[0] + f[1] * mean[1] * vmean2d[1]);
    float rx = mean[0] * rz, ry = mean[1] * rz;
    if (-lim_neg[0] <= rx && rx <= lim[0]) vx += -f[0] * rz2 * vJ[0][2];
    else vz += -f[0] * rz3 * vJ[0][2] * txy[0];
    if (-lim_neg[1] <= ry && ry <= lim[1]) vy += -f[1] * rz2 * vJ[1][2];
    else vz += -f[1] * rz3 * vJ[1][2] * txy[1];
    vz += -f[0] * rz2 * vJ[0][0] - f[1] * rz2 * vJ[1][1] + 2.0f * f[0] * txy[0] * rz3 * vJ[0][2] +
          2.0f * f[1] * txy[1] * rz3 * vJ[1][2];
    vmean[0] = vx; vmean[1] = vy; vmean[2] = vz;
}

/* render.jl:387-396 `add_blur` */
static void add_blur(m22 *S, float eps, float *det_blur, float *compensation) {
    float det_orig = S->m[0][0] * S->m[1][1] - S->m[0][1] * S->m[1][0];
    S->m[0][0] = S->m[0][0] + eps;
    S->m[1][1] = S->m[1][1] + eps;
    *det_blur = S->m[0][0] * S->m[1][1] - S->m[0][1] * S->m[1][0];
    *compensation = sqrtf(fmaxf(0.0f, det_orig / *det_blur));
}
/* render.jl:368-381 `inverse` (the `det ≈ 0f0` branch fires only for det == 0) */
static float inverse2(m22 x, m22 *inv) {
    float det = x.m[0][0] * x.m[1][1] - x.m[0][1] * x.m[1][0];
    if (det == 0.0f) { memset(inv, 0, sizeof *inv); return det; }
    float det_inv = 1.0f / det;
    float tmp = -x.m[0][1] * det_inv;
    inv->m[0][0] = x.m[1][1] * det_inv; inv->m[1][0] = tmp;
    inv->m[0][1] = tmp;                 inv->m[1][1] = x.m[0][0] * det_inv;
    return det;
}
/* render.jl:383-385 `∇inverse`: -x*vx*x */
static m22 grad_inverse(m22 x, m22 vx) {
    m22 nx;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) nx.m[i][j] = -x.m[i][j];
    return m22_mul(m22_mul(nx, vx), x);
}
/* render.jl:398-413 `∇add_blur` (compensation path; unused by rasterize, kept for K10) */
static m22 grad_add_blur(float comp, float vcomp, m22 Sb, float eps) {
    float det = Sb.m[0][0] * Sb.m[1][1] - Sb.m[0][1] * Sb.m[1][0];
    float vs = 0.5f * vcomp / (comp + 1e-6f);
    float ct = 1.0f - comp * comp;
    m22 o;
    o.m[0][0] = vs * (ct * Sb.m[0][0] - eps * det);
    o.m[1][0] = vs * ct * Sb.m[1][0];
    o.m[0][1] = vs * ct * Sb.m[0][1];
    o.m[1][1] = vs * (ct * Sb.m[1][1] - eps * det);
    return o;
}
/* render.jl:415-420 */
static float max_eigval_2D(m22 S, float det) {
    float mid = 0.5f * (S.m[0][0] + S.m[1][1]);
    return mid + sqrtf(fmaxf(0.1f, mid * mid - det));
}
/* spherical_harmonics.jl:174-181 `∇normalize` */
static void grad_normalize(const float d[3], const float v[3], float out[3]) {
    float s2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    float inv_s = 1.0f / sqrtf(s2 * s2 * s2);
    out[0] = ((s2 - d[0] * d[0]) * v[0] - d[1] * d[0] * v[1] - d[2] * d[0] * v[2]) * inv_s;
    out[1] = (-d[0] * d[1] * v[0] + (s2 - d[1] * d[1]) * v[1] - d[2] * d[1] * v[2]) * inv_s;
    out[2] = (-d[0] * d[2] * v[0] - d[1] * d[2] * v[1] + (s2 - d[2] * d[2]) * v[2]) * inv_s;
}
/* projection.jl:14-27 `gaussian_normal` -> n_cam, k (0-based), sign */
static void gaussian_normal(m33 Rw, m33 Rg, const float s[3], const float mc[3], float n[3], int *k_out,
                            float *sign_out) {
    int k = (s[0] <= s[1] && s[0] <= s[2]) ? 0 : (s[1] <= s[2]) ? 1 : 2;
    float ax[3] = {Rg.m[0][k], Rg.m[1][k], Rg.m[2][k]};
    float nc[3];
    for (int r = 0; r < 3; r++) nc[r] = Rw.m[r][0] * ax[0] + Rw.m[r][1] * ax[1] + Rw.m[r][2] * ax[2];
    float d = nc[0] * mc[0] + nc[1] * mc[1] + nc[2] * mc[2];
    float sign = d > 0.0f ? -1.0f : 1.0f;
    for (int r = 0; r < 3; r++) n[r] = sign * nc[r];
    *k_out = k; *sign_out = sign;
}

/* utils.jl:14-29 `get_rect` (float ceil-div exactly as `gpu_cld`) */
static void get_rect(const float px[2], int radius, const int grid[2], int rmin[2], int rmax[2]) {
    const int block[2] = {BLOCK_X, BLOCK_Y};
    for (int k = 0; k < 2; k++) {
        float lo = floorf((px[k] - (float)radius) / (float)block[k]);
        float hi_arg = (px[k] + (float)radius) + (float)block[k] - 1.0f;
        float hi = floorf(hi_arg / (float)block[k]);
        int ilo = (int)lo, ihi = (int)hi;
        rmin[k] = ilo < 0 ? 0 : (ilo > grid[k] ? grid[k] : ilo);
        rmax[k] = ihi < 0 ? 0 : (ihi > grid[k] ? grid[k] : ihi);
    }
}

/* ------------------------------------------------------------------------- */
/* helper exports for the unit tests K1, K3-K12 (float32 in/out, col-major 3x3)*/
/* ------------------------------------------------------------------------- */
ORC_API void orc_unnorm_quat2rot(const float *q, float *R_cm) { m33_to_colmajor(unnorm_quat2rot(q), R_cm); }
ORC_API void orc_grad_unnorm_quat2rot(const float *q, const float *vR_cm, float *vq) {
    grad_unnorm_quat2rot(q, m33_from_colmajor(vR_cm), vq);
}
ORC_API void orc_pos_world_to_cam(const float *R_cm, const float *t, const float *p, float *out) {
    pos_world_to_cam(m33_from_colmajor(R_cm), t, p, out);
}
ORC_API void orc_grad_pos_world_to_cam(const float *R_cm, const float *p, const float *v, float *vR_cm, float *vt,
                                       float *vp) {
    m33 vR;
    grad_pos_world_to_cam(m33_from_colmajor(R_cm), p, v, &vR, vt, vp);
    m33_to_colmajor(vR, vR_cm);
}
ORC_API void orc_covar_world_to_cam(const float *R_cm, const float *S_cm, float *out_cm) {
    m33_to_colmajor(covar_world_to_cam(m33_from_colmajor(R_cm), m33_from_colmajor(S_cm)), out_cm);
}
ORC_API void orc_grad_covar_world_to_cam(const float *R_cm, const float *S_cm, const float *vSc_cm,
                                         const float *vRin_cm, float *vR_cm, float *vS_cm) {
    m33 vR, vS;
    grad_covar_world_to_cam(m33_from_colmajor(R_cm), m33_from_colmajor(S_cm), m33_from_colmajor(vSc_cm),
                            m33_from_colmajor(vRin_cm), &vR, &vS);
    m33_to_colmajor(vR, vR_cm); m33_to_colmajor(vS, vS_cm);
}
static intr_t mk_intr(const float *focal, const int *res, const float *principal) {
    intr_t K;
    K.focal[0] = focal[0]; K.focal[1] = focal[1];
    K.res[0] = res[0]; K.res[1] = res[1];
    K.principal[0] = principal[0]; K.principal[1] = principal[1];
    return K;
}
/* 2x2 col-major: [m00, m10, m01, m11] */
static m22 m22_from_cm(const float *p) { m22 o; o.m[0][0] = p[0]; o.m[1][0] = p[1]; o.m[0][1] = p[2]; o.m[1][1] = p[3]; return o; }
static void m22_to_cm(m22 a, float *p) { p[0] = a.m[0][0]; p[1] = a.m[1][0]; p[2] = a.m[0][1]; p[3] = a.m[1][1]; }
ORC_API void orc_perspective_projection(const float *mean, const float *S_cm, const float *focal, const int *res,
                                        const float *principal, float *S2_cm, float *mean2d) {
    intr_t K = mk_intr(focal, res, principal);
    m22 S2;
    perspective_projection(mean, m33_from_colmajor(S_cm), &K, &S2, mean2d);
    m22_to_cm(S2, S2_cm);
}
ORC_API void orc_grad_perspective_projection(const float *mean, const float *S_cm, const float *focal,
                                             const int *res, const float *principal, const float *vS2_cm,
                                             const float *vmean2d, float *vS_cm, float *vmean) {
    intr_t K = mk_intr(focal, res, principal);
    m33 vS;
    grad_perspective_projection(mean, m33_from_colmajor(S_cm), &K, m22_from_cm(vS2_cm), vmean2d, &vS, vmean);
    m33_to_colmajor(vS, vS_cm);
}
ORC_API void orc_quat_scale_to_cov(const float *q, const float *s, float *S_cm) {
    m33_to_colmajor(quat_scale_to_cov(unnorm_quat2rot(q), s), S_cm);
}
ORC_API void orc_grad_quat_scale_to_cov(const float *q, const float *s, const float *vS_cm, const float *vRextra_cm,
                                        float *vq, float *vscale) {
    grad_quat_scale_to_cov(q, s, unnorm_quat2rot(q), m33_from_colmajor(vS_cm), m33_from_colmajor(vRextra_cm), vq,
                           vscale);
}
ORC_API float orc_inverse2(const float *x_cm, float *inv_cm) {
    m22 inv; float det = inverse2(m22_from_cm(x_cm), &inv); m22_to_cm(inv, inv_cm); return det;
}
ORC_API void orc_grad_inverse2(const float *x_cm, const float *vx_cm, float *out_cm) {
    m22_to_cm(grad_inverse(m22_from_cm(x_cm), m22_from_cm(vx_cm)), out_cm);
}
ORC_API void orc_add_blur(const float *S_cm, float eps, float *Sb_cm, float *det_blur, float *comp) {
    m22 S = m22_from_cm(S_cm); add_blur(&S, eps, det_blur, comp); m22_to_cm(S, Sb_cm);
}
ORC_API void orc_grad_add_blur(float comp, float vcomp, const float *conic_cm, float eps, float *out_cm) {
    m22_to_cm(grad_add_blur(comp, vcomp, m22_from_cm(conic_cm), eps), out_cm);
}
ORC_API void orc_grad_normalize(const float *d, const float *v, float *out) { grad_normalize(d, v, out); }
ORC_API void orc_gaussian_normal(const float *Rw_cm, const float *q, const float *s, const float *mc, float *n,
                                 int *k, float *sign) {
    gaussian_normal(m33_from_colmajor(Rw_cm), unnorm_quat2rot(q), s, mc, n, k, sign);
}
ORC_API void orc_get_rect(const float *px, int radius, const int *grid, int *rmin, int *rmax) {
    get_rect(px, radius, grid, rmin, rmax);
}

/* ------------------------------------------------------------------------- */
/* A.2 project! (projection.jl:39-130)                                        */
/* ------------------------------------------------------------------------- */
typedef struct {
    float R[9];  /* column-major world->camera rotation */
    float t[3];
    float focal[2];
    float principal[2]; /* normalised */
    float camera_center[3];
    int width, height;
    float near_plane, far_plane;
    int radius_clip;
    float blur_eps;
} orc_camera;

ORC_API void orc_project(int n, const float *means, const float *scales, const float *rots, const orc_camera *cam,
                         float *depths, int32_t *radii, float *means2d, float *conics, float *normals /*nullable*/) {
    m33 R = m33_from_colmajor(cam->R);
    int res[2] = {cam->width, cam->height};
    intr_t K = mk_intr(cam->focal, res, cam->principal);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        float mc[3];
        pos_world_to_cam(R, cam->t, means + 3 * i, mc);
        if (!(cam->near_plane < mc[2] && mc[2] < cam->far_plane)) { radii[i] = 0; continue; }
        m33 Rg = unnorm_quat2rot(rots + 4 * i);
        m33 Sigma = quat_scale_to_cov(Rg, scales + 3 * i);
        m33 Sc = covar_world_to_cam(R, Sigma);
        m22 S2; float m2[2];
        perspective_projection(mc, Sc, &K, &S2, m2);
        float det, comp;
        add_blur(&S2, cam->blur_eps, &det, &comp);
        if (!(det > 0.0f)) { radii[i] = 0; continue; }
        m22 inv; inverse2(S2, &inv);
        float lam = max_eigval_2D(S2, det);
        int radius = (int)ceilf(3.0f * sqrtf(lam));
        if (radius <= cam->radius_clip) { radii[i] = 0; continue; }
        if ((m2[0] + (float)radius) <= 0.0f || (m2[0] - (float)radius) >= (float)res[0] ||
            (m2[1] + (float)radius) <= 0.0f || (m2[1] - (float)radius) >= (float)res[1]) {
            radii[i] = 0; continue;
        }
        radii[i] = radius;
        means2d[2 * i] = m2[0]; means2d[2 * i + 1] = m2[1];
        depths[i] = mc[2];
        conics[3 * i] = inv.m[0][0]; conics[3 * i + 1] = inv.m[1][0]; conics[3 * i + 2] = inv.m[1][1];
        if (normals) {
            int k; float sg;
            gaussian_normal(R, Rg, scales + 3 * i, mc, normals + 3 * i, &k, &sg);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* A.3 SH colour (spherical_harmonics.jl:1-18, 41-74)                          */
/* shs layout: shs[ch + 3*k + 3*K*i], K = coefficients per Gaussian            */
/* ------------------------------------------------------------------------- */
static void sh_basis(int degree, const float d[3], float b[16]) {
    float x = d[0], y = d[1], z = d[2];
    b[0] = SH0;
    if (degree > 0) {
        b[1] = -SH1 * y; b[2] = SH1 * z; b[3] = -SH1 * x;
        if (degree > 1) {
            float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
            b[4] = SH2C1 * xy; b[5] = SH2C2 * yz; b[6] = SH2C3 * (2.0f * z2 - x2 - y2);
            b[7] = SH2C4 * xz; b[8] = SH2C5 * (x2 - y2);
            if (degree > 2) {
                b[9] = SH3C1 * y * (3.0f * x2 - y2);
                b[10] = SH3C2 * xy * z;
                b[11] = SH3C3 * y * (4.0f * z2 - x2 - y2);
                b[12] = SH3C4 * z * (2.0f * z2 - 3.0f * x2 - 3.0f * y2);
                b[13] = SH3C5 * x * (4.0f * z2 - x2 - y2);
                b[14] = SH3C6 * z * (x2 - y2);
                b[15] = SH3C7 * x * (x2 - 3.0f * y2);
            }
        }
    }
}

ORC_API void orc_sh_forward(int n, int K, int degree, const int32_t *radii, const float *means,
                            const float *cam_center, const float *shs, float *rgbs, uint8_t *clamped) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        if (!(radii[i] > 0)) continue;
        const float *sh = shs + (size_t)3 * K * i;
        float d[3] = {means[3 * i] - cam_center[0], means[3 * i + 1] - cam_center[1], means[3 * i + 2] - cam_center[2]};
        float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] *= inv; d[1] *= inv; d[2] *= inv;
        float b[16];
        sh_basis(degree, d, b);
        int nb = (degree + 1) * (degree + 1);
        for (int c = 0; c < 3; c++) {
            /* res = SH0*sh[0]; res = res - SH1*y*sh1 + SH1*z*sh2 - SH1*x*sh3; ... (left to right) */
            float res = b[0] * sh[c];
            for (int k = 1; k < nb; k++) res = res + b[k] * sh[3 * k + c];
            res = res + 0.5f + 1.1920929e-7f;
            clamped[3 * i + c] = res < 0.0f;
            rgbs[3 * i + c] = fmaxf(0.0f, res);
        }
    }
}

/* A.11 ∇SH (spherical_harmonics.jl:20-38, 76-181); vmeans is accumulated (+=) */
ORC_API void orc_sh_backward(int n, int K, int degree, const float *means, const float *cam_center,
                             const float *shs, const uint8_t *clamped, const float *vrgbs, float *vshs,
                             float *vmeans) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float *sh = shs + (size_t)3 * K * i;
        float *vsh = vshs + (size_t)3 * K * i;
        float d0[3] = {means[3 * i] - cam_center[0], means[3 * i + 1] - cam_center[1], means[3 * i + 2] - cam_center[2]};
        float inv = 1.0f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
        float x = d0[0] * inv, y = d0[1] * inv, z = d0[2] * inv;
        float dir[3] = {x, y, z};
        float vc[3];
        for (int c = 0; c < 3; c++) vc[c] = vrgbs[3 * i + c] * (1.0f - (float)clamped[3 * i + c]);
        float b[16];
        sh_basis(degree, dir, b);
        int nb = (degree + 1) * (degree + 1);
        for (int k = 0; k < nb; k++)
            for (int c = 0; c < 3; c++) vsh[3 * k + c] = b[k] * vc[c];
        float dcx[3] = {0, 0, 0}, dcy[3] = {0, 0, 0}, dcz[3] = {0, 0, 0};
#define S(k, c) sh[3 * (k) + (c)]
        if (degree > 0) {
            for (int c = 0; c < 3; c++) {
                dcx[c] = -SH1 * S(3, c); dcy[c] = -SH1 * S(1, c); dcz[c] = SH1 * S(2, c);
            }
            if (degree > 1) {
                float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
                for (int c = 0; c < 3; c++) {
                    dcx[c] = dcx[c] + SH2C1 * y * S(4, c) + SH2C3 * 2.0f * -x * S(6, c) + SH2C4 * z * S(7, c) +
                             SH2C5 * 2.0f * x * S(8, c);
                    dcy[c] = dcy[c] + SH2C1 * x * S(4, c) + SH2C2 * z * S(5, c) + SH2C3 * 2.0f * -y * S(6, c) +
                             SH2C5 * 2.0f * -y * S(8, c);
                    dcz[c] = dcz[c] + SH2C2 * y * S(5, c) + SH2C3 * 4.0f * z * S(6, c) + SH2C4 * x * S(7, c);
                }
                if (degree > 2) {
                    for (int c = 0; c < 3; c++) {
                        dcx[c] = dcx[c] + SH3C1 * S(9, c) * 3.0f * 2.0f * xy + SH3C2 * S(10, c) * yz +
                                 SH3C3 * S(11, c) * -2.0f * xy + SH3C4 * S(12, c) * -3.0f * 2.0f * xz +
                                 SH3C5 * S(13, c) * (-3.0f * x2 + 4.0f * z2 - y2) + SH3C6 * S(14, c) * 2.0f * xz +
                                 SH3C7 * S(15, c) * 3.0f * (x2 - y2);
                        dcy[c] = dcy[c] + SH3C1 * S(9, c) * 3.0f * (x2 - y2) + SH3C2 * S(10, c) * xz +
                                 SH3C3 * S(11, c) * (-3.0f * y2 + 4.0f * z2 - x2) +
                                 SH3C4 * S(12, c) * -3.0f * 2.0f * yz + SH3C5 * S(13, c) * -2.0f * xy +
                                 SH3C6 * S(14, c) * -2.0f * yz + SH3C7 * S(15, c) * -3.0f * 2.0f * xy;
                        dcz[c] = dcz[c] + SH3C2 * S(10, c) * xy + SH3C3 * S(11, c) * 4.0f * 2.0f * yz +
                                 SH3C4 * S(12, c) * 3.0f * (2.0f * z2 - x2 - y2) +
                                 SH3C5 * S(13, c) * 4.0f * 2.0f * xz + SH3C6 * S(14, c) * (x2 - y2);
                    }
                }
            }
        }
#undef S
        float vdir[3];
        vdir[0] = dcx[0] * vc[0] + dcx[1] * vc[1] + dcx[2] * vc[2];
        vdir[1] = dcy[0] * vc[0] + dcy[1] * vc[1] + dcy[2] * vc[2];
        vdir[2] = dcz[0] * vc[0] + dcz[1] * vc[1] + dcz[2] * vc[2];
        float vm[3];
        grad_normalize(d0, vdir, vm);
        for (int c = 0; c < 3; c++) vmeans[3 * i + c] += vm[c];
    }
}

/* ------------------------------------------------------------------------- */
/* A.4-A.7 binning (utils.jl:56-142, rasterizer.jl:325-378)                    */
/* ------------------------------------------------------------------------- */
ORC_API void orc_count_tiles(int n, const float *means2d, const int32_t *radii, const int *grid,
                             int32_t *tiles_touched) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        if (!(radii[i] > 0)) { tiles_touched[i] = 0; continue; }
        int rmin[2], rmax[2];
        get_rect(means2d + 2 * i, radii[i], grid, rmin, rmax);
        tiles_touched[i] = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
    }
}
/* rasterizer.jl:333-337: inclusive scan; returns n_rendered */
ORC_API int64_t orc_cumsum(int n, const int32_t *in, int32_t *out) {
    int32_t acc = 0;
    for (int i = 0; i < n; i++) { acc += in[i]; out[i] = acc; }
    return n > 0 ? (int64_t)acc : 0;
}
/* utils.jl:85-120; values are 0-based ids here */
ORC_API void orc_duplicate_with_keys(int n, const float *means2d, const float *depths, const int32_t *offsets_incl,
                                     const int32_t *radii, const int *grid, uint64_t *keys, uint32_t *values) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < n; i++) {
        int radius = radii[i];
        if (!(radius > 0)) continue;
        int rmin[2], rmax[2];
        get_rect(means2d + 2 * i, radius, grid, rmin, rmax);
        uint32_t dbits; memcpy(&dbits, depths + i, 4);
        int64_t off = i == 0 ? 0 : offsets_incl[i - 1];
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)y * (uint64_t)grid[0] + (uint64_t)x;
                key <<= 32; key |= dbits;
                keys[off] = key; values[off] = (uint32_t)i; off++;
            }
    }
}
/* rasterizer.jl:357-372 sortperm! + _permute! x2.  The reference's tie order is
 * library-defined (AcceleratedKernels / CUDA.jl); we fix STABLE (ascending
 * Gaussian id), SURVEY.md A.6.  Bottom-up merge sort on (key, index). */
ORC_API void orc_sort_pairs(int64_t d, const uint64_t *keys_in, const uint32_t *values_in, uint64_t *keys_out,
                            uint32_t *values_out) {
    if (d <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * d), *kb = (uint64_t *)malloc(sizeof(uint64_t) * d);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * d), *vb = (uint32_t *)malloc(sizeof(uint32_t) * d);
    memcpy(ka, keys_in, sizeof(uint64_t) * d); memcpy(va, values_in, sizeof(uint32_t) * d);
    for (int64_t w = 1; w < d; w *= 2) {
#pragma omp parallel for schedule(static)
        for (int64_t lo = 0; lo < d; lo += 2 * w) {
            int64_t mid = lo + w < d ? lo + w : d, hi = lo + 2 * w < d ? lo + 2 * w : d;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                if (ka[b] < ka[a]) { kb[o] = ka[b]; vb[o] = va[b]; b++; }
                else { kb[o] = ka[a]; vb[o] = va[a]; a++; }
                o++;
            }
            while (a < mid) { kb[o] = ka[a]; vb[o] = va[a]; a++; o++; }
            while (b < hi) { kb[o] = ka[b]; vb[o] = va[b]; b++; o++; }
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * d); memcpy(values_out, va, sizeof(uint32_t) * d);
    free(ka); free(kb); free(va); free(vb);
}
/* utils.jl:56-78; ranges[2*tile+{0,1}] = [start, end) 0-based; caller zero-fills (rasterizer.jl:375) */
ORC_API void orc_identify_tile_range(int64_t d, const uint64_t *keys, uint32_t *ranges) {
    for (int64_t i = 0; i < d; i++) {
        uint32_t tile = (uint32_t)(keys[i] >> 32);
        if (i == 0) ranges[2 * tile] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (tile != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * tile] = (uint32_t)i; }
        }
        if (i == d - 1) ranges[2 * tile + 1] = (uint32_t)d;
    }
}

/* ------------------------------------------------------------------------- */
/* A.8 render! (render.jl:1-130).  features: (C,N); image: (C,W,H)             */
/* ------------------------------------------------------------------------- */
ORC_API void orc_render(int W, int H, int C, const uint32_t *values, const float *means2d, const float *opac,
                        const float *conics, const float *features, const uint32_t *ranges, const float *background,
                        float *image, uint32_t *n_contrib, float *accum_alpha, uint8_t *covis /*nullable*/,
                        float *uncert /*nullable*/) {
    int gx_n = (W + BLOCK_X - 1) / BLOCK_X, gy_n = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int gy = 0; gy < gy_n; gy++)
        for (int gx = 0; gx < gx_n; gx++) {
            int tile = gy * gx_n + gx;
            int64_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = gx * BLOCK_X + lx, py = gy * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    float T = 1.0f, color[8] = {0}, unc = 0.0f;
                    uint32_t contributor = 0, last = 0;
                    for (int64_t p = r0; p < r1; p++) {
                        contributor++;
                        uint32_t id = values[p];
                        float dx = means2d[2 * id] - (float)px, dy = means2d[2 * id + 1] - (float)py;
                        float a = conics[3 * id], b = conics[3 * id + 1], c = conics[3 * id + 2];
                        float sigma = b * dx * dy + 0.5f * (a * (dx * dx) + c * (dy * dy));
                        if (sigma < 0.0f) continue;
                        float alpha = fminf(0.99f, opac[id] * expf(-sigma));
                        if (alpha < (1.0f / 255.0f)) continue;
                        float Tn = T * (1.0f - alpha);
                        if (Tn < 1e-4f) break;
                        for (int ch = 0; ch < C; ch++) color[ch] += features[(size_t)C * id + ch] * alpha * T;
                        if (uncert) unc += alpha * T;
                        if (covis && T > 0.5f) covis[id] = 1;
                        T = Tn; last = contributor;
                    }
                    size_t pi = (size_t)px + (size_t)W * py;
                    accum_alpha[pi] = T; n_contrib[pi] = last;
                    for (int ch = 0; ch < C; ch++) image[(size_t)C * pi + ch] = color[ch] + T * background[ch];
                    if (uncert) uncert[pi] = unc;
                }
        }
}

/* A.9 ∇render! (render.jl:132-286).  `deterministic`:
 *   0  the reference's own form: parallel tiles, float atomics on the outputs (render.jl:242,275-282; summation
 *      order varies run to run) — the "like-for-like" run and the timed CPU baseline;
 *   1  serial tile loop, DOUBLE accumulators ("truth" gradients; bit-reproducible);
 *   2  parallel tile loop, atomic adds on the same DOUBLE accumulators: the truth gradients at a cost that allows the
 *      large configs (5 M Gaussians @ 4K).  Every per-(pixel, splat) term is the same float value as in mode 1; only
 *      the order of the double additions varies, i.e. results agree with mode 1 to ~1e-16 relative before the final
 *      rounding to float (tested: identical floats but for rare last-bit ties).
 * Outputs must be zero-filled by caller. */
ORC_API void orc_render_bwd(int W, int H, int C, int n, const float *vpixels, const uint32_t *n_contrib,
                            const float *accum_alpha, const uint32_t *values, const float *means2d, const float *opac,
                            const float *conics, const float *features, const uint32_t *ranges,
                            const float *background, float *vfeatures, float *vopac, float *vconics,
                            float *vmeans2d, int deterministic) {
    int gx_n = (W + BLOCK_X - 1) / BLOCK_X, gy_n = (H + BLOCK_Y - 1) / BLOCK_Y;
    double *acc = NULL;
    int S = C + 6;
    if (deterministic) acc = (double *)calloc((size_t)n * S, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) collapse(2) if (deterministic != 1)
    for (int gy = 0; gy < gy_n; gy++)
        for (int gx = 0; gx < gx_n; gx++) {
            int tile = gy * gx_n + gx;
            int64_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            int64_t to_do = r1 - r0;
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = gx * BLOCK_X + lx, py = gy * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    size_t pi = (size_t)px + (size_t)W * py;
                    float T_final = accum_alpha[pi], T = T_final;
                    int64_t contributor = to_do, last_contributor = n_contrib[pi];
                    float accum_rec[8] = {0}, last_color[8] = {0}, last_alpha = 0.0f;
                    const float *vp = vpixels + (size_t)C * pi;
                    float bg_dot = 0.0f;
                    for (int ch = 0; ch < C; ch++) bg_dot += background[ch] * vp[ch];
                    for (int64_t k = 0; k < to_do; k++) {
                        contributor--;
                        if (contributor >= last_contributor) continue;
                        uint32_t id = values[r1 - 1 - k];
                        float dx = means2d[2 * id] - (float)px, dy = means2d[2 * id + 1] - (float)py;
                        float o = opac[id];
                        float a = conics[3 * id], b = conics[3 * id + 1], c = conics[3 * id + 2];
                        float sigma = b * dx * dy + 0.5f * (a * (dx * dx) + c * (dy * dy));
                        if (sigma < 0.0f) continue;
                        float G = expf(-sigma);
                        float alpha = fminf(0.99f, o * G);
                        if (alpha < (1.0f / 255.0f)) continue;
                        T = T / (1.0f - alpha);
                        float fac = alpha * T;
                        float valpha = 0.0f;
                        float vf[8];
                        for (int ch = 0; ch < C; ch++) {
                            vf[ch] = fac * vp[ch];
                            float col = features[(size_t)C * id + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.0f - last_alpha) * accum_rec[ch];
                            last_color[ch] = col;
                            valpha += (col - accum_rec[ch]) * vp[ch];
                        }
                        valpha *= T;
                        valpha += (-T_final / (1.0f - alpha)) * bg_dot;
                        last_alpha = alpha;
                        float vsigma = -o * G * valpha;
                        float vc0 = 0.5f * vsigma * (dx * dx), vc1 = 0.5f * vsigma * dx * dy,
                              vc2 = 0.5f * vsigma * (dy * dy);
                        float vx = vsigma * (a * dx + b * dy), vy = vsigma * (b * dx + c * dy);
                        float vo = G * valpha;
                        if (deterministic == 1) {
                            double *A = acc + (size_t)S * id;
                            for (int ch = 0; ch < C; ch++) A[ch] += vf[ch];
                            A[C] += vo; A[C + 1] += vc0; A[C + 2] += vc1; A[C + 3] += vc2; A[C + 4] += vx; A[C + 5] += vy;
                        } else if (deterministic == 2) {
                            double *A = acc + (size_t)S * id;
                            const double term[6] = {vo, vc0, vc1, vc2, vx, vy};
                            for (int ch = 0; ch < C; ch++) {
#pragma omp atomic
                                A[ch] += (double)vf[ch];
                            }
                            for (int q = 0; q < 6; q++) {
#pragma omp atomic
                                A[C + q] += term[q];
                            }
                        } else {
                            for (int ch = 0; ch < C; ch++) {
#pragma omp atomic
                                vfeatures[(size_t)C * id + ch] += vf[ch];
                            }
#pragma omp atomic
                            vopac[id] += vo;
#pragma omp atomic
                            vconics[3 * id] += vc0;
#pragma omp atomic
                            vconics[3 * id + 1] += vc1;
#pragma omp atomic
                            vconics[3 * id + 2] += vc2;
#pragma omp atomic
                            vmeans2d[2 * id] += vx;
#pragma omp atomic
                            vmeans2d[2 * id + 1] += vy;
                        }
                    }
                }
        }
    if (deterministic) {
        for (int i = 0; i < n; i++) {
            double *A = acc + (size_t)S * i;
            for (int ch = 0; ch < C; ch++) vfeatures[(size_t)C * i + ch] = (float)A[ch];
            vopac[i] = (float)A[C];
            vconics[3 * i] = (float)A[C + 1]; vconics[3 * i + 1] = (float)A[C + 2]; vconics[3 * i + 2] = (float)A[C + 3];
            vmeans2d[2 * i] = (float)A[C + 4]; vmeans2d[2 * i + 1] = (float)A[C + 5];
        }
        free(acc);
    }
}

/* ------------------------------------------------------------------------- */
/* A.10 ∇project! (projection.jl:132-257).  Outputs zero-filled by caller.     */
/* vdepths / vnormals / vR_out / vt_out nullable.                              */
/* ------------------------------------------------------------------------- */
ORC_API void orc_project_bwd(int n, const float *vmeans2d, const float *vconics, const float *vdepths,
                             const float *vnormals, const float *conics, const int32_t *radii, const float *means,
                             const float *scales, const float *rots, const orc_camera *cam, float *vmeans,
                             float *vscales, float *vrots, float *vR_out /*9, col-major*/, float *vt_out) {
    m33 R = m33_from_colmajor(cam->R);
    int res[2] = {cam->width, cam->height};
    intr_t K = mk_intr(cam->focal, res, cam->principal);
    double vRacc[9] = {0}, vtacc[3] = {0};
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        if (!(radii[i] > 0)) continue;
        m22 Ci, vCi;
        Ci.m[0][0] = conics[3 * i]; Ci.m[1][0] = conics[3 * i + 1]; Ci.m[0][1] = conics[3 * i + 1]; Ci.m[1][1] = conics[3 * i + 2];
        vCi.m[0][0] = vconics[3 * i]; vCi.m[1][0] = vconics[3 * i + 1]; vCi.m[0][1] = vconics[3 * i + 1]; vCi.m[1][1] = vconics[3 * i + 2];
        float mc[3];
        pos_world_to_cam(R, cam->t, means + 3 * i, mc);
        m33 Rg = unnorm_quat2rot(rots + 4 * i);
        m33 Sigma = quat_scale_to_cov(Rg, scales + 3 * i);
        m33 Sc = covar_world_to_cam(R, Sigma);
#ifdef ORC_REAL_DOUBLE
        {   /* float64 replay: the conic is re-derived from the raw inputs — the forward's fp32-rounded conic cannot hold the
             * thin eigenvalue of a needle's 2x2 covariance (projection.jl:96-103 restated) */
            m22 S2r; float m2r[2], det_r, comp_r;
            perspective_projection(mc, Sc, &K, &S2r, m2r);
            add_blur(&S2r, cam->blur_eps, &det_r, &comp_r);
            if (det_r > 0.0) inverse2(S2r, &Ci);
        }
#endif
        m22 vS2 = grad_inverse(Ci, vCi);
        m33 vSc; float vmc[3];
        grad_perspective_projection(mc, Sc, &K, vS2, vmeans2d + 2 * i, &vSc, vmc);
        if (vdepths) vmc[2] = vmc[2] + vdepths[i];
        m33 vR, vR2, vSigma; float vt[3], vm[3];
        grad_pos_world_to_cam(R, means + 3 * i, vmc, &vR, vt, vm);
        grad_covar_world_to_cam(R, Sigma, vSc, vR, &vR2, &vSigma);
        m33 vRg = m33_zero();
        if (vnormals) {
            float nn[3]; int k; float sg;
            gaussian_normal(R, Rg, scales + 3 * i, mc, nn, &k, &sg);
            for (int r = 0; r < 3; r++) {
                float g = R.m[0][r] * vnormals[3 * i] + R.m[1][r] * vnormals[3 * i + 1] + R.m[2][r] * vnormals[3 * i + 2];
                vRg.m[r][k] = sg * g;
            }
        }
        float vq[4], vs[3];
        grad_quat_scale_to_cov(rots + 4 * i, scales + 3 * i, Rg, vSigma, vRg, vq, vs);
        for (int c = 0; c < 3; c++) { vmeans[3 * i + c] = vm[c]; vscales[3 * i + c] = vs[c]; }
        for (int c = 0; c < 4; c++) vrots[4 * i + c] = vq[c];
        if (vR_out) {
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++)
                    if (fabsf(vR2.m[r][c]) > 1e-7f) {
#pragma omp atomic
                        vRacc[c * 3 + r] += vR2.m[r][c];
                    }
                if (fabsf(vt[r]) > 1e-7f) {
#pragma omp atomic
                    vtacc[r] += vt[r];
                }
            }
        }
    }
    if (vR_out) {
        for (int k = 0; k < 9; k++) vR_out[k] = (float)vRacc[k];
        for (int k = 0; k < 3; k++) vt_out[k] = (float)vtacc[k];
    }
}

/* ------------------------------------------------------------------------- */
/* A.12 fused SSIM (fused_ssim.jl:34-371); layout (W,H,CH,B), x fastest.       */
/* Restated per pixel: separable 11-tap blur, horizontal pass then vertical,   */
/* symmetric-pair accumulation order as in the kernel.                         */
/* ------------------------------------------------------------------------- */
static const float GAUSS[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                0.0075987582094967365f, 0.001028380123898387f};
static inline float pix(const float *img, int W, int H, int x, int y) {
    return (x < 0 || x >= W || y < 0 || y >= H) ? 0.0f : img[(size_t)x + (size_t)W * y];
}
ORC_API void orc_ssim_forward(int W, int H, int CH, int B, const float *img, const float *ref, float C1, float C2,
                              int train, float *ssim_map, float *dm_dmu1, float *dm_dsigma1_sq, float *dm_dsigma12) {
    size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(static) collapse(2)
    for (int pl = 0; pl < CH * B; pl++)
        for (int y = 0; y < H; y++) {
            const float *X = img + plane * pl, *Y = ref + plane * pl;
            for (int x = 0; x < W; x++) {
                float out[5] = {0, 0, 0, 0, 0};
                /* vertical pass over horizontally-convolved rows; both passes accumulate
                 * symmetric pairs d=1..5 (weights GAUSS[5-d]) first, centre last. */
                float row[11][5];
                for (int ry = -5; ry <= 5; ry++) {
                    float s[5] = {0, 0, 0, 0, 0};
                    int yy = y + ry;
                    for (int d = 1; d <= 5; d++) {
                        float w = GAUSS[5 - d];
                        float Xl = pix(X, W, H, x - d, yy), Yl = pix(Y, W, H, x - d, yy);
                        float Xr = pix(X, W, H, x + d, yy), Yr = pix(Y, W, H, x + d, yy);
                        s[0] += (Xl + Xr) * w; s[1] += (Xl * Xl + Xr * Xr) * w;
                        s[2] += (Yl + Yr) * w; s[3] += (Yl * Yl + Yr * Yr) * w;
                        s[4] += (Xl * Yl + Xr * Yr) * w;
                    }
                    float cx = pix(X, W, H, x, yy), cy = pix(Y, W, H, x, yy), wc = GAUSS[5];
                    s[0] += cx * wc; s[1] += cx * cx * wc; s[2] += cy * wc; s[3] += cy * cy * wc; s[4] += cx * cy * wc;
                    for (int k = 0; k < 5; k++) row[ry + 5][k] = s[k];
                }
                for (int d = 1; d <= 5; d++) {
                    float w = GAUSS[5 - d];
                    for (int k = 0; k < 5; k++) out[k] += (row[5 - d][k] + row[5 + d][k]) * w;
                }
                for (int k = 0; k < 5; k++) out[k] += row[5][k] * GAUSS[5];
                float mu1 = out[0], mu2 = out[2];
                float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
                float sigma1_sq = out[1] - mu1_sq, sigma2_sq = out[3] - mu2_sq, sigma12 = out[4] - mu1 * mu2;
                float A = mu1_sq + mu2_sq + C1, Bv = sigma1_sq + sigma2_sq + C2;
                float Cv = 2.0f * mu1 * mu2 + C1, Dv = 2.0f * sigma12 + C2;
                size_t o = plane * pl + (size_t)x + (size_t)W * y;
                ssim_map[o] = (Cv * Dv) / (A * Bv);
                if (train) {
                    dm_dmu1[o] = ((mu2 * 2.0f * Dv) / (A * Bv) - (mu2 * 2.0f * Cv) / (A * Bv) -
                                  (mu1 * 2.0f * Cv * Dv) / (A * A * Bv) + (mu1 * 2.0f * Cv * Dv) / (A * Bv * Bv));
                    dm_dsigma1_sq[o] = (-Cv * Dv) / (A * Bv * Bv);
                    dm_dsigma12[o] = (2.0f * Cv) / (A * Bv);
                }
            }
        }
}
ORC_API void orc_ssim_backward(int W, int H, int CH, int B, const float *img, const float *ref, const float *dL_dmap,
                               const float *dm_dmu1, const float *dm_dsigma1_sq, const float *dm_dsigma12,
                               float *dL_dimg) {
    size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(static) collapse(2)
    for (int pl = 0; pl < CH * B; pl++)
        for (int y = 0; y < H; y++) {
            const float *X = img + plane * pl, *Y = ref + plane * pl, *L = dL_dmap + plane * pl;
            const float *M0 = dm_dmu1 + plane * pl, *M1 = dm_dsigma1_sq + plane * pl, *M2 = dm_dsigma12 + plane * pl;
            for (int x = 0; x < W; x++) {
                float row[11][3];
                for (int ry = -5; ry <= 5; ry++) {
                    int yy = y + ry;
                    float a[3] = {0, 0, 0};
                    for (int d = 1; d <= 5; d++) {
                        float w = GAUSS[5 - d];
                        float cl = pix(L, W, H, x - d, yy), cr = pix(L, W, H, x + d, yy);
                        a[0] += (pix(M0, W, H, x - d, yy) * cl + pix(M0, W, H, x + d, yy) * cr) * w;
                        a[1] += (pix(M1, W, H, x - d, yy) * cl + pix(M1, W, H, x + d, yy) * cr) * w;
                        a[2] += (pix(M2, W, H, x - d, yy) * cl + pix(M2, W, H, x + d, yy) * cr) * w;
                    }
                    float cc = pix(L, W, H, x, yy), wc = GAUSS[5];
                    a[0] += pix(M0, W, H, x, yy) * cc * wc;
                    a[1] += pix(M1, W, H, x, yy) * cc * wc;
                    a[2] += pix(M2, W, H, x, yy) * cc * wc;
                    for (int k = 0; k < 3; k++) row[ry + 5][k] = a[k];
                }
                float s[3] = {0, 0, 0};
                for (int d = 1; d <= 5; d++) {
                    float w = GAUSS[5 - d];
                    for (int k = 0; k < 3; k++) s[k] += (row[5 - d][k] + row[5 + d][k]) * w;
                }
                for (int k = 0; k < 3; k++) s[k] += row[5][k] * GAUSS[5];
                size_t o = plane * pl + (size_t)x + (size_t)W * y;
                float p1 = X[(size_t)x + (size_t)W * y], p2 = Y[(size_t)x + (size_t)W * y];
                dL_dimg[o] = s[0] + 2.0f * p1 * s[1] + p2 * s[2];
            }
        }
}

static inline float scales_dim_pick(const float *scales, int scale_dims, int c, int i) {
    return scale_dims == 1 ? scales[i] : scales[c + 3 * (size_t)i];
}
/* ------------------------------------------------------------------------- */
/* A.1 functor prologue (rasterizer.jl:200-253) and its pullback              */
/* shs = hcat(sh_color (3,1,N), sh_remainder (3,KR,N)); σ(opacities);          */
/* exp(scales), an isotropic (1,N) scale tiled x3 (rasterizer.jl:235-247).     */
/* NerfUtils.sigmoid (external, NerfUtils 0.2) restated as 1/(1+exp(-x)).      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_prologue_forward(int n, int k_rest, int scale_dims, const float *sh_color, const float *sh_remainder,
                                  const float *opacities, const float *scales, float *shs, float *opacities_act,
                                  float *scales_act) {
    const int K = 1 + k_rest;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) shs[c + 3 * (size_t)K * i] = sh_color[c + 3 * (size_t)i];
        for (int j = 0; j < 3 * k_rest; j++) shs[3 + j + 3 * (size_t)K * i] = sh_remainder[j + 3 * (size_t)k_rest * i];
        opacities_act[i] = 1.0f / (1.0f + expf(-opacities[i]));
        for (int c = 0; c < 3; c++)
            scales_act[c + 3 * (size_t)i] = expf(scales_dim_pick(scales, scale_dims, c, i));
    }
}
/* what ChainRules derives for rasterizer.jl:218-247: hcat -> split; σ' = σ(1-σ); exp' = exp;
 * vcat(s, s, s) -> sum of the three cotangents */
ORC_API void orc_prologue_backward(int n, int k_rest, int scale_dims, const float *opacities_act,
                                   const float *scales_act, const float *vshs, const float *vopacities_act,
                                   const float *vscales_act, float *v_sh_color, float *v_sh_remainder,
                                   float *v_opacities, float *v_scales) {
    const int K = 1 + k_rest;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) v_sh_color[c + 3 * (size_t)i] = vshs[c + 3 * (size_t)K * i];
        for (int j = 0; j < 3 * k_rest; j++) v_sh_remainder[j + 3 * (size_t)k_rest * i] = vshs[3 + j + 3 * (size_t)K * i];
        const float a = opacities_act[i];
        v_opacities[i] = vopacities_act[i] * (a * (1.0f - a));
        float g[3];
        for (int c = 0; c < 3; c++) g[c] = vscales_act[c + 3 * (size_t)i] * scales_act[c + 3 * (size_t)i];
        if (scale_dims == 1) v_scales[i] = (g[0] + g[1]) + g[2];
        else for (int c = 0; c < 3; c++) v_scales[c + 3 * (size_t)i] = g[c];
    }
}

/* ------------------------------------------------------------------------- */
/* Adam step of `NU.step!` (call sites training.jl:234-239,778).  NerfUtils    */
/* 0.2 is an external dependency absent from /root/reference (Project.toml:76) */
/* — PARITY UNPINNED; restated from Kingma & Ba 2015, Algorithm 1 in the       */
/* "efficient" ordering of its §2 that NerfUtils uses:                         */
/*   μ = β1 μ + (1-β1) g;  ν = β2 ν + (1-β2) g²;                               */
/*   lr_t = lr · sqrt(1-β2^t) / (1-β1^t);  θ -= lr_t · μ / (sqrt(ν) + ϵ)       */
/* with t the optimizer's step counter after increment (ϵ = 1f-15 at every     */
/* reference call site).                                                       */
/* ------------------------------------------------------------------------- */
ORC_API float orc_adam_lr_t(float lr, float beta1, float beta2, uint32_t step) {
    const float t = (float)step;
    return lr * sqrtf(1.0f - powf(beta2, t)) / (1.0f - powf(beta1, t));
}
ORC_API void orc_adam_step(int64_t count, float *theta, const float *grad, float *mu, float *nu, uint32_t step,
                           float lr, float beta1, float beta2, float eps) {
    const float lr_t = orc_adam_lr_t(lr, beta1, beta2, step);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; i++) {
        const float g = grad[i];
        const float m = beta1 * mu[i] + (1.0f - beta1) * g;
        const float v = beta2 * nu[i] + (1.0f - beta2) * (g * g);
        mu[i] = m;
        nu[i] = v;
        theta[i] = theta[i] - lr_t * m / (sqrtf(v) + eps);
    }
}
