// shade.hip -- the elementwise half of the Stage-I render step on gfx950: NeuS alpha from (sdf, normal), split-sum
// shading algebra around the light/material MLPs (IDE / PE encoders, FG-LUT bilinear fetch, sRGB, exp/sigmoid heads),
// NeRF++ head, alpha compositing -- forward and hand-derived backward.  One thread per sample / per ray; all of it is
// HBM-bound glue around the MLP-chain kernels.  Replaces compute_sdf_alpha (network/renderer.py:484-512),
// AppShadingNetwork.forward (network/field.py:591-651), generate_ide_fn (utils/ref_utils.py:53-117), dr.texture
// (field.py:612), linear_to_srgb (utils/raw_utils.py:4-10), compute_density_alpha (renderer.py:514-520) and the
// compositing lines of render_core (renderer.py:578-579), plus everything autograd derived from them.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "ide.h"                                     // IDE_N, the compile-time coefficient table, ide_forward / ide_backward

namespace {

// ------------------------------------------------------------------------------------------------------------------
// IDE (deg_view = 5): 36 (m,l) pairs, l in {1,2,4,8,16}, polynomial coefficients mat[k][i], k <= 16.  The kernels use the
// compile-time table of ide.h; this libm evaluation of utils/ref_utils.py:53-82 stays as its check (once per process).
// ------------------------------------------------------------------------------------------------------------------

double fact(int n) { double r = 1.0; for (int i = 2; i <= n; ++i) r *= i; return r; }
double gen_binom(double a, int k) { double p = 1.0; for (int i = 0; i < k; ++i) p *= (a - i); return p / fact(k); }
double assoc_legendre_coeff(int l, int m, int k) {
    return ((m & 1) ? -1.0 : 1.0) * pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) * gen_binom(0.5 * (l + k + m - 1.0), l);
}
double sph_harm_coeff(int l, int m, int k) {
    return sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * assoc_legendre_coeff(l, m, k);
}

int init_ide_tables() {
    static bool done = false;
    if (done) return 0;
    float mat[17 * IDE_N];
    for (int i = 0; i < 17 * IDE_N; ++i) mat[i] = 0.f;
    int i = 0;
    for (int e = 0; e < 5; ++e) {
        const int l = 1 << e;
        for (int m = 0; m <= l; ++m, ++i) {
            for (int k = 0; k <= l - m; ++k) mat[k * IDE_N + i] = (float)sph_harm_coeff(l, m, k);
        }
    }
    if (!ide_tab_equals(mat)) return -1;               // the compile-time table of ide.h must be this libm one, bit for bit
    done = true;
    return 0;
}

template <int N_FREQ>
__device__ __forceinline__ void pe3(const float* p, float* out) {      // (unrolled: `out` stays in registers)
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = p[c];
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < N_FREQ; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 + 6 * k + c] = sinf(p[c] * f);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 + 6 * k + 3 + c] = cosf(p[c] * f);
        f *= 2.f;
    }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
constexpr float SRGB_EPS = 1.1920928955078125e-07f;
__device__ __forceinline__ float srgb_f(float x) {
    return x <= 0.0031308f ? (323.f / 25.f) * x : (211.f * powf(fmaxf(x, SRGB_EPS), 5.f / 12.f) - 11.f) / 200.f;
}
__device__ __forceinline__ float srgb_grad(float x) {
    if (x <= 0.0031308f) return 323.f / 25.f;
    return x >= SRGB_EPS ? (211.f / 200.f) * (5.f / 12.f) * powf(x, -7.f / 12.f) : 0.f;
}

// bilinear clamp fetch of the FG table lut[256(v)][256(u)][2]; returns d/du, d/dv too
__device__ __forceinline__ void fg_fetch(const float* __restrict__ lut, float u, float v, float& f0, float& f1,
                                         float& df0du, float& df1du, float& df0dv, float& df1dv) {
    const float W = 256.f;
    float uu = u * W - 0.5f, vv = v * W - 0.5f;
    const float gu = (uu >= 0.f && uu <= W - 1.f) ? W : 0.f, gv = (vv >= 0.f && vv <= W - 1.f) ? W : 0.f;
    uu = fminf(fmaxf(uu, 0.f), W - 1.f);
    vv = fminf(fmaxf(vv, 0.f), W - 1.f);
    const float u0 = fminf(floorf(uu), W - 2.f), v0 = fminf(floorf(vv), W - 2.f);
    const float fu = uu - u0, fv = vv - v0;
    const int iu = (int)u0, iv = (int)v0;
    const float2 t00 = reinterpret_cast<const float2*>(lut)[iv * 256 + iu];
    const float2 t01 = reinterpret_cast<const float2*>(lut)[iv * 256 + iu + 1];
    const float2 t10 = reinterpret_cast<const float2*>(lut)[(iv + 1) * 256 + iu];
    const float2 t11 = reinterpret_cast<const float2*>(lut)[(iv + 1) * 256 + iu + 1];
    f0 = (t00.x * (1 - fu) + t01.x * fu) * (1 - fv) + (t10.x * (1 - fu) + t11.x * fu) * fv;
    f1 = (t00.y * (1 - fu) + t01.y * fu) * (1 - fv) + (t10.y * (1 - fu) + t11.y * fu) * fv;
    df0du = ((t01.x - t00.x) * (1 - fv) + (t11.x - t10.x) * fv) * gu;
    df1du = ((t01.y - t00.y) * (1 - fv) + (t11.y - t10.y) * fv) * gu;
    df0dv = ((t10.x * (1 - fu) + t11.x * fu) - (t00.x * (1 - fu) + t01.x * fu)) * gv;
    df1dv = ((t10.y * (1 - fu) + t11.y * fu) - (t00.y * (1 - fu) + t01.y * fu)) * gv;
}

__device__ __forceinline__ void ray_dir(const float* __restrict__ d, int r, float* dh) {
    const float x = d[r * 3], y = d[r * 3 + 1], z = d[r * 3 + 2];
    const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    dh[0] = x / n; dh[1] = y / n; dh[2] = z / n;
}

// ------------------------------------------------------------------------------------------------------------------
// inner geometry: alpha, shading frame, eikonal term                                     (renderer.py:484-512, 574)
// geo[k] = { nhat(3), NoV, refl(3), |grad| }
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float inv_s_from(const float* variance) {
    return fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);
}

__global__ void sdf_alpha_fwd_kernel(const float* __restrict__ sdf4, const float* __restrict__ grad, const float* __restrict__ x4,
                                     const int* __restrict__ idx, const float* __restrict__ d, int T, const float* __restrict__ variance,
                                     float anneal, int n, float* __restrict__ alpha, float* __restrict__ geo, float* __restrict__ gerr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float s = inv_s_from(variance);
    float dh[3];
    ray_dir(d, idx[k] / T, dh);
    const float g[3] = {grad[k * 3], grad[k * 3 + 1], grad[k * 3 + 2]};
    const float sdf = sdf4[(size_t)k * 4], dist = x4[(size_t)k * 4 + 3];
    const float tc = dh[0] * g[0] + dh[1] * g[1] + dh[2] * g[2];
    const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - anneal) + fmaxf(-tc, 0.f) * anneal);
    const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;
    const float pc = sigmoid_f(ep * s), nc = sigmoid_f(en * s);
    const float raw = (pc - nc + 1e-5f) / (pc + 1e-5f);
    alpha[k] = fminf(fmaxf(raw, 0.f), 1.f);
    const float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const float gd = fmaxf(gn, 1e-12f);
    const float nh[3] = {g[0] / gd, g[1] / gd, g[2] / gd};
    const float nov = -(nh[0] * dh[0] + nh[1] * dh[1] + nh[2] * dh[2]);
    float* o = geo + (size_t)k * 8;
    o[0] = nh[0]; o[1] = nh[1]; o[2] = nh[2]; o[3] = nov;
    o[4] = nov * nh[0] * 2.f + dh[0]; o[5] = nov * nh[1] * 2.f + dh[1]; o[6] = nov * nh[2] * 2.f + dh[2];
    o[7] = gn;
    gerr[k] = (gn - 1.0f) * (gn - 1.0f);
}

// d_geo[k] = { d_nhat(3), d_NoV, d_refl(3), - } from the shader (may be NULL);  outputs d_sdf4[k*4], d_grad[k*3],
// per-thread d_inv_s into dinv[k] (summed by the caller)
__global__ void sdf_alpha_bwd_kernel(const float* __restrict__ sdf4, const float* __restrict__ grad, const float* __restrict__ x4,
                                     const int* __restrict__ idx, const float* __restrict__ d, int T, const float* __restrict__ variance,
                                     float anneal, int n, int n_pad, const float* __restrict__ d_alpha, const float* __restrict__ d_gerr,
                                     const float* __restrict__ d_geo, float* __restrict__ d_sdf4, float* __restrict__ d_grad,
                                     float* __restrict__ dinv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    if (k >= n) { reinterpret_cast<float4*>(d_sdf4)[k] = make_float4(0, 0, 0, 0); dinv[k] = 0.f; return; }
    const float s = inv_s_from(variance);
    float dh[3];
    ray_dir(d, idx[k] / T, dh);
    const float g[3] = {grad[k * 3], grad[k * 3 + 1], grad[k * 3 + 2]};
    const float sdf = sdf4[(size_t)k * 4], dist = x4[(size_t)k * 4 + 3];
    const float tc = dh[0] * g[0] + dh[1] * g[1] + dh[2] * g[2];
    const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - anneal) + fmaxf(-tc, 0.f) * anneal);
    const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;
    const float pc = sigmoid_f(ep * s), nc = sigmoid_f(en * s);
    const float den = pc + 1e-5f;
    const float raw = (pc - nc + 1e-5f) / den;
    const float draw = (raw >= 0.f && raw <= 1.f) ? d_alpha[k] : 0.f;
    const float dpc = draw * nc / (den * den), dnc = -draw / den;
    const float dps = dpc * pc * (1.f - pc), dns = dnc * nc * (1.f - nc);
    const float dep = s * dps, den_ = s * dns;
    dinv[k] = ep * dps + en * dns;
    reinterpret_cast<float4*>(d_sdf4)[k] = make_float4(dep + den_, 0.f, 0.f, 0.f);
    const float dic = (den_ - dep) * dist * 0.5f;
    const float dic_dtc = ((-tc * 0.5f + 0.5f) > 0.f ? 0.5f * (1.0f - anneal) : 0.f) + ((-tc) > 0.f ? anneal : 0.f);
    const float dtc = dic * dic_dtc;
    float dg[3] = {dtc * dh[0], dtc * dh[1], dtc * dh[2]};
    const float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const float gd = fmaxf(gn, 1e-12f);
    const float nh[3] = {g[0] / gd, g[1] / gd, g[2] / gd};
    // eikonal: (|g|-1)^2
#ifdef NERO_DBG_COHERENT_LOADS                        // (race hunt, round 5: device-scope loads of the two inputs only d_grad depends on)
    auto cl = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    const float de = d_gerr ? cl(d_gerr + k) * 2.f * (gn - 1.0f) : 0.f;
    float qv[8];
    if (d_geo) for (int c = 0; c < 8; ++c) qv[c] = cl(d_geo + (size_t)k * 8 + c);
#else
    const float de = d_gerr ? d_gerr[k] * 2.f * (gn - 1.0f) : 0.f;
#endif
    for (int c = 0; c < 3; ++c) dg[c] += de * nh[c];
    if (d_geo) {
#ifdef NERO_DBG_COHERENT_LOADS
        const float* q = qv;
#else
        const float* q = d_geo + (size_t)k * 8;
#endif
        const float v[3] = {-dh[0], -dh[1], -dh[2]};
        const float nov = nh[0] * v[0] + nh[1] * v[1] + nh[2] * v[2];
        const float drn = q[4] * nh[0] + q[5] * nh[1] + q[6] * nh[2];
        float dn[3];
        for (int c = 0; c < 3; ++c) dn[c] = q[c] + q[4 + c] * 2.f * nov + (q[3] + 2.f * drn) * v[c];
        const float dot = dn[0] * nh[0] + dn[1] * nh[1] + dn[2] * nh[2];
        if (gn >= 1e-12f)
            for (int c = 0; c < 3; ++c) dg[c] += (dn[c] - nh[c] * dot) / gd;
    }
    for (int c = 0; c < 3; ++c) d_grad[k * 3 + c] = dg[c];
}

// ------------------------------------------------------------------------------------------------------------------
// light-MLP input encodings (field.py:554-589):  Xd = IDE(n,1) [72], Xs = IDE(refl, rough) [72],
// Xi = [PE8(p)(51), IDE(refl,rough)(72), pad] ld 128,  Xo = [PE8(p)(51), PE6(refl)(39), pad] ld 96
// mat[k] = { metallic, roughness, albedo(3), -, -, - } (after sigmoid) is written here too.
// ------------------------------------------------------------------------------------------------------------------
// shader_config.sphere_direction (field.py:558-562, 582-586): the direction v is complemented by the normalised exit point of the ray
// (q, v) on the unit sphere, q = offset_points_to_sphere(p) (:380-388), t = get_sphere_intersection(q, v) (:390-396)
struct SphereDir { float q[3], b, S, t, len, s[3]; };
__device__ __forceinline__ SphereDir sphere_dir(const float* p, const float* v) {
    SphereDir e;
    const float pn = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int c = 0; c < 3; ++c) e.q[c] = pn > 0.999f ? p[c] / pn * 0.999f : p[c];
    e.b = e.q[0] * v[0] + e.q[1] * v[1] + e.q[2] * v[2];
    const float xtx = e.q[0] * e.q[0] + e.q[1] * e.q[1] + e.q[2] * e.q[2];
    e.S = sqrtf(e.b * e.b - xtx + 1.f + 1e-6f);
    e.t = -e.b + e.S;
    float u[3];
    for (int c = 0; c < 3; ++c) u[c] = e.q[c] + v[c] * e.t;
    e.len = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    for (int c = 0; c < 3; ++c) e.s[c] = u[c] / e.len;
    return e;
}
// dL/dv from gs = dL/ds:  gu = (gs - s <s,gs>) / |u|,  dv = t gu + q (b/S - 1) <v,gu>
__device__ __forceinline__ void sphere_dir_vjp(const SphereDir& e, const float* v, const float* gs, float* dv) {
    const float sg = e.s[0] * gs[0] + e.s[1] * gs[1] + e.s[2] * gs[2];
    float gu[3];
    for (int c = 0; c < 3; ++c) gu[c] = (gs[c] - e.s[c] * sg) / e.len;
    const float vg = v[0] * gu[0] + v[1] * gu[1] + v[2] * gu[2];
    const float k = (e.b / e.S - 1.f) * vg;
    for (int c = 0; c < 3; ++c) dv[c] += e.t * gu[c] + e.q[c] * k;
}

__global__ __launch_bounds__(ROW_BLOCK) void shade_encode_kernel(const float* __restrict__ x4, const float* __restrict__ geo, const float* __restrict__ m_raw,
                                    const float* __restrict__ r_raw, const float* __restrict__ a_raw, int n, int n_pad,
                                    float* __restrict__ mat, float* __restrict__ Xd, float* __restrict__ Xs, float* __restrict__ Xi,
                                    float* __restrict__ Xo, int sphere) {
    __shared__ float stage[ROW_BLOCK * 73];              // <= 72 columns of one output matrix at a time (rows_put / rows_flush, ide.h);
                                                         // rows k >= n are zero rows.  18.7 KB: 8 blocks per CU
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int k = row0 + threadIdx.x;
    const bool live = k < n;
    const int kk = live ? k : 0;                          // (dead threads compute on row 0 and store zeros: every thread reaches the barriers)
    const int ldd = sphere ? 144 : 72;
    const float* q = geo + (size_t)kk * 8;
    const float m = sigmoid_f(m_raw[(size_t)kk * 4]), r = sigmoid_f(r_raw[(size_t)kk * 4]);
    if (live) {
        float* mo = mat + (size_t)k * 8;
        mo[0] = m; mo[1] = r;
        for (int c = 0; c < 3; ++c) mo[2 + c] = sigmoid_f(a_raw[(size_t)k * 4 + c]);
        mo[5] = 0.f; mo[6] = 0.f; mo[7] = 0.f;
    }
    const float z = live ? 1.f : 0.f;
    const float p[3] = {x4[(size_t)kk * 4], x4[(size_t)kk * 4 + 1], x4[(size_t)kk * 4 + 2]};
    float e[72];
    ide_forward<true>(q[0], q[1], q[2], 1.0f, e);
    rows_put<72, 0, 72>(stage, e, z);
    rows_flush<72>(stage, Xd, ldd, 0, row0, n_pad);
    ide_forward<true>(q[4], q[5], q[6], r, e);
    rows_put<72, 0, 72>(stage, e, z);
    rows_flush<72>(stage, Xs, ldd, 0, row0, n_pad);
    float pe[51];
    pe3<8>(p, pe);
    // Xi = [PE-8(p) 51 | IDE(refl, rough) 72 | 0 x 5] as columns 0..63 and 64..127
    rows_put<64, 0, 51>(stage, pe, z);
    {
        float h[13];
#pragma unroll
        for (int c = 0; c < 13; ++c) h[c] = e[c];
        rows_put<64, 51, 13>(stage, h, z);
    }
    rows_flush<64>(stage, Xi, 128, 0, row0, n_pad);
    {
        float h[59];
#pragma unroll
        for (int c = 0; c < 59; ++c) h[c] = e[13 + c];
        rows_put<64, 0, 59>(stage, h, z);
    }
    rows_zero<64, 59, 5>(stage);
    rows_flush<64>(stage, Xi, 128, 64, row0, n_pad);
    // Xo = [PE-8(p) 51 | PE-6(refl) 39 | 0 x 6] as columns 0..47 and 48..95
    float pr[39];
    {
        const float rf[3] = {q[4], q[5], q[6]};
        pe3<6>(rf, pr);
    }
    {
        float h[48];
#pragma unroll
        for (int c = 0; c < 48; ++c) h[c] = pe[c];
        rows_put<48, 0, 48>(stage, h, z);
    }
    rows_flush<48>(stage, Xo, 96, 0, row0, n_pad);
    {
        float h[3] = {pe[48], pe[49], pe[50]};
        rows_put<48, 0, 3>(stage, h, z);
    }
    rows_put<48, 3, 39>(stage, pr, z);
    rows_zero<48, 42, 6>(stage);
    rows_flush<48>(stage, Xo, 96, 48, row0, n_pad);
    if (sphere) {                                         // (uniform over the launch)
        const float nv[3] = {q[0], q[1], q[2]}, rv[3] = {q[4], q[5], q[6]};
        const SphereDir sn = sphere_dir(p, nv);
        ide_forward<true>(sn.s[0], sn.s[1], sn.s[2], 1.0f, e);
        rows_put<72, 0, 72>(stage, e, z);
        rows_flush<72>(stage, Xd, ldd, 72, row0, n_pad);
        const SphereDir sr = sphere_dir(p, rv);
        ide_forward<true>(sr.s[0], sr.s[1], sr.s[2], r, e);
        rows_put<72, 0, 72>(stage, e, z);
        rows_flush<72>(stage, Xs, ldd, 72, row0, n_pad);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// human ("photo capturer") light input: intersection of the reflected ray with the z=0 plane of the per-image human
// frame, IPE of the hit position (predict_human_light / get_camera_plane_intersection / IPE, field.py:348-378, 536-552)
// Xh [rows,24]; hmask[k] = hit flag (1/0)
// ------------------------------------------------------------------------------------------------------------------
struct HumanGeom { float px, py, pz, dx, dy, dz, dzp, dist, ix, iy, h; bool hits0; };

__device__ __forceinline__ HumanGeom human_geom(const float* __restrict__ pose, const float* p, const float* rf) {
    HumanGeom g;
    g.px = pose[0] * p[0] + pose[1] * p[1] + pose[2] * p[2] + pose[3];
    g.py = pose[4] * p[0] + pose[5] * p[1] + pose[6] * p[2] + pose[7];
    g.pz = pose[8] * p[0] + pose[9] * p[1] + pose[10] * p[2] + pose[11];
    g.dx = pose[0] * rf[0] + pose[1] * rf[1] + pose[2] * rf[2];
    g.dy = pose[4] * rf[0] + pose[5] * rf[1] + pose[6] * rf[2];
    g.dz = pose[8] * rf[0] + pose[9] * rf[1] + pose[10] * rf[2];
    g.hits0 = fabsf(g.dz) > 1e-4f;
    g.dzp = g.hits0 ? g.dz : 1e-4f;
    g.dist = -g.pz / g.dzp;
    g.ix = g.px + g.dist * g.dx;
    g.iy = g.py + g.dist * g.dy;
    const float mx = g.ix * 0.3f, my = g.iy * 0.3f;
    g.h = (g.hits0 && sqrtf(mx * mx + my * my) < 1.5f && g.dist > 0.f) ? 1.f : 0.f;
    return g;
}

__global__ void human_encode_kernel(const float* __restrict__ x4, const float* __restrict__ geo, const float* __restrict__ mat,
                                    const int* __restrict__ idx, int T, const float* __restrict__ poses, int n, int n_pad,
                                    float* __restrict__ Xh, float* __restrict__ hmask) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    float* o = Xh + (size_t)k * 24;
    if (k >= n) { for (int c = 0; c < 24; ++c) o[c] = 0.f; hmask[k] = 0.f; return; }
    const float p[3] = {x4[(size_t)k * 4], x4[(size_t)k * 4 + 1], x4[(size_t)k * 4 + 2]};
    const float* q = geo + (size_t)k * 8;
    const float rf[3] = {q[4], q[5], q[6]};
    const HumanGeom g = human_geom(poses + (size_t)(idx[k] / T) * 12, p, rf);
    const float r = mat[(size_t)k * 8 + 1];
    const float mean[2] = {g.ix * 0.3f * g.h, g.iy * 0.3f * g.h};
    const float sd = g.dist * 0.3f;
    const float var = r * sd * sd * g.h;
    float sc = 1.f;
    for (int s = 0; s < 6; ++s) {
        for (int c = 0; c < 2; ++c) {
            const float sm = mean[c] * sc, sv = var * sc * sc;
            const float e = expf(-0.5f * sv);
            o[2 * s + c] = e * sinf(sm);
            o[12 + 2 * s + c] = e * sinf(sm + 1.5707963267948966f);
        }
        sc *= 2.f;
    }
    hmask[k] = g.h;
}

// dXh [rows,24] -> extra[k] = { d_refl(3), d_rough }
__global__ __launch_bounds__(128) void human_encode_bwd_kernel(const float* __restrict__ x4, const float* __restrict__ geo, const float* __restrict__ mat,
                                        const int* __restrict__ idx, int T, const float* __restrict__ poses, int n,
                                        const float* __restrict__ dXh, float* __restrict__ extra) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float p[3] = {x4[(size_t)k * 4], x4[(size_t)k * 4 + 1], x4[(size_t)k * 4 + 2]};
    const float* q = geo + (size_t)k * 8;
    const float rf[3] = {q[4], q[5], q[6]};
    const float* pose = poses + (size_t)(idx[k] / T) * 12;
    const HumanGeom g = human_geom(pose, p, rf);
    const float r = mat[(size_t)k * 8 + 1];
    const float mean[2] = {g.ix * 0.3f * g.h, g.iy * 0.3f * g.h};
    const float sd = g.dist * 0.3f;
    const float var = r * sd * sd * g.h;
    const float* gx = dXh + (size_t)k * 24;
    float dmean[2] = {0.f, 0.f}, dvar = 0.f;
    float sc = 1.f;
    for (int s = 0; s < 6; ++s) {
        for (int c = 0; c < 2; ++c) {
            const float sm = mean[c] * sc, sv = var * sc * sc;
            const float e = expf(-0.5f * sv);
            const float sn = sinf(sm), cs = sinf(sm + 1.5707963267948966f), ms = cosf(sm + 1.5707963267948966f);
            const float ga = gx[2 * s + c], gb = gx[12 + 2 * s + c];
            dmean[c] += sc * (ga * e * cosf(sm) + gb * e * ms);
            dvar += sc * sc * (-0.5f) * (ga * e * sn + gb * e * cs);
        }
        sc *= 2.f;
    }
    const float dix = 0.3f * g.h * dmean[0], diy = 0.3f * g.h * dmean[1];
    const float d_rough = dvar * sd * sd * g.h;
    float d_dist = dvar * r * 2.f * 0.09f * g.dist * g.h + dix * g.dx + diy * g.dy;
    const float ddx = g.dist * dix, ddy = g.dist * diy;
    const float ddz = g.hits0 ? d_dist * g.pz / (g.dzp * g.dzp) : 0.f;
    float* o = extra + (size_t)k * 4;
    o[0] = pose[0] * ddx + pose[4] * ddy + pose[8] * ddz;
    o[1] = pose[1] * ddx + pose[5] * ddy + pose[9] * ddz;
    o[2] = pose[2] * ddx + pose[6] * ddy + pose[10] * ddz;
    o[3] = d_rough;
}

// combine (field.py:601-623).  light heads are RAW outputs [rows,4]: diff(3), direct(3), indirect(3), occ(1).
// color[k*3..], occ_prob[k] (unclamped, for the occ loss)
__global__ void shade_combine_fwd_kernel(const float* __restrict__ geo, const float* __restrict__ mat, const float* __restrict__ Ld,
                                         const float* __restrict__ Ls, const float* __restrict__ Li, const float* __restrict__ Lo,
                                         const float* __restrict__ lut, float exp_max, int n, float* __restrict__ color,
                                         float* __restrict__ occ_prob, const float* __restrict__ Lh, const float* __restrict__ hmask) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* mo = mat + (size_t)k * 8;
    // human light: exp(min(raw,0)) * hit; channel 3 is the blend weight clamped to [0,1]   (field.py:548-551, 575)
    float hl[3] = {0.f, 0.f, 0.f}, hw = 0.f;
    if (Lh) {
        const float hm = hmask[k];
        for (int c = 0; c < 3; ++c) hl[c] = expf(fminf(Lh[(size_t)k * 4 + c], 0.f)) * hm;
        hw = fminf(fmaxf(expf(fminf(Lh[(size_t)k * 4 + 3], 0.f)) * hm, 0.f), 1.f);
    }
    const float m = mo[0], r = mo[1];
    const float nov = geo[(size_t)k * 8 + 3];
    float f0, f1, a0, a1, a2, a3;
    fg_fetch(lut, fminf(fmaxf(nov, 0.f), 1.f), fminf(fmaxf(r, 0.f), 1.f), f0, f1, a0, a1, a2, a3);
    const float occ = Lo[(size_t)k * 4] * 0.5f + 0.5f;
    const float oc = fminf(fmaxf(occ, 0.f), 1.f);
    occ_prob[k] = occ;
    for (int c = 0; c < 3; ++c) {
        const float a = mo[2 + c];
        const float dl = expf(fminf(Ld[(size_t)k * 4 + c], exp_max));
        const float direct = expf(fminf(Ls[(size_t)k * 4 + c], exp_max));
        const float indirect = expf(fminf(Li[(size_t)k * 4 + c], exp_max));
        const float sl = indirect * oc + (hl[c] * hw + direct * (1.f - hw)) * (1.f - oc);
        const float da = (1.f - m) * a, sa = 0.04f * (1.f - m) + m * a;
        const float lin = da * dl + (sa * f0 + f1) * sl;
        color[(size_t)k * 3 + c] = fminf(fmaxf(srgb_f(lin), 0.f), 1.f);
    }
}

// validation-only intermediates of the shader (inter_results=True, field.py:630-649): rec[k][32] =
//  0-2 specular_albedo, 3-5 clamp(specular_ref), 6-8 clamp(sRGB(specular_light)), 9-11 clamp(sRGB(specular_color)),
//  12-14 diffuse_albedo, 15-17 clamp(sRGB(diffuse_light)), 18-20 clamp(sRGB(diffuse_color)), 21 metallic, 22 roughness,
//  23 clamp(occ_prob), 24-26 indirect_light*occ, 27-29 sRGB(human_light*weight), 30-31 unused
__global__ void shade_inter_kernel(const float* __restrict__ geo, const float* __restrict__ mat, const float* __restrict__ Ld,
                                   const float* __restrict__ Ls, const float* __restrict__ Li, const float* __restrict__ Lo,
                                   const float* __restrict__ lut, float exp_max, int n, const float* __restrict__ Lh,
                                   const float* __restrict__ hmask, float* __restrict__ rec) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* mo = mat + (size_t)k * 8;
    const float m = mo[0], r = mo[1];
    const float nov = geo[(size_t)k * 8 + 3];
    float f0, f1, a0, a1, a2, a3;
    fg_fetch(lut, fminf(fmaxf(nov, 0.f), 1.f), fminf(fmaxf(r, 0.f), 1.f), f0, f1, a0, a1, a2, a3);
    float hl[3] = {0.f, 0.f, 0.f}, hw = 0.f;
    if (Lh) {
        const float hm = hmask[k];
        for (int c = 0; c < 3; ++c) hl[c] = expf(fminf(Lh[(size_t)k * 4 + c], 0.f)) * hm;
        hw = fminf(fmaxf(expf(fminf(Lh[(size_t)k * 4 + 3], 0.f)) * hm, 0.f), 1.f);
    }
    const float oc = fminf(fmaxf(Lo[(size_t)k * 4] * 0.5f + 0.5f, 0.f), 1.f);
    float* o = rec + (size_t)k * 32;
    for (int c = 0; c < 3; ++c) {
        const float a = mo[2 + c];
        const float dl = expf(fminf(Ld[(size_t)k * 4 + c], exp_max));
        const float direct = expf(fminf(Ls[(size_t)k * 4 + c], exp_max));
        const float indirect = expf(fminf(Li[(size_t)k * 4 + c], exp_max));
        const float sl = indirect * oc + (hl[c] * hw + direct * (1.f - hw)) * (1.f - oc);
        const float da = (1.f - m) * a, sa = 0.04f * (1.f - m) + m * a;
        const float sref = sa * f0 + f1;
        o[c] = sa;
        o[3 + c] = fminf(fmaxf(sref, 0.f), 1.f);
        o[6 + c] = fminf(fmaxf(srgb_f(sl), 0.f), 1.f);
        o[9 + c] = fminf(fmaxf(srgb_f(sref * sl), 0.f), 1.f);
        o[12 + c] = da;
        o[15 + c] = fminf(fmaxf(srgb_f(dl), 0.f), 1.f);
        o[18 + c] = fminf(fmaxf(srgb_f(da * dl), 0.f), 1.f);
        o[24 + c] = indirect * oc;
        o[27 + c] = srgb_f(hl[c] * hw);
    }
    o[21] = m; o[22] = r; o[23] = oc; o[30] = 0.f; o[31] = 0.f;
}

// backward of combine: writes the gradients of the RAW head outputs (dLd, dLs, dLi, dLo [rows,4]), the partial
// material grads dmat[k] = { d_metallic, d_rough(LUT part), d_albedo(3), d_NoV (LUT part: handed to shade_encode_bwd, which writes d_geo) }
__global__ void shade_combine_bwd_kernel(const float* __restrict__ geo, const float* __restrict__ mat, const float* __restrict__ Ld,
                                         const float* __restrict__ Ls, const float* __restrict__ Li, const float* __restrict__ Lo,
                                         const float* __restrict__ lut, float exp_max, int n, int n_pad,
                                         const float* __restrict__ d_color, const float* __restrict__ d_occ,
                                         float* __restrict__ dLd, float* __restrict__ dLs, float* __restrict__ dLi, float* __restrict__ dLo,
                                         float* __restrict__ dmat, float* __restrict__ d_geo, const float* __restrict__ Lh,
                                         const float* __restrict__ hmask, float* __restrict__ dLh) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    float4 z4 = make_float4(0, 0, 0, 0);
    if (k >= n) {
        reinterpret_cast<float4*>(dLd)[k] = z4; reinterpret_cast<float4*>(dLs)[k] = z4;
        reinterpret_cast<float4*>(dLi)[k] = z4; reinterpret_cast<float4*>(dLo)[k] = z4;
        if (dLh) reinterpret_cast<float4*>(dLh)[k] = z4;
        return;
    }
    float hl[3] = {0.f, 0.f, 0.f}, hw = 0.f, hw_raw = 0.f, hm = 0.f;
    if (Lh) {
        hm = hmask[k];
        for (int c = 0; c < 3; ++c) hl[c] = expf(fminf(Lh[(size_t)k * 4 + c], 0.f)) * hm;
        hw_raw = expf(fminf(Lh[(size_t)k * 4 + 3], 0.f)) * hm;
        hw = fminf(fmaxf(hw_raw, 0.f), 1.f);
    }
    float d_hl[3] = {0.f, 0.f, 0.f}, d_hw = 0.f;
    const float* mo = mat + (size_t)k * 8;
    const float m = mo[0], r = mo[1];
    const float nov = geo[(size_t)k * 8 + 3];
    float f0, f1, df0du, df1du, df0dv, df1dv;
    fg_fetch(lut, fminf(fmaxf(nov, 0.f), 1.f), fminf(fmaxf(r, 0.f), 1.f), f0, f1, df0du, df1du, df0dv, df1dv);
    const float occ = Lo[(size_t)k * 4] * 0.5f + 0.5f;
    const float oc = fminf(fmaxf(occ, 0.f), 1.f);
    float d_m = 0.f, d_f0 = 0.f, d_f1 = 0.f, d_oc = 0.f;
    float ld[4] = {0, 0, 0, 0}, ls[4] = {0, 0, 0, 0}, li[4] = {0, 0, 0, 0};
    float da_out[3];
    for (int c = 0; c < 3; ++c) {
        const float a = mo[2 + c];
        const float rd = Ld[(size_t)k * 4 + c], rs = Ls[(size_t)k * 4 + c], ri = Li[(size_t)k * 4 + c];
        const float dl = expf(fminf(rd, exp_max)), direct = expf(fminf(rs, exp_max)), indirect = expf(fminf(ri, exp_max));
        const float bl = hl[c] * hw + direct * (1.f - hw);
        const float sl = indirect * oc + bl * (1.f - oc);
        const float dalb = (1.f - m) * a, salb = 0.04f * (1.f - m) + m * a;
        const float sref = salb * f0 + f1;
        const float lin = dalb * dl + sref * sl;
        const float sg = srgb_f(lin);
        const float dlin = (sg >= 0.f && sg <= 1.f) ? d_color[(size_t)k * 3 + c] * srgb_grad(lin) : 0.f;
        const float d_dalb = dlin * dl, d_dl = dlin * dalb, d_sref = dlin * sl, d_sl = dlin * sref;
        const float d_salb = d_sref * f0;
        d_f0 += d_sref * salb;
        d_f1 += d_sref;
        d_m += -a * d_dalb + (a - 0.04f) * d_salb;
        da_out[c] = (1.f - m) * d_dalb + m * d_salb;
        ld[c] = rd <= exp_max ? d_dl * dl : 0.f;
        li[c] = ri <= exp_max ? d_sl * oc * indirect : 0.f;
        const float d_bl = d_sl * (1.f - oc);
        ls[c] = rs <= exp_max ? d_bl * (1.f - hw) * direct : 0.f;
        d_hl[c] = d_bl * hw;
        d_hw += d_bl * (hl[c] - direct);
        d_oc += d_sl * (indirect - bl);
    }
    if (dLh) {
        float o4[4];
        for (int c = 0; c < 3; ++c) o4[c] = Lh[(size_t)k * 4 + c] <= 0.f ? d_hl[c] * hl[c] : 0.f;        // hl = exp(raw)*hm
        const float dhw_raw = (hw_raw >= 0.f && hw_raw <= 1.f) ? d_hw : 0.f;
        o4[3] = Lh[(size_t)k * 4 + 3] <= 0.f ? dhw_raw * hw_raw : 0.f;
        reinterpret_cast<float4*>(dLh)[k] = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    float d_occ_tot = (occ >= 0.f && occ <= 1.f) ? d_oc : 0.f;
    if (d_occ) d_occ_tot += d_occ[k];
    reinterpret_cast<float4*>(dLd)[k] = make_float4(ld[0], ld[1], ld[2], 0.f);
    reinterpret_cast<float4*>(dLs)[k] = make_float4(ls[0], ls[1], ls[2], 0.f);
    reinterpret_cast<float4*>(dLi)[k] = make_float4(li[0], li[1], li[2], 0.f);
    reinterpret_cast<float4*>(dLo)[k] = make_float4(0.5f * d_occ_tot, 0.f, 0.f, 0.f);
    const float d_u = (nov >= 0.f && nov <= 1.f) ? d_f0 * df0du + d_f1 * df1du : 0.f;
    const float d_v = (r >= 0.f && r <= 1.f) ? d_f0 * df0dv + d_f1 * df1dv : 0.f;
    // Whole 32-byte rows, and d_NoV travels in dmat[k][5] (round 5): rounds 1-4 dropped it into d_geo[k][3] here, a 4-byte store per
    // 32-byte row into a buffer a memset had zeroed and shade_encode_bwd then completed with seven more scalar stores -- three kernels
    // building the same cache lines from byte-masked pieces.  With a second stream's kernels running beside the step (NERO_STREAMS=3)
    // sdf_alpha_bwd then occasionally read 16-row blocks of d_geo WITHOUT shade_encode_bwd's part (scripts/r05/dbg_streams.py: d_grad off by
    // exactly the size of those terms, the buffer itself final and identical afterwards).  Now every row of d_geo is written once, by
    // one kernel, as two float4 stores.
    float4* dm = reinterpret_cast<float4*>(dmat + (size_t)k * 8);
    dm[0] = make_float4(d_m, d_v, da_out[0], da_out[1]);
    dm[1] = make_float4(da_out[2], d_u, 0.f, 0.f);
    (void)d_geo;
}

// backward of the encodings: dXd, dXs [rows,72], dXi [rows,128] (cols 51..122 = IDE part) -> d_geo (d_nhat, d_NoV from dmat[k][5],
// d_refl: whole rows), total roughness gradient; then the RAW material head gradients dm_raw/dr_raw/da_raw [rows,4]
// (Measured and dropped: bringing the gradient rows in through LDS like the forward encoders' stores (rows_load, rows.h) -- 95 -> 234 us:
// four staged loads mean eight barriers, the LDS reads sit inside the unrolled IDE chains, and at 274 VGPRs one wave per SIMD hides none of it.)
__global__ __launch_bounds__(128) void shade_encode_bwd_kernel(const float* __restrict__ geo, const float* __restrict__ mat, const float* __restrict__ dXd,
                                        const float* __restrict__ dXs, const float* __restrict__ dXi, const float* __restrict__ dmat,
                                        int n, int n_pad, float* __restrict__ d_geo, float* __restrict__ dm_raw,
                                        float* __restrict__ dr_raw, float* __restrict__ da_raw, const float* __restrict__ extra,
                                        const float* __restrict__ x4, int sphere) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    float4 z4 = make_float4(0, 0, 0, 0);
    if (k >= n) {
        reinterpret_cast<float4*>(dm_raw)[k] = z4; reinterpret_cast<float4*>(dr_raw)[k] = z4; reinterpret_cast<float4*>(da_raw)[k] = z4;
        reinterpret_cast<float4*>(d_geo)[2 * k] = z4; reinterpret_cast<float4*>(d_geo)[2 * k + 1] = z4;
        return;
    }
    const float* q = geo + (size_t)k * 8;
    const float* mo = mat + (size_t)k * 8;
    const float* dm = dmat + (size_t)k * 8;
    const float m = mo[0], r = mo[1];
    const int ldd = sphere ? 144 : 72;
    const float* gd = dXd + (size_t)k * ldd;
    const float* gs_ = dXs + (size_t)k * ldd;
    const float* gi_ = dXi + (size_t)k * 128 + 51;
    float dnx = 0.f, dny = 0.f, dnz = 0.f, dk1 = 0.f;
    ide_backward<true>(q[0], q[1], q[2], 1.0f, [&](int c) { return gd[c]; }, dnx, dny, dnz, dk1);
    float drx = 0.f, dry = 0.f, drz = 0.f, dkr = 0.f;
    ide_backward<true>(q[4], q[5], q[6], r, [&](int c) { return gs_[c] + gi_[c]; }, drx, dry, drz, dkr);
    if (sphere) {
        const float p[3] = {x4[(size_t)k * 4], x4[(size_t)k * 4 + 1], x4[(size_t)k * 4 + 2]};
        const float nv[3] = {q[0], q[1], q[2]}, rv[3] = {q[4], q[5], q[6]};
        float gs[3], dv[3], dk = 0.f;
        const SphereDir sn = sphere_dir(p, nv);
        gs[0] = gs[1] = gs[2] = 0.f;
        ide_backward<true>(sn.s[0], sn.s[1], sn.s[2], 1.0f, [&](int c) { return gd[72 + c]; }, gs[0], gs[1], gs[2], dk);
        dv[0] = dv[1] = dv[2] = 0.f;
        sphere_dir_vjp(sn, nv, gs, dv);
        dnx += dv[0]; dny += dv[1]; dnz += dv[2];
        const SphereDir sr = sphere_dir(p, rv);
        gs[0] = gs[1] = gs[2] = 0.f;
        ide_backward<true>(sr.s[0], sr.s[1], sr.s[2], r, [&](int c) { return gs_[72 + c]; }, gs[0], gs[1], gs[2], dkr);
        dv[0] = dv[1] = dv[2] = 0.f;
        sphere_dir_vjp(sr, rv, gs, dv);
        drx += dv[0]; dry += dv[1]; drz += dv[2];
    }
    float d_r = dm[1] + dkr;
    if (extra) {                                   // human-light branch: d_refl(3), d_rough
        const float* ex = extra + (size_t)k * 4;
        drx += ex[0]; dry += ex[1]; drz += ex[2]; d_r += ex[3];
    }
    reinterpret_cast<float4*>(d_geo)[2 * k] = make_float4(dnx, dny, dnz, dm[5]);         // (d_NoV from shade_combine_bwd: dmat[k][5])
    reinterpret_cast<float4*>(d_geo)[2 * k + 1] = make_float4(drx, dry, drz, 0.f);
    reinterpret_cast<float4*>(dm_raw)[k] = make_float4(dm[0] * m * (1.f - m), 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(dr_raw)[k] = make_float4(d_r * r * (1.f - r), 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(da_raw)[k] = make_float4(dm[2] * mo[2] * (1.f - mo[2]), dm[3] * mo[3] * (1.f - mo[3]), dm[4] * mo[4] * (1.f - mo[4]), 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// NeRF++ head (renderer.py:346-347, 518-519): alpha = 1 - exp(-softplus(sigma) dist), color = sRGB(exp(min(rgb,5)))
// ------------------------------------------------------------------------------------------------------------------
__global__ void nerf_head_fwd_kernel(const float* __restrict__ sig4, const float* __restrict__ rgb4, const float* __restrict__ dist,
                                     int n, float* __restrict__ alpha, float* __restrict__ color) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float s = sig4[(size_t)k * 4];
    const float sp = s > 20.f ? s : log1pf(expf(s));
    alpha[k] = 1.0f - expf(-sp * dist[k]);
    for (int c = 0; c < 3; ++c) color[(size_t)k * 3 + c] = srgb_f(expf(fminf(rgb4[(size_t)k * 4 + c], 5.0f)));
}

__global__ void nerf_head_bwd_kernel(const float* __restrict__ sig4, const float* __restrict__ rgb4, const float* __restrict__ dist,
                                     int n, int n_pad, const float* __restrict__ d_alpha, const float* __restrict__ d_color,
                                     float* __restrict__ d_sig4, float* __restrict__ d_rgb4) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    if (k >= n) { reinterpret_cast<float4*>(d_sig4)[k] = make_float4(0, 0, 0, 0); reinterpret_cast<float4*>(d_rgb4)[k] = make_float4(0, 0, 0, 0); return; }
    const float s = sig4[(size_t)k * 4];
    const float sp = s > 20.f ? s : log1pf(expf(s));
    const float dsp = s > 20.f ? 1.f : sigmoid_f(s);
    const float ds = d_alpha[k] * expf(-sp * dist[k]) * dist[k] * dsp;
    reinterpret_cast<float4*>(d_sig4)[k] = make_float4(ds, 0.f, 0.f, 0.f);
    float o[3];
    for (int c = 0; c < 3; ++c) {
        const float raw = rgb4[(size_t)k * 4 + c];
        const float e = expf(fminf(raw, 5.0f));
        o[c] = raw <= 5.0f ? d_color[(size_t)k * 3 + c] * srgb_grad(e) * e : 0.f;
    }
    reinterpret_cast<float4*>(d_rgb4)[k] = make_float4(o[0], o[1], o[2], 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// compositing (renderer.py:578-579): scatter the compact inner/outer results into [R,T], w_i = a_i prod_{j<i}(1-a_j+1e-7)
// ------------------------------------------------------------------------------------------------------------------
__global__ void scatter_samples_kernel(const float* __restrict__ a, const float* __restrict__ c, const int* __restrict__ idx, int n,
                                       float* __restrict__ alphaRT, float* __restrict__ colorRT) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = idx[k];
    alphaRT[s] = a[k];
    for (int q = 0; q < 3; ++q) colorRT[(size_t)s * 3 + q] = c[(size_t)k * 3 + q];
}

__global__ void composite_fwd_kernel(const float* __restrict__ alphaRT, const float* __restrict__ colorRT, int R, int T,
                                     float* __restrict__ weights, float* __restrict__ rgb) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float Tr = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int i = 0; i < T; ++i) {
        const float a = alphaRT[(size_t)r * T + i];
        const float w = a * Tr;
        weights[(size_t)r * T + i] = w;
        const float* c = colorRT + ((size_t)r * T + i) * 3;
        c0 += c[0] * w; c1 += c[1] * w; c2 += c[2] * w;
        Tr *= (1.0f - a + 1e-7f);
    }
    rgb[r * 3] = c0; rgb[r * 3 + 1] = c1; rgb[r * 3 + 2] = c2;
}

// d_rgb [R,3] -> d_alphaRT [R,T], d_colorRT [R,T,3]
__global__ void composite_bwd_kernel(const float* __restrict__ alphaRT, const float* __restrict__ colorRT,
                                     const float* __restrict__ weights, const float* __restrict__ d_rgb, int R, int T,
                                     float* __restrict__ d_alphaRT, float* __restrict__ d_colorRT) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float g0 = d_rgb[r * 3], g1 = d_rgb[r * 3 + 1], g2 = d_rgb[r * 3 + 2];
    float S = 0.f;                                   // sum_{j>i} dw_j w_j
    for (int i = T - 1; i >= 0; --i) {
        const size_t s = (size_t)r * T + i;
        const float a = alphaRT[s], w = weights[s];
        const float* c = colorRT + s * 3;
        const float dw = c[0] * g0 + c[1] * g1 + c[2] * g2;
        d_colorRT[s * 3] = w * g0; d_colorRT[s * 3 + 1] = w * g1; d_colorRT[s * 3 + 2] = w * g2;
        // T_i = w_i / a_i is unstable for a_i -> 0; recompute transmittance by the forward product instead
        d_alphaRT[s] = -S / (1.0f - a + 1e-7f);      // + dw * T_i added in the second sweep
        S += dw * w;
    }
    float Tr = 1.f;
    for (int i = 0; i < T; ++i) {
        const size_t s = (size_t)r * T + i;
        const float a = alphaRT[s];
        const float* c = colorRT + s * 3;
        const float dw = c[0] * g0 + c[1] * g1 + c[2] * g2;
        d_alphaRT[s] += dw * Tr;
        Tr *= (1.0f - a + 1e-7f);
    }
}

// ---- compositing, one wave per ray (lane i <-> samples i, i + 64, ...): coalesced row loads / stores; the transmittance is a wave
// prefix product and the backward's tail sum a wave suffix sum (fp32; the compositing outputs are floating point, so a different
// association is within their 1e-4 tolerance -- the bit-exact contracts are the sampler's integer outputs, not these) --------------
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v *= t; }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_suffix_sum(float v, int lane) {       // inclusive: sum over lanes >= this one
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(v, o); if (lane + o < 64) v += t; }
    return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void composite_fwd_wave_kernel(const float* __restrict__ alphaRT, const float* __restrict__ colorRT, int R, int T,
                                                                 float* __restrict__ weights, float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float carry = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int i0 = 0; i0 < T; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < T;
        const size_t s = (size_t)r * T + (ok ? i : 0);
        const float a = ok ? alphaRT[s] : 0.f;
        const float f = ok ? (1.0f - a + 1e-7f) : 1.f;
        const float incl = wave_incl_prod(f, lane);
        const float up = __shfl_up(incl, 1);
        const float Tr = carry * (lane ? up : 1.f);    // exclusive product
        const float w = a * Tr;
        if (ok) {
            weights[s] = w;
            c0 += colorRT[s * 3] * w; c1 += colorRT[s * 3 + 1] * w; c2 += colorRT[s * 3 + 2] * w;
        }
        carry *= __shfl(incl, 63);
    }
    c0 = wave_sum_all(c0); c1 = wave_sum_all(c1); c2 = wave_sum_all(c2);
    if (lane == 0) { rgb[r * 3] = c0; rgb[r * 3 + 1] = c1; rgb[r * 3 + 2] = c2; }
}

__global__ __launch_bounds__(256) void composite_bwd_wave_kernel(const float* __restrict__ alphaRT, const float* __restrict__ colorRT,
                                                                 const float* __restrict__ weights, const float* __restrict__ d_rgb, int R, int T,
                                                                 float* __restrict__ d_alphaRT, float* __restrict__ d_colorRT) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float g0 = d_rgb[r * 3], g1 = d_rgb[r * 3 + 1], g2 = d_rgb[r * 3 + 2];
    // S_i = sum_{j > i} dw_j w_j as a SUFFIX scan (not total - prefix: behind an opaque sample the tail terms are ~1e-7 of the total and
    // are divided by 1 - alpha ~ 1e-7 again)
    float a[3], dw[3], q[3];
    int nch = 0;
    for (int i0 = 0; i0 < T && nch < 3; i0 += 64, ++nch) {
        const int i = i0 + lane;
        const bool ok = i < T;
        const size_t s = (size_t)r * T + (ok ? i : 0);
        a[nch] = ok ? alphaRT[s] : 0.f;
        const float w = ok ? weights[s] : 0.f;
        dw[nch] = ok ? (colorRT[s * 3] * g0 + colorRT[s * 3 + 1] * g1 + colorRT[s * 3 + 2] * g2) : 0.f;
        q[nch] = dw[nch] * w;
        if (ok) { d_colorRT[s * 3] = w * g0; d_colorRT[s * 3 + 1] = w * g1; d_colorRT[s * 3 + 2] = w * g2; }
    }
    float S[3];
    float tail = 0.f;
    for (int c = nch - 1; c >= 0; --c) {
        const float incl = wave_suffix_sum(q[c], lane);
        const float dn = __shfl_down(incl, 1);
        S[c] = tail + (lane < 63 ? dn : 0.f);          // exclusive suffix + everything in the later chunks
        tail += __shfl(incl, 0);
    }
    float carry_T = 1.f;
    for (int c = 0; c < nch; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < T;
        const float f = ok ? (1.0f - a[c] + 1e-7f) : 1.f;
        const float ip = wave_incl_prod(f, lane);
        const float up = __shfl_up(ip, 1);
        const float Tr = carry_T * (lane ? up : 1.f);
        if (ok) d_alphaRT[(size_t)r * T + i] = -S[c] / f + dw[c] * Tr;
        carry_T *= __shfl(ip, 63);
    }
}

__global__ void gather_sample_grads_kernel(const float* __restrict__ d_alphaRT, const float* __restrict__ d_colorRT,
                                           const int* __restrict__ idx, int n, float* __restrict__ d_a, float* __restrict__ d_c) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = idx[k];
    d_a[k] = d_alphaRT[s];
    for (int q = 0; q < 3; ++q) d_c[(size_t)k * 3 + q] = d_colorRT[(size_t)s * 3 + q];
}

}  // namespace

#define GRID1D(n) dim3(((n) + 127) / 128), dim3(128), 0, (hipStream_t)stream
#define CHECK_IDE() do { if (init_ide_tables() != 0) return nero_fail(NERO_ERR_LAUNCH, "IDE coefficient table of ide.h differs from the libm one"); } while (0)

extern "C" {

int nero_sdf_alpha_fwd(const float* sdf4, const float* grad, const float* x4, const int* idx, const float* d, int T,
                       const float* variance, float anneal, int n, float* alpha, float* geo, float* gerr, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(sdf_alpha_fwd_kernel, GRID1D(n), sdf4, grad, x4, idx, d, T, variance, anneal, n, alpha, geo, gerr);
    return nero_check_launch("nero_sdf_alpha_fwd");
}

int nero_sdf_alpha_bwd(const float* sdf4, const float* grad, const float* x4, const int* idx, const float* d, int T,
                       const float* variance, float anneal, int n, const float* d_alpha, const float* d_gerr, const float* d_geo,
                       float* d_sdf4, float* d_grad, float* dinv, void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(sdf_alpha_bwd_kernel, GRID1D(n_pad), sdf4, grad, x4, idx, d, T, variance, anneal, n, n_pad, d_alpha, d_gerr, d_geo, d_sdf4, d_grad, dinv);
    return nero_check_launch("nero_sdf_alpha_bwd");
}

int nero_shade_encode(const float* x4, const float* geo, const float* m_raw, const float* r_raw, const float* a_raw, int n,
                      float* mat, float* Xd, float* Xs, float* Xi, float* Xo, int sphere_direction, void* stream) {
    CHECK_IDE();
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(shade_encode_kernel, dim3((n_pad + ROW_BLOCK - 1) / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, x4, geo, m_raw, r_raw, a_raw, n, n_pad, mat, Xd, Xs, Xi, Xo, sphere_direction);
    return nero_check_launch("nero_shade_encode");
}

int nero_shade_combine_fwd(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                           const float* lut, float exp_max, int n, float* color, float* occ_prob, const float* Lh, const float* hmask,
                           void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(shade_combine_fwd_kernel, GRID1D(n), geo, mat, Ld, Ls, Li, Lo, lut, exp_max, n, color, occ_prob, Lh, hmask);
    return nero_check_launch("nero_shade_combine_fwd");
}

int nero_shade_inter_results(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                             const float* lut, float exp_max, int n, const float* Lh, const float* hmask, float* rec, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(shade_inter_kernel, GRID1D(n), geo, mat, Ld, Ls, Li, Lo, lut, exp_max, n, Lh, hmask, rec);
    return nero_check_launch("nero_shade_inter_results");
}

int nero_shade_combine_bwd(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                           const float* lut, float exp_max, int n, const float* d_color, const float* d_occ, float* dLd, float* dLs,
                           float* dLi, float* dLo, float* dmat, float* d_geo, const float* Lh, const float* hmask, float* dLh,
                           void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(shade_combine_bwd_kernel, GRID1D(n_pad), geo, mat, Ld, Ls, Li, Lo, lut, exp_max, n, n_pad, d_color, d_occ, dLd, dLs, dLi, dLo, dmat, d_geo, Lh, hmask, dLh);
    return nero_check_launch("nero_shade_combine_bwd");
}

int nero_shade_encode_bwd(const float* geo, const float* mat, const float* dXd, const float* dXs, const float* dXi, const float* dmat,
                          int n, float* d_geo, float* dm_raw, float* dr_raw, float* da_raw, const float* extra, const float* x4,
                          int sphere_direction, void* stream) {
    CHECK_IDE();
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    if (sphere_direction && !x4) return nero_fail(NERO_ERR_ARG, "nero_shade_encode_bwd: sphere_direction needs x4");
    hipLaunchKernelGGL(shade_encode_bwd_kernel, GRID1D(n_pad), geo, mat, dXd, dXs, dXi, dmat, n, n_pad, d_geo, dm_raw, dr_raw, da_raw, extra, x4,
                       sphere_direction);
    return nero_check_launch("nero_shade_encode_bwd");
}

int nero_human_encode(const float* x4, const float* geo, const float* mat, const int* idx, int T, const float* poses, int n,
                      float* Xh, float* hmask, void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(human_encode_kernel, GRID1D(n_pad), x4, geo, mat, idx, T, poses, n, n_pad, Xh, hmask);
    return nero_check_launch("nero_human_encode");
}

int nero_human_encode_bwd(const float* x4, const float* geo, const float* mat, const int* idx, int T, const float* poses, int n,
                          const float* dXh, float* extra, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(human_encode_bwd_kernel, GRID1D(n), x4, geo, mat, idx, T, poses, n, dXh, extra);
    return nero_check_launch("nero_human_encode_bwd");
}

int nero_nerf_head_fwd(const float* sig4, const float* rgb4, const float* dist, int n, float* alpha, float* color, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(nerf_head_fwd_kernel, GRID1D(n), sig4, rgb4, dist, n, alpha, color);
    return nero_check_launch("nero_nerf_head_fwd");
}

int nero_nerf_head_bwd(const float* sig4, const float* rgb4, const float* dist, int n, const float* d_alpha, const float* d_color,
                       float* d_sig4, float* d_rgb4, void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(nerf_head_bwd_kernel, GRID1D(n_pad), sig4, rgb4, dist, n, n_pad, d_alpha, d_color, d_sig4, d_rgb4);
    return nero_check_launch("nero_nerf_head_bwd");
}

int nero_scatter_samples(const float* a, const float* c, const int* idx, int n, float* alphaRT, float* colorRT, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(scatter_samples_kernel, GRID1D(n), a, c, idx, n, alphaRT, colorRT);
    return nero_check_launch("nero_scatter_samples");
}

int nero_composite_fwd(const float* alphaRT, const float* colorRT, int R, int T, float* weights, float* rgb, void* stream) {
    if (R == 0) return NERO_OK;
    if (T <= 192) hipLaunchKernelGGL(composite_fwd_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, alphaRT, colorRT, R, T, weights, rgb);
    else hipLaunchKernelGGL(composite_fwd_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, alphaRT, colorRT, R, T, weights, rgb);
    return nero_check_launch("nero_composite_fwd");
}

int nero_composite_bwd(const float* alphaRT, const float* colorRT, const float* weights, const float* d_rgb, int R, int T,
                       float* d_alphaRT, float* d_colorRT, void* stream) {
    if (R == 0) return NERO_OK;
    if (T <= 192) hipLaunchKernelGGL(composite_bwd_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, alphaRT, colorRT, weights, d_rgb, R, T, d_alphaRT, d_colorRT);
    else hipLaunchKernelGGL(composite_bwd_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, alphaRT, colorRT, weights, d_rgb, R, T, d_alphaRT, d_colorRT);
    return nero_check_launch("nero_composite_bwd");
}

int nero_gather_sample_grads(const float* d_alphaRT, const float* d_colorRT, const int* idx, int n, float* d_a, float* d_c, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(gather_sample_grads_kernel, GRID1D(n), d_alphaRT, d_colorRT, idx, n, d_a, d_c);
    return nero_check_launch("nero_gather_sample_grads");
}

}  // extern "C"
