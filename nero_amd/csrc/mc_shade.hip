// mc_shade.hip -- Stage-II Monte-Carlo shading on gfx950: direction sampling (cosine-weighted + GGX importance), light-MLP
// input encodings for hit / miss secondary rays, the microfacet estimator and its hand-derived backward.
// Replaces MCShadingNetwork.shade_mixed / sample_*_directions / get_lights / fresnel / geometry / distribution
// (network/field.py:756-1012) and what autograd derives from them.  The light MLPs themselves run on the MLP-chain engine and
// the secondary rays on the BVH tracer (bvh.hip); this file is the per-direction glue.  'direction' outer light, no human
// lights (the bell material config); the bear variants are rejected by the host driver for now.
//
// Row r = p*D + j  (p = surface point, j = direction; j < Dd diffuse, j >= Dd specular).
// Point record pt[p][32]:  0-2 v, 3-5 n, 6-8 refl, 9 metallic, 10 roughness, 11-13 albedo, 14 NoV, 15-17 x_d, 18-20 y_d,
//                          21-23 x_s, 24-26 y_s, 27 az offset diffuse (rad), 28 az offset specular (rad), 29-31 p
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "ide.h"                                     // IDE_N, the compile-time coefficient table, ide_forward / ide_backward

// IDE helpers live in shade.hip's translation unit; re-declare the small pieces needed here (same formulas, own TU).
namespace {


double fact(int n) { double r = 1.0; for (int i = 2; i <= n; ++i) r *= i; return r; }
double gen_binom(double a, int k) { double p = 1.0; for (int i = 0; i < k; ++i) p *= (a - i); return p / fact(k); }
double sph_coeff(int l, int m, int k) {
    const double al = ((m & 1) ? -1.0 : 1.0) * pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) * gen_binom(0.5 * (l + k + m - 1.0), l);
    return sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * al;
}
int init_tables() {
    static bool done = false;
    if (done) return 0;
    float mat[17 * IDE_N];
    for (int i = 0; i < 17 * IDE_N; ++i) mat[i] = 0.f;
    int i = 0;
    for (int e = 0; e < 5; ++e) {
        const int l = 1 << e;
        for (int m = 0; m <= l; ++m, ++i) {
            for (int k = 0; k <= l - m; ++k) mat[k * IDE_N + i] = (float)sph_coeff(l, m, k);
        }
    }
    if (!ide_tab_equals(mat)) return -1;               // the compile-time table of ide.h must be this libm one, bit for bit
    done = true;
    return 0;
}

// IDE with kappa_inv = 0 (no attenuation): ide_forward<false> / ide_backward<false> of ide.h
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void norm3(const float* a, float* o) {
    const float n = fmaxf(sqrtf(dot3(a, a)), 1e-12f);
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}
__device__ __forceinline__ void ortho3(const float* d, float* o) {            // get_orthogonal_directions, field.py:756-766
    const float o0[3] = {d[1], -d[0], 0.f}, o1[3] = {-d[2], 0.f, d[0]};
    const bool m0 = sqrtf(dot3(o0, o0)) > sqrtf(dot3(o1, o1));
    norm3(m0 ? o0 : o1, o);
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float sat(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float satg(float x) { return (x >= 0.f && x <= 1.f) ? 1.f : 0.f; }   // gradient gate of clamp(x,0,1)

constexpr float TWO_PI = 6.283185307179586f;

// per-point record (field.py:1014-1017, 951-957)
__global__ void mc_point_setup_kernel(const float* __restrict__ pts, const float* __restrict__ view, const float* __restrict__ normals,
                                      const float* __restrict__ mat5, const float* __restrict__ rand_d, const float* __restrict__ rand_s,
                                      int P_, float* __restrict__ pt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P_) return;
    float v[3], n[3];
    norm3(view + p * 3, v);
    norm3(normals + p * 3, n);
    const float nv = dot3(v, n);
    float* o = pt + (size_t)p * 32;
    float refl[3];
    for (int c = 0; c < 3; ++c) { o[c] = v[c]; o[3 + c] = n[c]; refl[c] = nv * n[c] * 2.f - v[c]; o[6 + c] = refl[c]; }
    for (int c = 0; c < 5; ++c) o[9 + c] = mat5[p * 5 + c];
    o[14] = sat(nv);
    float x[3], y[3];
    ortho3(n, x); cross3(n, x, y);
    for (int c = 0; c < 3; ++c) { o[15 + c] = x[c]; o[18 + c] = y[c]; }
    ortho3(refl, x); cross3(refl, x, y);
    for (int c = 0; c < 3; ++c) { o[21 + c] = x[c]; o[24 + c] = y[c]; }
    o[27] = rand_d ? rand_d[p] * TWO_PI : -1.f;       // < 0: no random azimuth (inference)
    o[28] = rand_s ? rand_s[p] * TWO_PI : -1.f;
    for (int c = 0; c < 3; ++c) o[29 + c] = pts[p * 3 + c];
}

struct SpecSample { float cphi, sphi, cost, sint; };

__device__ __forceinline__ void diffuse_dir(const float* q, float az_t, float el, float* w) {      // field.py:768-787
    float az = az_t * TWO_PI;
    if (q[27] >= 0.f) az = fmodf(az + q[27], TWO_PI);
    const float es = sqrtf(el + 1e-7f), cz = sqrtf(1.f - el + 1e-7f);
    const float cx = es * cosf(az), cy = es * sinf(az);
    for (int c = 0; c < 3; ++c) w[c] = cx * q[15 + c] + cy * q[18 + c] + cz * q[3 + c];
}
__device__ __forceinline__ SpecSample specular_dir(const float* q, float az_t, float el, float* w) { // field.py:789-810
    const float a = q[10];
    SpecSample s;
    s.cost = sqrtf((1.0f - el + 1e-6f) / (1.0f + (a * a - 1.0f) * el + 1e-6f) + 1e-6f);
    s.sint = sqrtf(1.f - s.cost * s.cost + 1e-6f);
    float phi = TWO_PI * az_t;
    if (q[28] >= 0.f) phi = fmodf(phi + q[28], TWO_PI);
    s.cphi = cosf(phi); s.sphi = sinf(phi);
    for (int c = 0; c < 3; ++c) w[c] = s.cphi * s.sint * q[21 + c] + s.sphi * s.sint * q[24 + c] + s.cost * q[6 + c];
    return s;
}

// directions + ray origins for the tracer (field.py:859-860: origin = p + 1e-5 w)
__global__ void mc_dirs_kernel(const float* __restrict__ pt, const float* __restrict__ tab_d, const float* __restrict__ tab_s,
                               int P_, int Dd, int Ds, float* __restrict__ dirs, float* __restrict__ orig) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = Dd + Ds;
    if (row >= P_ * D) return;
    const int p = row / D, j = row - p * D;
    const float* q = pt + (size_t)p * 32;
    float w[3];
    if (j < Dd) diffuse_dir(q, tab_d[2 * j], tab_d[2 * j + 1], w);
    else specular_dir(q, tab_s[2 * (j - Dd)], tab_s[2 * (j - Dd) + 1], w);
    for (int c = 0; c < 3; ++c) { dirs[(size_t)row * 3 + c] = w[c]; orig[(size_t)row * 3 + c] = q[29 + c] + w[c] * 1e-5f; }
}

// exit point of the ray (p', w) on the unit sphere, p' = 0.999 p when |p| > 0.999   (field.py:843-849, 390-396)
struct SphereExit { float pp[3], dtx, s, dist, sph[3]; };
__device__ __forceinline__ SphereExit sphere_exit(const float* p, const float* w) {
    SphereExit e;
    const float n = sqrtf(dot3(p, p));
    const float sc = n > 0.999f ? 0.999f : 1.f;
    for (int c = 0; c < 3; ++c) e.pp[c] = p[c] * sc;
    e.dtx = dot3(e.pp, w);
    e.s = sqrtf(e.dtx * e.dtx - dot3(e.pp, e.pp) + 1.f + 1e-6f);
    e.dist = -e.dtx + e.s;
    for (int c = 0; c < 3; ++c) e.sph[c] = e.pp[c] + w[c] * e.dist;
    return e;
}

// miss rows: X[k] = IDE(w, 0) (72) [ | IDE(sphere exit point, 0) (72) when sphere != 0 ]   (predict_outer_lights, field.py:836-854)
__global__ __launch_bounds__(ROW_BLOCK) void mc_encode_miss_kernel(const float* __restrict__ dirs, const int* __restrict__ idx, const float* __restrict__ pt, int D,
                                      int sphere, int n, int n_pad, float* __restrict__ X) {
    __shared__ float stage[ROW_BLOCK * 73];               // rows_put / rows_flush (ide.h): the block's rows leave row-major, coalesced
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int k = row0 + threadIdx.x;
    const bool live = k < n;
    const float z = live ? 1.f : 0.f;                     // rows n .. n_pad-1 are zero rows
    const int ld = sphere ? 144 : 72;
    const int row = idx[live ? k : 0];
    const float* w = dirs + (size_t)row * 3;
    float e[72];
    ide_forward<false>(w[0], w[1], w[2], 0.f, e);
    rows_put<72, 0, 72>(stage, e, z);
    rows_flush<72>(stage, X, ld, 0, row0, n_pad);
    if (sphere) {
        const SphereExit se = sphere_exit(pt + (size_t)(row / D) * 32 + 29, w);
        ide_forward<false>(se.sph[0], se.sph[1], se.sph[2], 0.f, e);
        rows_put<72, 0, 72>(stage, e, z);
        rows_flush<72>(stage, X, ld, 72, row0, n_pad);
    }
}

// human light input for miss rows (get_human_light, field.py:820-834): zero-variance IPE of the hit on the z=0 plane of the
// per-point human frame.  Xh [rows,24], hmask[rows]
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
__global__ void mc_human_encode_kernel(const float* __restrict__ dirs, const int* __restrict__ idx, const float* __restrict__ pt, int D,
                                       const float* __restrict__ poses, int n, int n_pad, float* __restrict__ Xh, float* __restrict__ hmask) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pad) return;
    float* o = Xh + (size_t)k * 24;
    if (k >= n) { for (int c = 0; c < 24; ++c) o[c] = 0.f; hmask[k] = 0.f; return; }
    const int row = idx[k], p = row / D;
    const HumanGeom g = human_geom(poses + (size_t)p * 12, pt + (size_t)p * 32 + 29, dirs + (size_t)row * 3);
    const float mean[2] = {g.ix * 0.3f * g.h, g.iy * 0.3f * g.h};
    float sc = 1.f;
    for (int s = 0; s < 6; ++s) {
        for (int c = 0; c < 2; ++c) { o[2 * s + c] = sinf(mean[c] * sc); o[12 + 2 * s + c] = sinf(mean[c] * sc + 1.5707963267948966f); }
        sc *= 2.f;
    }
    hmask[k] = g.h;
}

// hit rows: X[k] = [PE8(x_hit) (51), IDE(reflect(-w about n_hit), 0) (72), pad]  ld 128; n_hit = -normalize(face normal)
// (get_inner_lights, field.py:812-818; NeROMaterialRenderer.trace flips the normal, renderer.py:722-723)
__device__ __forceinline__ void hit_reflection(const float* w, const float* fn, float* nh, float* vv, float* refl) {
    const float neg[3] = {-fn[0], -fn[1], -fn[2]};
    norm3(neg, nh);                      // trace(): -normal, normalised
    norm3(nh, nh);                       // get_inner_lights normalises again (idempotent)
    const float mv[3] = {-w[0], -w[1], -w[2]};
    norm3(mv, vv);
    const float d = dot3(vv, nh);
    for (int c = 0; c < 3; ++c) refl[c] = d * nh[c] * 2.f - vv[c];
}
__global__ __launch_bounds__(ROW_BLOCK) void mc_encode_hit_kernel(const float* __restrict__ dirs, const float* __restrict__ pos, const float* __restrict__ fnrm,
                                     const int* __restrict__ idx, int n, int n_pad, float* __restrict__ X) {
    __shared__ float stage[ROW_BLOCK * 73];
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int k = row0 + threadIdx.x;
    const bool live = k < n;
    const float z = live ? 1.f : 0.f;
    const int row = idx[live ? k : 0];
    const float* x = pos + (size_t)row * 3;
    // X row = [PE-8(x_hit) 51 | IDE(reflect(-w, n_hit)) 72 | 0 x 5], written as columns 0..63 and 64..127
    float pe[51];
#pragma unroll
    for (int c = 0; c < 3; ++c) pe[c] = x[c];
    {
        float f = 1.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pe[3 + 6 * i + c] = sinf(x[c] * f);
#pragma unroll
            for (int c = 0; c < 3; ++c) pe[3 + 6 * i + 3 + c] = cosf(x[c] * f);
            f *= 2.f;
        }
    }
    float nh[3], vv[3], refl[3], e[72];
    hit_reflection(dirs + (size_t)row * 3, fnrm + (size_t)row * 3, nh, vv, refl);
    ide_forward<false>(refl[0], refl[1], refl[2], 0.f, e);
    rows_put<64, 0, 51>(stage, pe, z);
    {
        float h[13];
#pragma unroll
        for (int c = 0; c < 13; ++c) h[c] = e[c];
        rows_put<64, 51, 13>(stage, h, z);
    }
    rows_flush<64>(stage, X, 128, 0, row0, n_pad);
    {
        float h[59];
#pragma unroll
        for (int c = 0; c < 59; ++c) h[c] = e[13 + c];
        rows_put<64, 0, 59>(stage, h, z);
    }
    rows_zero<64, 59, 5>(stage);
    rows_flush<64>(stage, X, 128, 64, row0, n_pad);
}

// ---------------------------------------------------------------------------------------------------------------------
// estimator (shade_mixed, field.py:950-998).  One wave per point; lane handles directions lane, lane+64, ...
// slot[row] >= 0: miss row index into outer_raw; < 0: hit row index -(slot)-1 into inner_raw.
// ---------------------------------------------------------------------------------------------------------------------
struct Brdf { float H[3], hlen, hov, nol, noh, fc, gv, gl, denv, denl, dg, t, dden, prob, W, N, Q, G, sv, sl, Tv, Tl; };

// geom: 0 = Schlick-GGX product (geometry_schlick, field.py:892-903), 1 = height-correlated Smith (geometry_ggx_smith_correlated, :905-913)
__device__ __forceinline__ void brdf_terms(const float* q, const float* w, bool diffuse, float wd, float ws, int geom, Brdf& b) {
    const float* v = q; const float* n = q + 3;
    const float r = q[10], nov = q[14];
    float h[3] = {v[0] + w[0], v[1] + w[1], v[2] + w[2]};
    b.hlen = fmaxf(sqrtf(dot3(h, h)), 1e-12f);
    for (int c = 0; c < 3; ++c) b.H[c] = h[c] / b.hlen;
    b.hov = sat(dot3(b.H, v));
    b.nol = sat(dot3(n, w));
    b.noh = sat(dot3(n, b.H));
    const float cc = sat(1.f - b.hov);
    b.fc = cc * cc * cc * cc * cc;
    const float k = r * 0.5f;
    b.denv = nov * (1.f - k) + k + 1e-5f; b.denl = b.nol * (1.f - k) + k + 1e-5f;
    b.gv = nov / b.denv; b.gl = b.nol / b.denl;
    const float a2 = r * r;
    if (geom == 0) b.G = b.gv * b.gl;
    else {
        // G = 1 / (1 + f(nov) + f(nol)), f(c) = 0.5 sqrt(1 + a2 (1-c^2)/(c^2+1e-7)) - 0.5   =>   G = 2 / (sv + sl)
        const float cv2 = nov * nov, cl2 = b.nol * b.nol;
        b.Tv = (1.f - cv2) / (cv2 + 1e-7f);
        b.Tl = (1.f - cl2) / (cl2 + 1e-7f);
        b.sv = sqrtf(1.f + a2 * b.Tv);
        b.sl = sqrtf(1.f + a2 * b.Tl);
        b.G = 2.f / (b.sv + b.sl);
    }
    b.t = b.noh * b.noh * (a2 - 1.f) + 1.f;
    b.dden = 3.14159265358979f * b.t * b.t + 1e-4f;
    b.dg = a2 / b.dden;
    b.prob = diffuse ? b.nol / 3.14159265358979f * wd : b.dg * b.noh / (4.f * b.hov + 1e-5f) * ws;
    b.N = b.dg * b.G;
    b.Q = 4.f * nov * b.prob + 1e-5f;
    b.W = b.N / b.Q;
}

struct Lights { const float* outer_raw; const float* inner_raw; const float* human_raw; const float* hmask; float emax, imax; int geom; int n_hum; };

// DEAD rays (round 6).  With the Schlick-GGX geometry term (the reference's default and the only one its configurations use, field.py:702) a
// direction below the shading horizon has NoL = saturate(n.w) = 0, so G = g(NoV) g(0) = 0 and its estimator weight D G / (4 NoV p + 1e-5) is
// EXACTLY zero; the clamp passes no gradient for n.w < 0, so every derivative of the weight is zero too.  Such a ray (always a GGX-sampled
// one: the cosine-weighted diffuse directions lie above the horizon) contributes exactly nothing to any output or gradient of shade_mixed
// (field.py:950-1012) whatever light it would fetch -- and it is the expensive kind: it starts at the surface and runs through the inside of
// the mesh (hundreds of dependent BVH steps; every one of them hits, i.e. it is a row of the inner-light MLP).  nero_mc_dead_rays flags
// them (n.w < -1e-6: a margin that no difference in contraction between two kernels can cross), the tracer skips flagged rays, the split
// gives them the slot below -- neither a miss row nor a hit row -- and the estimator reads L = 0 for them.
constexpr int DEAD_SLOT = -2147483647 - 1;

// L = near * ( miss: outer (1-hw) + hl hw ; hit: inner )      (get_lights, field.py:866-879)
__device__ __forceinline__ void light_value(int s, const Lights& P_, float near, float* L, float* outer, float* hl, float& hw, float& hw_raw) {
    hw = 0.f; hw_raw = 0.f;
    for (int c = 0; c < 3; ++c) { outer[c] = 0.f; hl[c] = 0.f; }
    if (s == DEAD_SLOT) {
        for (int c = 0; c < 3; ++c) L[c] = 0.f;
    } else if (s >= 0) {
        for (int c = 0; c < 3; ++c) outer[c] = expf(fminf(P_.outer_raw[(size_t)s * 4 + c], P_.emax));
        if (P_.human_raw && s < P_.n_hum) {                 // (miss rows [0, n_hum) own a row of the human-light MLP: nero_mc_split_classes)
            const float hm = P_.hmask[s];
            for (int c = 0; c < 3; ++c) hl[c] = expf(fminf(P_.human_raw[(size_t)s * 4 + c], 0.f)) * hm;
            hw_raw = expf(fminf(P_.human_raw[(size_t)s * 4 + 3], 0.f)) * hm;
            hw = sat(hw_raw);
        }
        for (int c = 0; c < 3; ++c) L[c] = (outer[c] * (1.f - hw) + hl[c] * hw) * near;
    } else {
        const int k = -s - 1;
        for (int c = 0; c < 3; ++c) L[c] = expf(fminf(P_.inner_raw[(size_t)k * 4 + c], P_.imax)) * near;
    }
}

__global__ __launch_bounds__(64) void mc_combine_fwd_kernel(const float* __restrict__ pt, const float* __restrict__ dirs,
                                                            const float* __restrict__ depth, const int* __restrict__ slot,
                                                            Lights LP, int P_, int Dd, int Ds,
                                                            float* __restrict__ rgb_lin, float* __restrict__ dl_mean, float* __restrict__ sl_mean, float* __restrict__ spec_lin) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= P_) return;
    const int D = Dd + Ds;
    const float* q = pt + (size_t)p * 32;
    const float wd = (float)Dd / (float)D, ws = (float)Ds / (float)D;
    const float m = q[9];
    float spec[3] = {0, 0, 0}, diff[3] = {0, 0, 0}, dl[3] = {0, 0, 0}, sl[3] = {0, 0, 0};
    for (int j = lane; j < D; j += 64) {
        const size_t row = (size_t)p * D + j;
        const float w[3] = {dirs[row * 3], dirs[row * 3 + 1], dirs[row * 3 + 2]};
        Brdf b;
        brdf_terms(q, w, j < Dd, wd, ws, LP.geom, b);
        float L[3], ou[3], hl[3], hw, hwr;
        light_value(slot[row], LP, depth[row] > 1e-5f ? 1.f : 0.f, L, ou, hl, hw, hwr);
        for (int c = 0; c < 3; ++c) {
            const float F0 = 0.04f * (1.f - m) + m * q[11 + c];
            const float Fr = F0 + (1.f - F0) * b.fc;
            spec[c] += Fr * L[c] * b.W;
            sl[c] += L[c] * b.W;
            if (j < Dd) { diff[c] += q[11 + c] * (1.f - m) * L[c]; dl[c] += L[c]; }
        }
    }
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) {
            spec[c] += __shfl_xor(spec[c], off); diff[c] += __shfl_xor(diff[c], off);
            dl[c] += __shfl_xor(dl[c], off); sl[c] += __shfl_xor(sl[c], off);
        }
    }
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) {
            rgb_lin[p * 3 + c] = diff[c] / (float)Dd + spec[c] / (float)D;
            dl_mean[p * 3 + c] = dl[c] / (float)Dd;
            sl_mean[p * 3 + c] = sl[c] / (float)D;
            if (spec_lin) spec_lin[p * 3 + c] = spec[c] / (float)D;
        }
    }
}

// backward of the estimator: d_rgb_lin [P,3], d_dl_mean [P,3] ->
//   d_outer_raw [n_miss_pad,4], d_inner_raw [n_hit_pad,4] (raw head gradients), d_mat5 [P,5] (metallic, roughness, albedo),
//   d_wspec [P*Ds,3] (gradient w.r.t. the specular directions through the BRDF terms)
__global__ __launch_bounds__(64) void mc_combine_bwd_kernel(const float* __restrict__ pt, const float* __restrict__ dirs,
                                                            const float* __restrict__ depth, const int* __restrict__ slot,
                                                            Lights LP, int P_, int Dd, int Ds,
                                                            const float* __restrict__ d_rgb, const float* __restrict__ d_dl,
                                                            float* __restrict__ d_outer_raw, float* __restrict__ d_inner_raw,
                                                            float* __restrict__ d_human_raw, float* __restrict__ d_mat5, float* __restrict__ d_wspec) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= P_) return;
    const int D = Dd + Ds;
    const float* q = pt + (size_t)p * 32;
    const float* v = q; const float* n = q + 3;
    const float wd = (float)Dd / (float)D, ws = (float)Ds / (float)D;
    const float m = q[9], r = q[10], nov = q[14];
    const float k = r * 0.5f;
    float gs[3], gd[3], gl_[3];
    for (int c = 0; c < 3; ++c) { gs[c] = d_rgb[p * 3 + c] / (float)D; gd[c] = d_rgb[p * 3 + c] / (float)Dd; gl_[c] = d_dl ? d_dl[p * 3 + c] / (float)Dd : 0.f; }
    float dm = 0.f, dr = 0.f, da[3] = {0, 0, 0};
    for (int j = lane; j < D; j += 64) {
        const size_t row = (size_t)p * D + j;
        const bool diffuse = j < Dd;
        const float w[3] = {dirs[row * 3], dirs[row * 3 + 1], dirs[row * 3 + 2]};
        Brdf b;
        brdf_terms(q, w, diffuse, wd, ws, LP.geom, b);
        const float near = depth[row] > 1e-5f ? 1.f : 0.f;
        const int s = slot[row];
        float L[3], ou[3], hl[3], hw, hwr;
        light_value(s, LP, near, L, ou, hl, hw, hwr);
        float dL[3], dW = 0.f, dFc = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float a = q[11 + c];
            const float F0 = 0.04f * (1.f - m) + m * a;
            const float Fr = F0 + (1.f - F0) * b.fc;
            dL[c] = gs[c] * Fr * b.W + (diffuse ? gd[c] * a * (1.f - m) + gl_[c] : 0.f);
            const float dF = gs[c] * L[c] * b.W;
            const float dF0 = dF * (1.f - b.fc);
            dFc += dF * (1.f - F0);
            dW += gs[c] * Fr * L[c];
            dm += dF0 * (a - 0.04f);
            da[c] += dF0 * m;
            if (diffuse) { da[c] += gd[c] * (1.f - m) * L[c]; dm -= gd[c] * a * L[c]; }
        }
        // raw head gradients: L = near * (outer (1-hw) + hl hw) with outer = exp(min(raw, cap)), or near * inner (a dead ray has no row)
        if (s == DEAD_SLOT) {
        } else if (s >= 0) {
            float o4[3], h4[4] = {0.f, 0.f, 0.f, 0.f};
            float dhw = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float dl_ = dL[c] * near;
                o4[c] = LP.outer_raw[(size_t)s * 4 + c] <= LP.emax ? dl_ * (1.f - hw) * ou[c] : 0.f;
                if (LP.human_raw && s < LP.n_hum) {
                    h4[c] = LP.human_raw[(size_t)s * 4 + c] <= 0.f ? dl_ * hw * hl[c] : 0.f;
                    dhw += dl_ * (hl[c] - ou[c]);
                }
            }
            reinterpret_cast<float4*>(d_outer_raw)[s] = make_float4(o4[0], o4[1], o4[2], 0.f);
            if (LP.human_raw && s < LP.n_hum) {
                const float g = (hwr >= 0.f && hwr <= 1.f) ? dhw : 0.f;
                h4[3] = LP.human_raw[(size_t)s * 4 + 3] <= 0.f ? g * hwr : 0.f;
                reinterpret_cast<float4*>(d_human_raw)[s] = make_float4(h4[0], h4[1], h4[2], h4[3]);
            }
        } else {
            const int kk = -s - 1;
            float o4[3];
            for (int c = 0; c < 3; ++c) o4[c] = LP.inner_raw[(size_t)kk * 4 + c] <= LP.imax ? dL[c] * L[c] : 0.f;
            reinterpret_cast<float4*>(d_inner_raw)[kk] = make_float4(o4[0], o4[1], o4[2], 0.f);
        }
        // W = N / Q
        float dDg = dW * b.G / b.Q;
        const float dG = dW * b.dg / b.Q;
        const float dprob = -dW * b.N / (b.Q * b.Q) * 4.f * nov;
        float dnol;
        if (LP.geom == 0) {
            const float dgl = dG * b.gv, dgv = dG * b.gl;
            // g(x) = x / (x(1-k)+k+eps): dg/dk = -x(1-x)/den^2 ; dg/dx = (k+eps)/den^2
            dr += 0.5f * (dgv * (-nov * (1.f - nov) / (b.denv * b.denv)) + dgl * (-b.nol * (1.f - b.nol) / (b.denl * b.denl)));
            dnol = dgl * (k + 1e-5f) / (b.denl * b.denl);
        } else {
            // G = 2/(sv+sl): dG/ds = -G^2/2;  ds/da2 = T/(2s);  dsl/dnol = a2 T'(nol)/(2 sl),  T'(c) = -2c(1+eps)/(c^2+eps)^2
            const float ds = dG * (-0.5f * b.G * b.G);
            const float a2g = r * r;
            dr += ds * (b.Tv / (2.f * b.sv) + b.Tl / (2.f * b.sl)) * 2.f * r;
            const float cl2 = b.nol * b.nol + 1e-7f;
            dnol = ds * a2g * (-2.f * b.nol * (1.f + 1e-7f) / (cl2 * cl2)) / (2.f * b.sl);
        }
        float dnoh = 0.f, dhov = 0.f;
        if (!diffuse) {
            const float e = 4.f * b.hov + 1e-5f;
            dDg += dprob * b.noh / e * ws;
            dnoh += dprob * b.dg / e * ws;
            dhov += -dprob * b.dg * b.noh * 4.f / (e * e) * ws;
        }
        // Dg(noh, r)
        const float a2 = r * r;
        const float dDg_da2 = 1.f / b.dden - a2 * (2.f * 3.14159265358979f * b.t * b.noh * b.noh) / (b.dden * b.dden);
        dr += dDg * dDg_da2 * 2.f * r;
        dnoh += dDg * (-a2 * (2.f * 3.14159265358979f * b.t * 2.f * b.noh * (a2 - 1.f)) / (b.dden * b.dden));
        // Fc = clamp(1-hov,0,1)^5
        const float cc = sat(1.f - b.hov);
        dhov += -5.f * cc * cc * cc * cc * dFc;
        if (!diffuse) {
            // through H = normalize(v + w), hov = sat(H.v), noh = sat(n.H), nol = sat(n.w)
            const float rh = dot3(b.H, v), rn = dot3(n, b.H), rl = dot3(n, w);
            float dH[3], dw[3];
            for (int c = 0; c < 3; ++c) dH[c] = dhov * satg(rh) * v[c] + dnoh * satg(rn) * n[c];
            const float hd = dot3(b.H, dH);
            for (int c = 0; c < 3; ++c) dw[c] = (dH[c] - b.H[c] * hd) / b.hlen + dnol * satg(rl) * n[c];
            const size_t sr = (size_t)p * Ds + (j - Dd);
            for (int c = 0; c < 3; ++c) d_wspec[sr * 3 + c] = dw[c];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        dm += __shfl_xor(dm, off); dr += __shfl_xor(dr, off);
        for (int c = 0; c < 3; ++c) da[c] += __shfl_xor(da[c], off);
    }
    if (lane == 0) {
        float* o = d_mat5 + (size_t)p * 5;
        o[0] = dm; o[1] = dr; o[2] = da[0]; o[3] = da[1]; o[4] = da[2];
    }
}

// specular directions depend on the roughness: add the light-input gradients (IDE backward of the MLP input gradients) to
// d_wspec and fold everything into d_roughness (field.py:794-809).  One wave per point.
__global__ __launch_bounds__(64) void mc_dir_bwd_kernel(const float* __restrict__ pt, const float* __restrict__ dirs,
                                                        const float* __restrict__ fnrm, const int* __restrict__ slot,
                                                        const float* __restrict__ tab_s, const float* __restrict__ dX_miss,
                                                        const float* __restrict__ dX_hit, const float* __restrict__ d_wspec,
                                                        int P_, int Dd, int Ds, float* __restrict__ d_mat5, int sphere,
                                                        const float* __restrict__ dXh, const float* __restrict__ poses, int n_hum) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= P_) return;
    const int D = Dd + Ds;
    const float* q = pt + (size_t)p * 32;
    const float a = q[10];
    float dr = 0.f;
    for (int js = lane; js < Ds; js += 64) {
        const size_t row = (size_t)p * D + Dd + js;
        const float w[3] = {dirs[row * 3], dirs[row * 3 + 1], dirs[row * 3 + 2]};
        float dw[3] = {d_wspec[((size_t)p * Ds + js) * 3], d_wspec[((size_t)p * Ds + js) * 3 + 1], d_wspec[((size_t)p * Ds + js) * 3 + 2]};
        const int s = slot[row];
        float dk_unused = 0.f;
        if (s == DEAD_SLOT) {                                  // no light row: nothing comes back through the light inputs (d_wspec is zero as well)
        } else if (s >= 0) {
            const int ldm = sphere ? 144 : 72;
            const float* gm = dX_miss + (size_t)s * ldm;
            ide_backward<false>(w[0], w[1], w[2], 0.f, [&](int c) { return gm[c]; }, dw[0], dw[1], dw[2], dk_unused);
            if (sphere) {
                // sph = p' + w dist(w), dist = -<p',w> + sqrt(<p',w>^2 - |p'|^2 + 1 + 1e-6)
                const SphereExit se = sphere_exit(q + 29, w);
                float ds[3] = {0.f, 0.f, 0.f};
                ide_backward<false>(se.sph[0], se.sph[1], se.sph[2], 0.f, [&](int c) { return gm[72 + c]; }, ds[0], ds[1], ds[2], dk_unused);
                const float dd = dot3(ds, w) * (se.dtx / se.s - 1.f);
                for (int c = 0; c < 3; ++c) dw[c] += se.dist * ds[c] + dd * se.pp[c];
            }
            if (dXh && s < n_hum) {
                // IPE(mean, 0) of the human-plane hit: mean = 0.3 h inter_xy
                const float* pose = poses + (size_t)p * 12;
                const HumanGeom hg = human_geom(pose, q + 29, w);
                const float* gx = dXh + (size_t)s * 24;
                const float mean[2] = {hg.ix * 0.3f * hg.h, hg.iy * 0.3f * hg.h};
                float dmean[2] = {0.f, 0.f};
                float sc = 1.f;
                for (int t = 0; t < 6; ++t) {
                    for (int c = 0; c < 2; ++c)
                        dmean[c] += sc * (gx[2 * t + c] * cosf(mean[c] * sc) + gx[12 + 2 * t + c] * cosf(mean[c] * sc + 1.5707963267948966f));
                    sc *= 2.f;
                }
                const float dix = 0.3f * hg.h * dmean[0], diy = 0.3f * hg.h * dmean[1];
                const float d_dist = dix * hg.dx + diy * hg.dy;
                const float ddx = hg.dist * dix, ddy = hg.dist * diy;
                const float ddz = hg.hits0 ? d_dist * hg.pz / (hg.dzp * hg.dzp) : 0.f;
                dw[0] += pose[0] * ddx + pose[4] * ddy + pose[8] * ddz;
                dw[1] += pose[1] * ddx + pose[5] * ddy + pose[9] * ddz;
                dw[2] += pose[2] * ddx + pose[6] * ddy + pose[10] * ddz;
            }
        } else {
            const int kk = -s - 1;
            const float* gh = dX_hit + (size_t)kk * 128 + 51;
            float nh[3], vv[3], refl[3];
            hit_reflection(w, fnrm + row * 3, nh, vv, refl);
            float drf[3] = {0.f, 0.f, 0.f};
            ide_backward<false>(refl[0], refl[1], refl[2], 0.f, [&](int c) { return gh[c]; }, drf[0], drf[1], drf[2], dk_unused);
            // refl = 2 (vv.n) n - vv ; vv = normalize(-w)
            const float dn = dot3(drf, nh);
            float dvv[3];
            for (int c = 0; c < 3; ++c) dvv[c] = 2.f * nh[c] * dn - drf[c];
            const float wl = fmaxf(sqrtf(dot3(w, w)), 1e-12f);
            const float pv = dot3(vv, dvv);
            for (int c = 0; c < 3; ++c) dw[c] += -(dvv[c] - vv[c] * pv) / wl;
        }
        // w = sint (cphi x + sphi y) + cost z,  cost(a) = sqrt(A/B + 1e-6), sint = sqrt(1 - cost^2 + 1e-6)
        const float el = tab_s[2 * js + 1];
        float wtmp[3];
        const SpecSample sp = specular_dir(q, tab_s[2 * js], el, wtmp);
        const float A = 1.0f - el + 1e-6f, B = 1.0f + (a * a - 1.0f) * el + 1e-6f;
        const float dcost = (1.f / (2.f * sp.cost)) * (-A / (B * B)) * (2.f * a * el);
        const float dsint = -(sp.cost / sp.sint) * dcost;
        float txy[3];
        for (int c = 0; c < 3; ++c) txy[c] = sp.cphi * q[21 + c] + sp.sphi * q[24 + c];
        dr += dot3(dw, txy) * dsint + dot3(dw, q + 6) * dcost;
    }
    for (int off = 32; off > 0; off >>= 1) dr += __shfl_xor(dr, off);
    if (lane == 0) d_mat5[(size_t)p * 5 + 1] += dr;
}


// hum[row] = 1 when the ray reaches the photographer's region of the camera plane (get_human_light, field.py:819-824: hits & |mean| < 1.5 &
// dist > 0) -- the human-light MLP's output is multiplied by this mask (field.py:829), so every other row of that MLP is exactly dead
__global__ void mc_human_flags_kernel(const float* __restrict__ pt, const float* __restrict__ dirs, const float* __restrict__ poses, int P_, int D,
                                      unsigned char* __restrict__ hum) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= P_ * D) return;
    const int p = row / D;
    const HumanGeom g = human_geom(poses + (size_t)p * 12, pt + (size_t)p * 32 + 29, dirs + (size_t)row * 3);
    hum[row] = g.h != 0.f ? 1 : 0;
}

// dead[row] = 1 for a specular direction below the shading horizon by more than the margin (see DEAD_SLOT above), 0 otherwise
__global__ void mc_dead_rays_kernel(const float* __restrict__ pt, const float* __restrict__ dirs, int P_, int Dd, int Ds,
                                    unsigned char* __restrict__ dead) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = Dd + Ds;
    if (row >= P_ * D) return;
    const int p = row / D, j = row - p * D;
    const float* n = pt + (size_t)p * 32 + 3;
    const float w[3] = {dirs[(size_t)row * 3], dirs[(size_t)row * 3 + 1], dirs[(size_t)row * 3 + 2]};
    dead[row] = (j >= Dd && dot3(n, w) < -1e-6f) ? 1 : 0;
}

// ---- hit / miss split of the P*D secondary rays (get_lights, network/field.py:861-877: lights[miss] = outer(...), lights[hit] =
// inner(...)): ordered compaction of the ray ids by `depth < 10` into two index lists + the slot map the combine kernels read.
// Three launches: per-block ballot counts, one-block exclusive scan, ordered scatter (position = block base + wave base + population
// count of the lower lanes) -- the order torch.nonzero produces, without its host round trip for the sizes of intermediate tensors.
constexpr int SPLIT_BLOCK = 1024;                  // rays per 256-thread block (4 per thread, wave-contiguous chunks of 64)
// Classes: dead rays (flagged by `dead`; neither list), hits (depth < 10, not dead), misses (the rest).  With `hum` the miss list is
// PARTITIONED: the misses that reach the photographer's region (hum[i] != 0) first, in ray order, then the other misses, in ray order -- the
// human-light MLP then runs on the miss rows [0, n_hum) only (its output is multiplied by that mask: field.py:829).  Without `hum` the miss list
// is in ray order (what torch.nonzero yields).  tmp: three per-block count arrays of nb + 1 ints -- hits, dead rays, human misses.
__device__ __forceinline__ bool is_dead(const unsigned char* dead, int i) { return dead != nullptr && dead[i] != 0; }
struct RayClass { bool dead, hit, hmiss; };
__device__ __forceinline__ RayClass ray_class(const float* __restrict__ depth, const unsigned char* __restrict__ dead,
                                              const unsigned char* __restrict__ hum, int i, int n) {
    RayClass c;
    c.dead = i < n && is_dead(dead, i);
    c.hit = i < n && !c.dead && depth[i] < 10.0f;
    c.hmiss = i < n && !c.dead && !c.hit && hum != nullptr && hum[i] != 0;
    return c;
}
__global__ __launch_bounds__(256) void mc_split_count_kernel(const float* __restrict__ depth, const unsigned char* __restrict__ dead,
                                                             const unsigned char* __restrict__ hum, int n, int nb, int* __restrict__ tmp) {
    __shared__ int wsum[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int c[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SPLIT_BLOCK + (wv * 4 + k) * 64 + lane;
        const RayClass rc = ray_class(depth, dead, hum, i, n);
        c[0] += __popcll(__ballot(rc.hit));
        c[1] += __popcll(__ballot(rc.dead));
        c[2] += __popcll(__ballot(rc.hmiss));
    }
    if (lane == 0) for (int a = 0; a < 3; ++a) wsum[a][wv] = c[a];
    __syncthreads();
    if (threadIdx.x < 3) tmp[threadIdx.x * (nb + 1) + blockIdx.x] = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
}
// exclusive scans of the three count arrays in place (one 1024-thread block, any nb); counts = (n_miss, n_hit[, n_hum])
__global__ __launch_bounds__(1024) void mc_split_scan_kernel(int* __restrict__ tmp, int nb, int n, int* __restrict__ counts, int three) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    int totals[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        int* const arr = tmp + a * (nb + 1);
        int s = 0;
        for (int k = 0; k < per; ++k) { const int b = tid * per + k; if (b < nb) s += arr[b]; }
        __syncthreads();
        part[tid] = s;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int base = part[tid] - s;
        for (int k = 0; k < per; ++k) {
            const int b = tid * per + k;
            if (b < nb) { const int c = arr[b]; arr[b] = base; base += c; }
        }
        totals[a] = part[1023];
    }
    if (tid == 1023) {
        counts[0] = n - totals[0] - totals[1]; counts[1] = totals[0];
        tmp[3 * (nb + 1) - 1] = totals[2];                                 // (the scatter kernel's base of the plain misses; the slot behind the third array)
        if (three) counts[2] = totals[2];
    }
}
__global__ __launch_bounds__(256) void mc_split_scatter_kernel(const float* __restrict__ depth, const unsigned char* __restrict__ dead,
                                                               const unsigned char* __restrict__ hum, int n, int nb,
                                                               const int* __restrict__ tmp, int* __restrict__ slot, int* __restrict__ miss_idx,
                                                               int* __restrict__ hit_idx) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int n_hum = tmp[3 * (nb + 1) - 1];
    // hits / dead rays / human misses before this wave's first chunk inside the block: the chunks are wave-contiguous, so count the earlier
    // waves' chunks directly
    int hits_before = tmp[blockIdx.x], dead_before = tmp[nb + 1 + blockIdx.x], hm_before = tmp[2 * (nb + 1) + blockIdx.x];
    for (int c = 0; c < wv * 4; ++c) {
        const int i = blockIdx.x * SPLIT_BLOCK + c * 64 + lane;
        const RayClass rc = ray_class(depth, dead, hum, i, n);
        hits_before += __popcll(__ballot(rc.hit));
        dead_before += __popcll(__ballot(rc.dead));
        hm_before += __popcll(__ballot(rc.hmiss));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = blockIdx.x * SPLIT_BLOCK + (wv * 4 + k) * 64;
        const int i = i0 + lane;
        const RayClass rc = ray_class(depth, dead, hum, i, n);
        const unsigned long long mh = __ballot(rc.hit), md = __ballot(rc.dead), mm = __ballot(rc.hmiss);
        if (i < n) {
            if (rc.dead) slot[i] = DEAD_SLOT;
            else if (rc.hit) { const int q = hits_before + __popcll(mh & lower); hit_idx[q] = i; slot[i] = -q - 1; }
            else if (rc.hmiss) { const int q = hm_before + __popcll(mm & lower); miss_idx[q] = i; slot[i] = q; }
            else {                                                       // plain misses before i = i - hits - dead rays - human misses before i
                const int q = n_hum + (i0 - hits_before - dead_before - hm_before) + (lane - __popcll(mh & lower) - __popcll(md & lower) - __popcll(mm & lower));
                miss_idx[q] = i; slot[i] = q;
            }
        }
        hits_before += __popcll(mh);
        dead_before += __popcll(md);
        hm_before += __popcll(mm);
    }
}
}  // namespace

#define GRID1D(n) dim3(((n) + 127) / 128), dim3(128), 0, (hipStream_t)stream
#define CHECK_T() do { if (init_tables() != 0) return nero_fail(NERO_ERR_LAUNCH, "IDE coefficient table of ide.h differs from the libm one"); } while (0)

extern "C" {

int nero_mc_point_setup(const float* pts, const float* view, const float* normals, const float* mat5, const float* rand_d,
                        const float* rand_s, int P, float* pt, void* stream) {
    if (P == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_point_setup_kernel, GRID1D(P), pts, view, normals, mat5, rand_d, rand_s, P, pt);
    return nero_check_launch("nero_mc_point_setup");
}

int nero_mc_dirs(const float* pt, const float* tab_d, const float* tab_s, int P, int Dd, int Ds, float* dirs, float* origins, void* stream) {
    if (P * (Dd + Ds) == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_dirs_kernel, GRID1D(P * (Dd + Ds)), pt, tab_d, tab_s, P, Dd, Ds, dirs, origins);
    return nero_check_launch("nero_mc_dirs");
}

int nero_mc_encode_miss(const float* dirs, const int* idx, const float* pt, int D, int sphere, int n, float* X, void* stream) {
    CHECK_T();
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_encode_miss_kernel, dim3((n_pad + ROW_BLOCK - 1) / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, dirs, idx, pt, D, sphere, n, n_pad, X);
    return nero_check_launch("nero_mc_encode_miss");
}

int nero_mc_human_encode(const float* dirs, const int* idx, const float* pt, int D, const float* poses, int n, float* Xh, float* hmask,
                         void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_human_encode_kernel, GRID1D(n_pad), dirs, idx, pt, D, poses, n, n_pad, Xh, hmask);
    return nero_check_launch("nero_mc_human_encode");
}

int nero_mc_encode_hit(const float* dirs, const float* pos, const float* face_normals, const int* idx, int n, float* X, void* stream) {
    CHECK_T();
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_encode_hit_kernel, dim3((n_pad + ROW_BLOCK - 1) / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, dirs, pos, face_normals, idx, n, n_pad, X);
    return nero_check_launch("nero_mc_encode_hit");
}

// (_h: the human-light MLP owns a row for the miss rows [0, n_hum) only -- nero_mc_split_classes; the plain entry points: for every miss row)
int nero_mc_combine_fwd_h(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                          const float* inner_raw, const float* human_raw, const float* hmask, int n_hum, float exp_max, float inner_exp_max, int P,
                          int Dd, int Ds, int geometry_type, float* rgb_lin, float* dl_mean, float* sl_mean, float* spec_lin, void* stream) {
    if (P == 0) return NERO_OK;
    if (geometry_type != 0 && geometry_type != 1) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_mc_combine_fwd: unknown geometry_type");
    Lights LP{outer_raw, inner_raw, human_raw, hmask, exp_max, inner_exp_max, geometry_type, n_hum};
    hipLaunchKernelGGL(mc_combine_fwd_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, pt, dirs, depth, slot, LP, P, Dd, Ds, rgb_lin,
                       dl_mean, sl_mean, spec_lin);
    return nero_check_launch("nero_mc_combine_fwd");
}
int nero_mc_combine_fwd(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                        const float* inner_raw, const float* human_raw, const float* hmask, float exp_max, float inner_exp_max, int P,
                        int Dd, int Ds, int geometry_type, float* rgb_lin, float* dl_mean, float* sl_mean, float* spec_lin, void* stream) {
    return nero_mc_combine_fwd_h(pt, dirs, depth, slot, outer_raw, inner_raw, human_raw, hmask, 0x7fffffff, exp_max, inner_exp_max, P, Dd, Ds,
                                 geometry_type, rgb_lin, dl_mean, sl_mean, spec_lin, stream);
}

int nero_mc_combine_bwd_h(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                          const float* inner_raw, const float* human_raw, const float* hmask, int n_hum, float exp_max, float inner_exp_max, int P,
                          int Dd, int Ds, int geometry_type, const float* d_rgb, const float* d_dl, float* d_outer_raw, float* d_inner_raw,
                          float* d_human_raw, float* d_mat5, float* d_wspec, void* stream) {
    if (P == 0) return NERO_OK;
    if (geometry_type != 0 && geometry_type != 1) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_mc_combine_bwd: unknown geometry_type");
    Lights LP{outer_raw, inner_raw, human_raw, hmask, exp_max, inner_exp_max, geometry_type, n_hum};
    hipLaunchKernelGGL(mc_combine_bwd_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, pt, dirs, depth, slot, LP, P, Dd, Ds, d_rgb,
                       d_dl, d_outer_raw, d_inner_raw, d_human_raw, d_mat5, d_wspec);
    return nero_check_launch("nero_mc_combine_bwd");
}
int nero_mc_combine_bwd(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                        const float* inner_raw, const float* human_raw, const float* hmask, float exp_max, float inner_exp_max, int P,
                        int Dd, int Ds, int geometry_type, const float* d_rgb, const float* d_dl, float* d_outer_raw, float* d_inner_raw,
                        float* d_human_raw, float* d_mat5, float* d_wspec, void* stream) {
    return nero_mc_combine_bwd_h(pt, dirs, depth, slot, outer_raw, inner_raw, human_raw, hmask, 0x7fffffff, exp_max, inner_exp_max, P, Dd, Ds,
                                 geometry_type, d_rgb, d_dl, d_outer_raw, d_inner_raw, d_human_raw, d_mat5, d_wspec, stream);
}

int nero_mc_dir_bwd_h(const float* pt, const float* dirs, const float* face_normals, const int* slot, const float* tab_s,
                      const float* dX_miss, const float* dX_hit, const float* d_wspec, int P, int Dd, int Ds, float* d_mat5, int sphere,
                      const float* dXh, const float* poses, int n_hum, void* stream) {
    CHECK_T();
    if (P == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_dir_bwd_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, pt, dirs, face_normals, slot, tab_s, dX_miss, dX_hit,
                       d_wspec, P, Dd, Ds, d_mat5, sphere, dXh, poses, n_hum);
    return nero_check_launch("nero_mc_dir_bwd");
}
int nero_mc_dir_bwd(const float* pt, const float* dirs, const float* face_normals, const int* slot, const float* tab_s,
                    const float* dX_miss, const float* dX_hit, const float* d_wspec, int P, int Dd, int Ds, float* d_mat5, int sphere,
                    const float* dXh, const float* poses, void* stream) {
    return nero_mc_dir_bwd_h(pt, dirs, face_normals, slot, tab_s, dX_miss, dX_hit, d_wspec, P, Dd, Ds, d_mat5, sphere, dXh, poses, 0x7fffffff, stream);
}


int nero_mc_split_tmp_ints(int n) { return 3 * ((n + SPLIT_BLOCK - 1) / SPLIT_BLOCK + 1); }

int nero_mc_dead_rays(const float* pt, const float* dirs, int P, int Dd, int Ds, int geometry_type, unsigned char* dead, void* stream) {
    if (!pt || !dirs || !dead || P < 0 || Dd < 0 || Ds < 0) return nero_fail(NERO_ERR_ARG, "nero_mc_dead_rays: bad argument");
    const int N = P * (Dd + Ds);
    if (N == 0) return NERO_OK;
    if (geometry_type != 0) {                       // the height-correlated Smith term keeps a small non-zero weight at NoL = 0 (field.py:905-913): nothing is dead
        if (hipMemsetAsync(dead, 0, (size_t)N, (hipStream_t)stream) != hipSuccess) return nero_fail(NERO_ERR_LAUNCH, "nero_mc_dead_rays: memset failed");
        return NERO_OK;
    }
    hipLaunchKernelGGL(mc_dead_rays_kernel, GRID1D(N), pt, dirs, P, Dd, Ds, dead);
    return nero_check_launch("nero_mc_dead_rays");
}

int nero_mc_human_flags(const float* pt, const float* dirs, const float* poses, int P, int D, unsigned char* hum, void* stream) {
    if (!pt || !dirs || !poses || !hum || P < 0 || D < 1) return nero_fail(NERO_ERR_ARG, "nero_mc_human_flags: bad argument");
    if (P == 0) return NERO_OK;
    hipLaunchKernelGGL(mc_human_flags_kernel, GRID1D(P * D), pt, dirs, poses, P, D, hum);
    return nero_check_launch("nero_mc_human_flags");
}

static int split_impl(const float* depth, const unsigned char* dead, const unsigned char* hum, int n, int* slot, int* miss_idx, int* hit_idx,
                      int* counts, int three, int* tmp, void* stream) {
    if (!depth || !slot || !miss_idx || !hit_idx || !counts || !tmp || n < 0) return nero_fail(NERO_ERR_ARG, "nero_mc_split: bad argument");
    if (n == 0) { (void)hipMemsetAsync(counts, 0, three ? 12 : 8, (hipStream_t)stream); return NERO_OK; }
    const int nb = (n + SPLIT_BLOCK - 1) / SPLIT_BLOCK;
    hipLaunchKernelGGL(mc_split_count_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, depth, dead, hum, n, nb, tmp);
    hipLaunchKernelGGL(mc_split_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tmp, nb, n, counts, three);
    hipLaunchKernelGGL(mc_split_scatter_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, depth, dead, hum, n, nb, tmp, slot, miss_idx, hit_idx);
    return nero_check_launch("nero_mc_split");
}

int nero_mc_split_classes(const float* depth, const unsigned char* dead, const unsigned char* hum, int n, int* slot, int* miss_idx, int* hit_idx,
                          int* counts3, int* tmp, void* stream) {
    return split_impl(depth, dead, hum, n, slot, miss_idx, hit_idx, counts3, 1, tmp, stream);
}

int nero_mc_split_dead(const float* depth, const unsigned char* dead, int n, int* slot, int* miss_idx, int* hit_idx, int* counts, int* tmp,
                       void* stream) {
    return split_impl(depth, dead, nullptr, n, slot, miss_idx, hit_idx, counts, 0, tmp, stream);
}

int nero_mc_split(const float* depth, int n, int* slot, int* miss_idx, int* hit_idx, int* counts, int* tmp, void* stream) {
    return split_impl(depth, nullptr, nullptr, n, slot, miss_idx, hit_idx, counts, 0, tmp, stream);
}

}  // extern "C"
