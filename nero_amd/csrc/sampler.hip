// sampler.hip -- hierarchical ray sampling for Stage I (gfx950): coarse z, NeuS section weights, deterministic
// inverse-CDF importance samples, sorted merge, background z, and the inner/outer sample compaction.
// Replaces NeROShapeRenderer.sample_ray / upsample / cat_z_vals (network/renderer.py:355-443) and sample_pdf
// (network/field.py:399-429).  Scans run in the order the oracle states (sequential, float64 running value rounded to
// float32 per element) so that integer outputs (searchsorted indices, merge permutation) are bit-exact under teacher forcing.
// One thread owns one ray: the per-ray state is <= 128 floats and the whole stage is a few hundred microseconds; the
// SDF evaluations between rounds (the real cost) go through the MLP-chain kernel.
#include <hip/hip_runtime.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "rows.h"

#pragma clang fp contract(off)      // keep a*b+c un-fused: the oracle's float32 steps are stated without FMA

namespace {

constexpr int MAXS = 160;   // max z-values per ray handled by the per-thread buffers

// torch.linspace(start, end, steps)[i] in float32 (ATen: symmetric evaluation around the midpoint)
__device__ __forceinline__ float linspace_f32(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    return i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// z[r, i] = near + (far-near) * lin(0,1,n)[i] (+ (rand1-0.5)*2/n)          (renderer.py:411-417)
__global__ void coarse_z_kernel(const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ rand1,
                                int R, int n, float* __restrict__ z, int ldz) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * n) return;
    const int r = idx / n, i = idx - r * n;
    float v = near[r] + (far[r] - near[r]) * linspace_f32(0.f, 1.f, n, i);
    if (rand1) v = v + (rand1[r] - 0.5f) * 2.0f / (float)n;
    z[(size_t)r * ldz + i] = v;
}

// z_bg[r, j] = far / flip(zo)[j] + 1/n_bg, zo = lin(1e-3, 1-1/(n_bg+1), n_bg) optionally stratified-jittered (renderer.py:413-425)
__global__ void background_z_kernel(const float* __restrict__ far, const float* __restrict__ rand_bg, int R, int nb,
                                    float* __restrict__ z, int ldz, int col0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * nb) return;
    const int r = idx / nb, j = idx - r * nb;
    const int k = nb - 1 - j;                          // flip
    const float end = 1.0f - 1.0f / ((float)nb + 1.0f);
    float zo = linspace_f32(1e-3f, end, nb, k);
    if (rand_bg) {
        const float prev = k > 0 ? linspace_f32(1e-3f, end, nb, k - 1) : zo;
        const float next = k < nb - 1 ? linspace_f32(1e-3f, end, nb, k + 1) : zo;
        const float lower = k > 0 ? 0.5f * (zo + prev) : zo;
        const float upper = k < nb - 1 ? 0.5f * (next + zo) : zo;
        zo = lower + (upper - lower) * rand_bg[(size_t)r * nb + k];
    }
    z[(size_t)r * ldz + col0 + j] = far[r] / zo + 1.0f / (float)nb;
}


// this thread's PE-6 row [x(3), sin / cos of 2^j x, j < 6 (36), 0] into its row of a staged 40-column piece (pitch 41); live = 0: a zero row
__device__ __forceinline__ void pe6_stage_row(float* __restrict__ mine, const float (&p)[3], float z) {
#pragma unroll
    for (int c = 0; c < 3; ++c) mine[c] = p[c] * z;
    float f = 1.f;
    for (int j = 0; j < 6; ++j, f *= 2.f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) mine[3 + 6 * j + c] = sinf(p[c] * f) * z;
#pragma unroll
        for (int c = 0; c < 3; ++c) mine[6 + 6 * j + c] = cosf(p[c] * f) * z;
    }
    mine[39] = 0.f;
}

// PE-6 rows (ld 40) of the points o + d * z[r, col0 + j], row = r * ncols + j: a thread per ROW (every lane on the same column: no
// sine / cosine divergence), the block's rows leave through LDS (rows.h)
__global__ __launch_bounds__(ROW_BLOCK) void ray_points_pe_row_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                                                      int ldz, int col0, int ncols, int R, int n_pad, float* __restrict__ pe) {
    __shared__ float stage[ROW_BLOCK * 41];
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int row = row0 + threadIdx.x;
    const bool live = row < R * ncols;
    const int rr = live ? row : 0;
    const int r = rr / ncols, j = rr - r * ncols;
    const float t = z[(size_t)r * ldz + col0 + j];
    const float p[3] = {o[r * 3] + d[r * 3] * t, o[r * 3 + 1] + d[r * 3 + 1] * t, o[r * 3 + 2] + d[r * 3 + 2] * t};
    pe6_stage_row(stage + threadIdx.x * 41, p, live ? 1.f : 0.f);
    rows_flush<40>(stage, pe, 40, 0, row0, n_pad);
}

// inner rows: x4 = the gathered point, pe = its PE-6 row
__global__ __launch_bounds__(ROW_BLOCK) void gather_inner_row_kernel(const float* __restrict__ pts4, const int* __restrict__ idx, int n, int n_pad,
                                                                     float* __restrict__ x4, float* __restrict__ pe) {
    __shared__ float stage[ROW_BLOCK * 41];
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int k = row0 + threadIdx.x;
    const bool live = k < n;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) x = reinterpret_cast<const float4*>(pts4)[idx[k]];
    if (k < n_pad) reinterpret_cast<float4*>(x4)[k] = x;
    const float p[3] = {x.x, x.y, x.z};
    pe6_stage_row(stage + threadIdx.x * 41, p, live ? 1.f : 0.f);
    rows_flush<40>(stage, pe, 40, 0, row0, n_pad);
}

// outer rows: PE-10 of [p/|p|, 1/|p|] (84 + 4 pad) -> pe88, PE-4 of the view direction (27 + 5 pad) -> pev32, dist.  One thread per ROW,
// the rows of a block leave through LDS (rows.h).  The column-quad form of round 3 (a thread per 4 columns) stored coalesced too, but a
// wave mixed sine and cosine columns -- both branches ran for every lane -- and each of a row's 30 threads renormalised the point:
// 170 us per step.  Here every lane evaluates the same column at the same time.
__global__ __launch_bounds__(ROW_BLOCK) void gather_outer_row_kernel(const float* __restrict__ pts4, const float* __restrict__ d, const int* __restrict__ idx,
                                                                     int T, int n, int n_pad, float* __restrict__ pe88, float* __restrict__ pev32,
                                                                     float* __restrict__ dist) {
    __shared__ float stage[ROW_BLOCK * 45];
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int k = row0 + threadIdx.x;
    const bool live = k < n;
    const float z = live ? 1.f : 0.f;                     // rows n .. n_pad-1 are zero rows
    const int s = idx[live ? k : 0];
    const float4 x = reinterpret_cast<const float4*>(pts4)[s];
    if (k < n_pad) dist[k] = live ? x.w : 0.f;
    float* mine = stage + threadIdx.x * 45;               // this thread's row of the staged <= 44-column piece (pitch 45)
    {
        const float nrm = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z);
        const float p[4] = {x.x / nrm, x.y / nrm, x.z / nrm, 1.0f / nrm};
        // columns 0..43: p (4), frequencies 0..4 (sin 4, cos 4 each); columns 44..87: frequencies 5..9, 4 x zero padding
#pragma unroll
        for (int c = 0; c < 4; ++c) mine[c] = p[c] * z;
        float f = 1.f;
        for (int j = 0; j < 5; ++j, f *= 2.f) {
#pragma unroll
            for (int c = 0; c < 4; ++c) mine[4 + 8 * j + c] = sinf(p[c] * f) * z;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine[8 + 8 * j + c] = cosf(p[c] * f) * z;
        }
        rows_flush<44>(stage, pe88, 88, 0, row0, n_pad);
        for (int j = 0; j < 5; ++j, f *= 2.f) {
#pragma unroll
            for (int c = 0; c < 4; ++c) mine[8 * j + c] = sinf(p[c] * f) * z;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine[4 + 8 * j + c] = cosf(p[c] * f) * z;
        }
#pragma unroll
        for (int c = 40; c < 44; ++c) mine[c] = 0.f;
        rows_flush<44>(stage, pe88, 88, 44, row0, n_pad);
    }
    {
        const int r = s / T;
        const float dxr = d[r * 3], dyr = d[r * 3 + 1], dzr = d[r * 3 + 2];
        const float dn = fmaxf(sqrtf(dxr * dxr + dyr * dyr + dzr * dzr), 1e-12f);
        const float w[3] = {-(dxr / dn), -(dyr / dn), -(dzr / dn)};
        // 27 columns (w, 4 frequencies) + 5 x zero padding, staged at the same pitch
#pragma unroll
        for (int c = 0; c < 3; ++c) mine[c] = w[c] * z;
        float f = 1.f;
        for (int j = 0; j < 4; ++j, f *= 2.f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) mine[3 + 6 * j + c] = sinf(w[c] * f) * z;
#pragma unroll
            for (int c = 0; c < 3; ++c) mine[6 + 6 * j + c] = cosf(w[c] * f) * z;
        }
#pragma unroll
        for (int c = 27; c < 32; ++c) mine[c] = 0.f;
        rows_flush_cols<44>(stage, pev32, 32, 0, 32, row0, n_pad);
    }
}

// Per-ray working arrays live in LDS as columns of a [index][64 lanes] table (lane-consecutive -> conflict-free): the scans
// below are dependent chains of ~100 steps per ray, and private (scratch) arrays would make every step a trip through the
// global-memory path.  Storage only -- the arithmetic and its order are unchanged.
struct Col {
    float* p;
    __device__ __forceinline__ float& operator[](int i) const { return p[i << 6]; }
};
__device__ __forceinline__ Col ray_col(float* smem, int table, int n_max) { return Col{smem + (size_t)table * n_max * 64 + (threadIdx.x & 63)}; }
inline int ray_tables_bytes(int tables, int n_max) { return tables * n_max * 64 * (int)sizeof(float); }

// deterministic inverse-CDF sampling of `m` values from bins z[0..n) with weights w[0..n-1)   (field.py:399-429, det=True)
__device__ void sample_pdf_det(const Col& zb, const Col& w, const Col& cdf, int n, int m, float* out, int* inds_out) {
    // scans: float64 running value, rounded to float32 only where an element is stored (torch-CPU cumsum semantics)
    double acc = 0.0;
    for (int i = 0; i < n - 1; ++i) acc += (double)(w[i] + 1e-5f);
    const float norm = (float)acc;
    cdf[0] = 0.f;
    acc = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        const float pdf = (w[i] + 1e-5f) / norm;
        acc += (double)pdf;
        cdf[i + 1] = (float)acc;
    }
    const float u0 = 0.5f / (float)m, u1 = 1.0f - 0.5f / (float)m;
    int idx = 0;
    for (int j = 0; j < m; ++j) {
        const float u = linspace_f32(u0, u1, m, j);
        while (idx < n && cdf[idx] <= u) ++idx;        // searchsorted(right=True): first index with cdf > u
        const int below = idx - 1 > 0 ? idx - 1 : 0;
        const int above = idx < n - 1 ? idx : n - 1;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - cdf[below]) / denom;
        out[j] = zb[below] + t * (zb[above] - zb[below]);
        if (inds_out) inds_out[j] = idx;
    }
}

// one up-sampling round: NeuS section weights from (z, sdf) -> m new z          (renderer.py:355-385)
__global__ void upsample_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int ldz,
                                const float* __restrict__ sdf, int lds, int n, const float* __restrict__ variance, float inv_s_cap,
                                int m, int R, float* __restrict__ z_new, float* __restrict__ w_out, int* __restrict__ inds_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    // renderer.py:434-438: inv_s = min(exp(10 v), 64*2^i) (clip_sample_variance) or the fixed 64*2^i (variance == NULL)
    const float inv_s = variance ? fminf(expf(variance[0] * 10.0f), inv_s_cap) : inv_s_cap;
    extern __shared__ float ray_smem[];
    const Col zl = ray_col(ray_smem, 0, n), w = ray_col(ray_smem, 1, n), cdf = ray_col(ray_smem, 2, n);
    const float ox = o[r * 3], oy = o[r * 3 + 1], oz = o[r * 3 + 2];
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz_ = d[r * 3 + 2];
    for (int i = 0; i < n; ++i) zl[i] = z[(size_t)r * ldz + i];
    double T = 1.0;
    float prev_cos = 0.f;
    float px = ox + dx * zl[0], py = oy + dy * zl[0], pz = oz + dz_ * zl[0];
    float rad_prev = sqrtf(px * px + py * py + pz * pz);
    float s_prev = sdf[(size_t)r * lds];
    for (int i = 0; i < n - 1; ++i) {
        px = ox + dx * zl[i + 1]; py = oy + dy * zl[i + 1]; pz = oz + dz_ * zl[i + 1];
        const float rad_next = sqrtf(px * px + py * py + pz * pz);
        const float s_next = sdf[(size_t)r * lds + i + 1];
        const float inside = (rad_prev < 1.0f || rad_next < 1.0f) ? 1.f : 0.f;
        const float dist = zl[i + 1] - zl[i];
        const float mid = (s_prev + s_next) * 0.5f;
        const float cosv = (s_next - s_prev) / (dist + 1e-5f);
        float c = fminf(prev_cos, cosv);
        c = fminf(fmaxf(c, -1e3f), 0.f) * inside;
        prev_cos = cosv;
        const float pe_ = mid - c * dist * 0.5f, ne_ = mid + c * dist * 0.5f;
        const float pc = sigmoid_f(pe_ * inv_s), nc = sigmoid_f(ne_ * inv_s);
        const float alpha = (pc - nc + 1e-5f) / (pc + 1e-5f);
        w[i] = alpha * (float)T;
        T *= (double)(1.0f - alpha + 1e-7f);
        rad_prev = rad_next;
        s_prev = s_next;
    }
    if (w_out) for (int i = 0; i < n - 1; ++i) w_out[(size_t)r * (n - 1) + i] = w[i];
    float zn[32];
    int ind[32];
    sample_pdf_det(zl, w, cdf, n, m, zn, inds_out ? ind : nullptr);
    for (int j = 0; j < m; ++j) {
        z_new[(size_t)r * m + j] = zn[j];
        if (inds_out) inds_out[(size_t)r * m + j] = ind[j];
    }
}

// standalone sample_pdf (tests / occ-loss march): bins [R,n], weights [R,n-1] -> samples [R,m]
__global__ void sample_pdf_kernel(const float* __restrict__ bins, int ldb, const float* __restrict__ w, int ldw, int n, int m, int R,
                                  float* __restrict__ out, int* __restrict__ inds_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    extern __shared__ float ray_smem[];
    const Col zl = ray_col(ray_smem, 0, n), wl = ray_col(ray_smem, 1, n), cdf = ray_col(ray_smem, 2, n);
    float zn[32];
    int ind[32];
    for (int i = 0; i < n; ++i) zl[i] = bins[(size_t)r * ldb + i];
    for (int i = 0; i < n - 1; ++i) wl[i] = w[(size_t)r * ldw + i];
    sample_pdf_det(zl, wl, cdf, n, m, zn, inds_out ? ind : nullptr);
    for (int j = 0; j < m; ++j) {
        out[(size_t)r * m + j] = zn[j];
        if (inds_out) inds_out[(size_t)r * m + j] = ind[j];
    }
}

// stable merge of the sorted lists z[r, 0..n) and z_new[r, 0..m) (ties: z first), in place in z; sdf permuted alike
// (renderer.py:387-401: cat + sort + gather)
__global__ void merge_sorted_kernel(float* __restrict__ z, int ldz, int n, float* __restrict__ sdf, int lds,
                                    const float* __restrict__ z_new, int m, const float* __restrict__ sdf_new, int ldsn,
                                    int R, int* __restrict__ index_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    extern __shared__ float ray_smem[];
    const Col za = ray_col(ray_smem, 0, n), sa = ray_col(ray_smem, 1, n);
    const bool hs = sdf != nullptr && sdf_new != nullptr;
    for (int i = 0; i < n; ++i) { za[i] = z[(size_t)r * ldz + i]; if (hs) sa[i] = sdf[(size_t)r * lds + i]; }
    int i = 0, j = 0;
    for (int k = 0; k < n + m; ++k) {
        const float zn = j < m ? z_new[(size_t)r * m + j] : 0.f;
        const bool take_old = (j >= m) || (i < n && za[i] <= zn);
        if (take_old) {
            z[(size_t)r * ldz + k] = za[i];
            if (hs) sdf[(size_t)r * lds + k] = sa[i];
            if (index_out) index_out[(size_t)r * (n + m) + k] = i;
            ++i;
        } else {
            z[(size_t)r * ldz + k] = zn;
            if (hs) sdf[(size_t)r * lds + k] = sdf_new[((size_t)r * m + j) * ldsn];
            if (index_out) index_out[(size_t)r * (n + m) + k] = n + j;
            ++j;
        }
    }
}


// ---- wave-per-ray versions (one wavefront owns one ray; lane i holds samples i and i + 64) ------------------------------------------
// The per-sample arithmetic runs lane-parallel with coalesced row loads; the two scans whose ORDER is part of the contract (the
// float64 running transmittance / cumulative sums, rounded to float32 where stored) stay sequential: a wave-uniform loop reads the
// i-th element with v_readlane and every lane advances the same running value, lane i keeping what the serial loop would have stored
// -- the identical operation sequence, hence the identical bits (tests: teacher-forced `inds` / merge permutation torch.equal).
// searchsorted and the merge ranks are ballot population counts.  4096 rays = 4096 waves (16 per CU) instead of 64.
__device__ __forceinline__ float lane_elem(float e0, float e1, int i) {            // element i of the (i, i + 64) lane layout, wave-uniform
    return i < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e0), i)) : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1), i - 64));
}
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// cdf / inverse-CDF part shared by upsample and sample_pdf: w (n-1 weights in the lane layout) -> m samples.  zs / cs: wave-private
// LDS rows of 128 floats holding z and (on return) the cdf.
__device__ __forceinline__ void wave_sample_pdf(const float* zs, float* cs, float w0, float w1, int n, int m, int lane, float& out, int& ind) {
    double acc = 0.0;
    for (int i = 0; i < n - 1; ++i) acc += (double)(lane_elem(w0, w1, i) + 1e-5f);
    const float norm = (float)acc;
    const float p0 = (w0 + 1e-5f) / norm, p1 = (w1 + 1e-5f) / norm;
    acc = 0.0;
    float c0 = 0.f, c1 = 0.f;                          // cdf[lane], cdf[lane + 64]; cdf[0] = 0
    for (int i = 0; i < n - 1; ++i) {
        acc += (double)lane_elem(p0, p1, i);
        const float v = (float)acc;
        if (i + 1 == lane) c0 = v;
        if (i + 1 == lane + 64) c1 = v;
    }
    cs[lane] = c0;
    cs[lane + 64] = c1;
    __builtin_amdgcn_wave_barrier();
    const float u0 = 0.5f / (float)m, u1 = 1.0f - 0.5f / (float)m;
    int idx = 0;
    for (int j = 0; j < m; ++j) {                      // searchsorted(right=True) on the non-decreasing cdf = #{k < n: cdf[k] <= u}
        const float u = linspace_f32(u0, u1, m, j);
        const int cnt = popc64(__ballot(lane < n && c0 <= u)) + popc64(__ballot(lane + 64 < n && c1 <= u));
        if (j == lane) idx = cnt;
    }
    out = 0.f;
    ind = idx;
    if (lane < m) {
        const float u = linspace_f32(u0, u1, m, lane);
        const int below = idx - 1 > 0 ? idx - 1 : 0;
        const int above = idx < n - 1 ? idx : n - 1;
        float denom = cs[above] - cs[below];
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - cs[below]) / denom;
        out = zs[below] + t * (zs[above] - zs[below]);
    }
}

// 4 rays per 256-thread workgroup
__global__ __launch_bounds__(256) void upsample_wave_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int ldz,
                                                            const float* __restrict__ sdf, int lds, int n, const float* __restrict__ variance,
                                                            float inv_s_cap, int m, int R, float* __restrict__ z_new, float* __restrict__ w_out,
                                                            int* __restrict__ inds_out) {
    __shared__ float sm[4][4][128];                    // per wave: z, sdf, radius / cdf, cos
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    float* zs = sm[wv][0]; float* ss = sm[wv][1]; float* rs = sm[wv][2]; float* cs = sm[wv][3];
    const float inv_s = variance ? fminf(expf(variance[0] * 10.0f), inv_s_cap) : inv_s_cap;
    const float ox = o[r * 3], oy = o[r * 3 + 1], oz = o[r * 3 + 2];
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz_ = d[r * 3 + 2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = lane + 64 * e;
        float zv = 0.f, sv = 0.f, rad = 0.f;
        if (i < n) {
            zv = z[(size_t)r * ldz + i];
            sv = sdf[(size_t)r * lds + i];
            const float px = ox + dx * zv, py = oy + dy * zv, pz = oz + dz_ * zv;
            rad = sqrtf(px * px + py * py + pz * pz);
        }
        zs[i] = zv; ss[i] = sv; rs[i] = rad;
    }
    __builtin_amdgcn_wave_barrier();
    float alpha[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {                      // section i: samples i, i + 1
        const int i = lane + 64 * e;
        float cosv = 0.f;
        if (i < n - 1) cosv = (ss[i + 1] - ss[i]) / ((zs[i + 1] - zs[i]) + 1e-5f);
        cs[i] = cosv;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = lane + 64 * e;
        if (i < n - 1) {
            const float inside = (rs[i] < 1.0f || rs[i + 1] < 1.0f) ? 1.f : 0.f;
            const float dist = zs[i + 1] - zs[i];
            const float mid = (ss[i] + ss[i + 1]) * 0.5f;
            const float prev_cos = i > 0 ? cs[i - 1] : 0.f;
            float c = fminf(prev_cos, cs[i]);
            c = fminf(fmaxf(c, -1e3f), 0.f) * inside;
            const float pe_ = mid - c * dist * 0.5f, ne_ = mid + c * dist * 0.5f;
            const float pc = sigmoid_f(pe_ * inv_s), nc = sigmoid_f(ne_ * inv_s);
            alpha[e] = (pc - nc + 1e-5f) / (pc + 1e-5f);
        }
    }
    __builtin_amdgcn_wave_barrier();                   // (cs is reused for the cdf below)
    // w_i = alpha_i * (float)T_i,  T_{i+1} = T_i * (double)(1 - alpha_i + 1e-7f): the serial order, every lane in step
    double Tr = 1.0;
    float w0 = 0.f, w1 = 0.f;
    for (int i = 0; i < n - 1; ++i) {
        const float a = lane_elem(alpha[0], alpha[1], i);
        const float w = a * (float)Tr;
        if (i == lane) w0 = w;
        if (i == lane + 64) w1 = w;
        Tr *= (double)(1.0f - a + 1e-7f);
    }
    if (w_out) {
        if (lane < n - 1) w_out[(size_t)r * (n - 1) + lane] = w0;
        if (lane + 64 < n - 1) w_out[(size_t)r * (n - 1) + lane + 64] = w1;
    }
    float out; int ind;
    wave_sample_pdf(zs, cs, w0, w1, n, m, lane, out, ind);
    if (lane < m) {
        z_new[(size_t)r * m + lane] = out;
        if (inds_out) inds_out[(size_t)r * m + lane] = ind;
    }
}

__global__ __launch_bounds__(256) void sample_pdf_wave_kernel(const float* __restrict__ bins, int ldb, const float* __restrict__ w, int ldw, int n, int m,
                                                              int R, float* __restrict__ out, int* __restrict__ inds_out) {
    __shared__ float sm[4][2][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    float* zs = sm[wv][0]; float* cs = sm[wv][1];
    zs[lane] = lane < n ? bins[(size_t)r * ldb + lane] : 0.f;
    zs[lane + 64] = lane + 64 < n ? bins[(size_t)r * ldb + lane + 64] : 0.f;
    const float w0 = lane < n - 1 ? w[(size_t)r * ldw + lane] : 0.f, w1 = lane + 64 < n - 1 ? w[(size_t)r * ldw + lane + 64] : 0.f;
    __builtin_amdgcn_wave_barrier();
    float v; int ind;
    wave_sample_pdf(zs, cs, w0, w1, n, m, lane, v, ind);
    if (lane < m) {
        out[(size_t)r * m + lane] = v;
        if (inds_out) inds_out[(size_t)r * m + lane] = ind;
    }
}

// stable merge by ranks: an old element i lands at i + #{j: z_new[j] < z_i}, a new element j at j + #{i: z_i <= z_new[j]} (ties: old
// first) -- for two sorted lists exactly the permutation the serial two-pointer merge produces
__global__ __launch_bounds__(256) void merge_sorted_wave_kernel(float* __restrict__ z, int ldz, int n, float* __restrict__ sdf, int lds,
                                                                const float* __restrict__ z_new, int m, const float* __restrict__ sdf_new, int ldsn,
                                                                int R, int* __restrict__ index_out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    const bool hs = sdf != nullptr && sdf_new != nullptr;
    float zo[2], so[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = lane + 64 * e;
        zo[e] = i < n ? z[(size_t)r * ldz + i] : 0.f;
        if (hs && i < n) so[e] = sdf[(size_t)r * lds + i];
    }
    const float zn = lane < m ? z_new[(size_t)r * m + lane] : 0.f;
    const float sn = (hs && lane < m) ? sdf_new[((size_t)r * m + lane) * ldsn] : 0.f;
    int rank_old[2] = {0, 0}, rank_new = 0;
    for (int j = 0; j < m; ++j) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zn), j));
        rank_old[0] += v < zo[0] ? 1 : 0;
        rank_old[1] += v < zo[1] ? 1 : 0;
        const int cnt = popc64(__ballot(lane < n && zo[0] <= v)) + popc64(__ballot(lane + 64 < n && zo[1] <= v));
        if (j == lane) rank_new = cnt;
    }
    // every load of this ray's row happened above; the stores below go to the same row (one wave owns it)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = lane + 64 * e;
        if (i < n) {
            const int k = i + rank_old[e];
            z[(size_t)r * ldz + k] = zo[e];
            if (hs) sdf[(size_t)r * lds + k] = so[e];
            if (index_out) index_out[(size_t)r * (n + m) + k] = i;
        }
    }
    if (lane < m) {
        const int k = lane + rank_new;
        z[(size_t)r * ldz + k] = zn;
        if (hs) sdf[(size_t)r * lds + k] = sn;
        if (index_out) index_out[(size_t)r * (n + m) + k] = n + lane;
    }
}

// render_prep, one wave per ray, lane i <-> samples i, i + 64, i + 128 (T <= 192): coalesced z loads / float4 stores, the inner
// count is a ballot population count
__global__ __launch_bounds__(256) void render_prep_wave_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int R,
                                                               int T, float* __restrict__ pts4, int* __restrict__ ray_counts) {
    __shared__ float sm[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    float* zs = sm[wv];
    for (int i = lane; i < T; i += 64) zs[i] = z[(size_t)r * T + i];
    __builtin_amdgcn_wave_barrier();
    const float ox = o[r * 3], oy = o[r * 3 + 1], oz = o[r * 3 + 2];
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz_ = d[r * 3 + 2];
    int cnt = 0;
    for (int i0 = 0; i0 < T; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < T) {
            const int k = i < T - 1 ? i : T - 2;       // the last section repeats the previous length (renderer.py:554-558)
            const float dist = zs[k + 1] - zs[k];
            const float mid = zs[i] + dist * 0.5f;
            const float x = ox + dx * mid, y = oy + dy * mid, zz = oz + dz_ * mid;
            reinterpret_cast<float4*>(pts4)[(size_t)r * T + i] = make_float4(x, y, zz, dist);
            in = sqrtf(x * x + y * y + zz * zz) <= 1.0f;
        }
        cnt += popc64(__ballot(in));
    }
    if (lane == 0) ray_counts[r] = cnt;
}

// ordered compaction with wave ballots: position inside the ray = population count of the lower lanes
__global__ __launch_bounds__(256) void compact_wave_kernel(const float* __restrict__ pts4, const int* __restrict__ ray_off_in, int R, int T,
                                                           int* __restrict__ inner_idx, int* __restrict__ outer_idx) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    int ki = ray_off_in[r];
    int ko = r * T - ki;
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < T; i0 += 64) {
        const int i = i0 + lane;
        bool valid = i < T, in = false;
        if (valid) {
            const float4 v = reinterpret_cast<const float4*>(pts4)[(size_t)r * T + i];
            in = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z) <= 1.0f;
        }
        const unsigned long long mi = __ballot(valid && in), mo = __ballot(valid && !in);
        if (valid) {
            if (in) inner_idx[ki + popc64(mi & lower)] = r * T + i;
            else outer_idx[ko + popc64(mo & lower)] = r * T + i;
        }
        ki += popc64(mi);
        ko += popc64(mo);
    }
}

// copy a strided column (head output [rows,4] col 0) into the per-ray sdf table
__global__ void scatter_sdf_kernel(const float* __restrict__ src, int lds_src, int R, int n, float* __restrict__ sdf, int lds) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * n) return;
    const int r = idx / n, i = idx - r * n;
    sdf[(size_t)r * lds + i] = src[(size_t)idx * lds_src];
}

// ---- render preparation: section lengths, mid points, inner/outer split (renderer.py:550-565) ---------------------
// pts4[r*T+i] = (x, y, z, dist); flag = |p| <= 1.  One thread per ray keeps the sample order (ray-major), exactly the
// order boolean-mask indexing produces in the reference.
__global__ void render_prep_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int R, int T,
                                   float* __restrict__ pts4, int* __restrict__ ray_counts) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int cnt = 0;
    const float ox = o[r * 3], oy = o[r * 3 + 1], oz = o[r * 3 + 2];
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz_ = d[r * 3 + 2];
    float zc = z[(size_t)r * T];
    float last = 0.f;
    for (int i = 0; i < T; ++i) {
        float dist;
        float zn = zc;
        if (i < T - 1) { zn = z[(size_t)r * T + i + 1]; dist = zn - zc; last = dist; } else dist = last;
        const float mid = zc + dist * 0.5f;
        const float x = ox + dx * mid, y = oy + dy * mid, zz = oz + dz_ * mid;
        float4 v = make_float4(x, y, zz, dist);
        reinterpret_cast<float4*>(pts4)[(size_t)r * T + i] = v;
        cnt += (sqrtf(x * x + y * y + zz * zz) <= 1.0f) ? 1 : 0;
        zc = zn;
    }
    ray_counts[r] = cnt;
}

// exclusive scan of the per-ray inner counts (single workgroup), totals to counts[0] (inner) / counts[1] (outer)
__global__ void ray_scan_kernel(const int* __restrict__ ray_counts, int R, int T, int* __restrict__ ray_off_in, int* __restrict__ counts) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (R + 1023) / 1024;
    int s = 0;
    for (int k = 0; k < per; ++k) { const int r = tid * per + k; if (r < R) s += ray_counts[r]; }
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        const int r = tid * per + k;
        if (r < R) { ray_off_in[r] = base; base += ray_counts[r]; }
    }
    if (tid == 1023) { counts[0] = part[1023]; counts[1] = R * T - part[1023]; }
}

}  // namespace

#define GRID1D(n) dim3(((n) + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int nero_coarse_z(const float* near, const float* far, const float* rand1, int R, int n, float* z, int ldz, void* stream) {
    if (!near || !far || !z || n > MAXS) return nero_fail(NERO_ERR_ARG, "nero_coarse_z: bad argument");
    if (R == 0) return NERO_OK;
    hipLaunchKernelGGL(coarse_z_kernel, GRID1D(R * n), near, far, rand1, R, n, z, ldz);
    return nero_check_launch("nero_coarse_z");
}

int nero_background_z(const float* far, const float* rand_bg, int R, int n_bg, float* z, int ldz, int col0, void* stream) {
    if (!far || !z) return nero_fail(NERO_ERR_ARG, "nero_background_z: bad argument");
    if (R == 0 || n_bg == 0) return NERO_OK;
    hipLaunchKernelGGL(background_z_kernel, GRID1D(R * n_bg), far, rand_bg, R, n_bg, z, ldz, col0);
    return nero_check_launch("nero_background_z");
}

int nero_ray_points_pe(const float* o, const float* d, const float* z, int ldz, int col0, int ncols, int R, float* pe, void* stream) {
    if (!o || !d || !z || !pe) return nero_fail(NERO_ERR_ARG, "nero_ray_points_pe: bad argument");
    const int n_pad = NERO_ROW_PAD(R * ncols);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(ray_points_pe_row_kernel, dim3(n_pad / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, o, d, z, ldz, col0, ncols, R, n_pad, pe);
    return nero_check_launch("nero_ray_points_pe");
}

int nero_upsample(const float* o, const float* d, const float* z, int ldz, const float* sdf, int lds, int n,
                  const float* variance, float inv_s_cap, int m, int R, float* z_new, float* w_out, int* inds_out, void* stream) {
    if (!o || !d || !z || !sdf || !z_new || n > MAXS || n < 2 || m > 32) return nero_fail(NERO_ERR_ARG, "nero_upsample: bad argument");
    if (R == 0) return NERO_OK;
    if (n <= 128 && m <= 64) {                          // one wave per ray (lane i <-> samples i, i + 64)
        hipLaunchKernelGGL(upsample_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, o, d, z, ldz, sdf, lds, n, variance, inv_s_cap, m, R, z_new, w_out, inds_out);
        return nero_check_launch("nero_upsample");
    }
    NERO_ONCE(hipFuncSetAttribute((const void*)upsample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ray_tables_bytes(3, MAXS)));
    hipLaunchKernelGGL(upsample_kernel, dim3((R + 63) / 64), dim3(64), ray_tables_bytes(3, n), (hipStream_t)stream, o, d, z, ldz, sdf, lds, n, variance, inv_s_cap, m, R, z_new, w_out, inds_out);
    return nero_check_launch("nero_upsample");
}

int nero_sample_pdf(const float* bins, int ldb, const float* w, int ldw, int n, int m, int R, float* out, int* inds_out, void* stream) {
    if (!bins || !w || !out || n > MAXS || n < 2 || m > 32) return nero_fail(NERO_ERR_ARG, "nero_sample_pdf: bad argument");
    if (R == 0) return NERO_OK;
    if (n <= 128 && m <= 64) {
        hipLaunchKernelGGL(sample_pdf_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, bins, ldb, w, ldw, n, m, R, out, inds_out);
        return nero_check_launch("nero_sample_pdf");
    }
    NERO_ONCE(hipFuncSetAttribute((const void*)sample_pdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ray_tables_bytes(3, MAXS)));
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((R + 63) / 64), dim3(64), ray_tables_bytes(3, n), (hipStream_t)stream, bins, ldb, w, ldw, n, m, R, out, inds_out);
    return nero_check_launch("nero_sample_pdf");
}

int nero_merge_sorted(float* z, int ldz, int n, float* sdf, int lds, const float* z_new, int m, const float* sdf_new, int ldsn,
                      int R, int* index_out, void* stream) {
    if (!z || !z_new || n + m > MAXS) return nero_fail(NERO_ERR_ARG, "nero_merge_sorted: bad argument");
    if (R == 0) return NERO_OK;
    if (n <= 128 && m <= 64) {
        hipLaunchKernelGGL(merge_sorted_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, ldz, n, sdf, lds, z_new, m, sdf_new, ldsn, R, index_out);
        return nero_check_launch("nero_merge_sorted");
    }
    NERO_ONCE(hipFuncSetAttribute((const void*)merge_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ray_tables_bytes(2, MAXS)));
    hipLaunchKernelGGL(merge_sorted_kernel, dim3((R + 63) / 64), dim3(64), ray_tables_bytes(2, n), (hipStream_t)stream, z, ldz, n, sdf, lds, z_new, m, sdf_new, ldsn, R, index_out);
    return nero_check_launch("nero_merge_sorted");
}

int nero_scatter_sdf(const float* src, int ld_src, int R, int n, float* sdf, int lds, void* stream) {
    if (!src || !sdf) return nero_fail(NERO_ERR_ARG, "nero_scatter_sdf: bad argument");
    if (R * n == 0) return NERO_OK;
    hipLaunchKernelGGL(scatter_sdf_kernel, GRID1D(R * n), src, ld_src, R, n, sdf, lds);
    return nero_check_launch("nero_scatter_sdf");
}

int nero_render_prep(const float* o, const float* d, const float* z, int R, int T, float* pts4, int* ray_counts, int* ray_off,
                     int* counts, void* stream) {
    if (!o || !d || !z || !pts4 || !ray_counts || !ray_off || !counts) return nero_fail(NERO_ERR_ARG, "nero_render_prep: bad argument");
    if (R == 0) return NERO_OK;
    if (T >= 2 && T <= 192) hipLaunchKernelGGL(render_prep_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, o, d, z, R, T, pts4, ray_counts);
    else hipLaunchKernelGGL(render_prep_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, o, d, z, R, T, pts4, ray_counts);
    hipLaunchKernelGGL(ray_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ray_counts, R, T, ray_off, counts);
    return nero_check_launch("nero_render_prep");
}

int nero_compact(const float* pts4, const int* ray_off, int R, int T, int* inner_idx, int* outer_idx, void* stream) {
    if (!pts4 || !ray_off || !inner_idx || !outer_idx) return nero_fail(NERO_ERR_ARG, "nero_compact: bad argument");
    if (R == 0) return NERO_OK;
    hipLaunchKernelGGL(compact_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, pts4, ray_off, R, T, inner_idx, outer_idx);
    return nero_check_launch("nero_compact");
}

int nero_gather_inner(const float* pts4, const int* idx, int n, float* x4, float* pe, void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    if (!pts4 || !idx || !x4 || !pe) return nero_fail(NERO_ERR_ARG, "nero_gather_inner: bad argument");
    hipLaunchKernelGGL(gather_inner_row_kernel, dim3(n_pad / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, pts4, idx, n, n_pad, x4, pe);
    return nero_check_launch("nero_gather_inner");
}

int nero_gather_outer(const float* pts4, const float* d, const int* idx, int T, int n, float* pe88, float* pev32, float* dist, void* stream) {
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    if (!pts4 || !d || !idx || !pe88 || !pev32 || !dist) return nero_fail(NERO_ERR_ARG, "nero_gather_outer: bad argument");
    hipLaunchKernelGGL(gather_outer_row_kernel, dim3(n_pad / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, pts4, d, idx, T, n, n_pad, pe88, pev32, dist);
    return nero_check_launch("nero_gather_outer");
}

}  // extern "C"

// ---- secondary-ray occlusion march (compute_occ_loss / get_intersection / get_weights, network/renderer.py:522-548,
//      network/field.py:432-484) -------------------------------------------------------------------------------------
namespace {

// surface candidates: |p| < 0.999 & |sdf| < thresh & grad . d < 0        (renderer.py:530-533)
__global__ void occ_candidates_kernel(const float* __restrict__ x4, const float* __restrict__ sdf4, const float* __restrict__ grad,
                                      const int* __restrict__ idx, const float* __restrict__ d, int T, float thresh, int n,
                                      unsigned char* __restrict__ flag) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float x = x4[(size_t)k * 4], y = x4[(size_t)k * 4 + 1], z = x4[(size_t)k * 4 + 2];
    const int r = idx[k] / T;
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz = d[r * 3 + 2];
    const float dn = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    const float dot = (grad[k * 3] * (dx / dn) + grad[k * 3 + 1] * (dy / dn)) + grad[k * 3 + 2] * (dz / dn);
    const bool ok = (sqrtf(x * x + y * y + z * z) < 0.999f) && (fabsf(sdf4[(size_t)k * 4]) < thresh) && (dot < 0.f);
    flag[k] = ok ? 1 : 0;
}

// z[p, i] = maxd_p * lin(0,1,n)[i],  maxd = -<p,d> + sqrt(<p,d>^2 - |p|^2 + 1 + 1e-6)      (field.py:390-396, 474-475)
__global__ void occ_z_kernel(const float* __restrict__ o, const float* __restrict__ d, int P_, int n, float* __restrict__ z) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P_ * n) return;
    const int p = idx / n, i = idx - p * n;
    const float ox = o[p * 3], oy = o[p * 3 + 1], oz = o[p * 3 + 2], dx = d[p * 3], dy = d[p * 3 + 1], dz = d[p * 3 + 2];
    const float dtx = (ox * dx + oy * dy) + oz * dz, xtx = (ox * ox + oy * oy) + oz * oz;
    const float maxd = -dtx + sqrtf(dtx * dtx - xtx + 1.0f + 1e-6f);
    z[idx] = maxd * linspace_f32(0.f, 1.f, n, i);
}

// get_weights (field.py:432-452): weights [P, n-1] (or only their sum when w_out == NULL)
__global__ void section_weights_kernel(const float* __restrict__ z, const float* __restrict__ sdf, int lds, int n,
                                       const float* __restrict__ variance, int P_, float* __restrict__ w_out, float* __restrict__ wsum) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P_) return;
    const float inv_s = expf(variance[0] * 10.0f);
    double T = 1.0;
    float sum = 0.f;
    float zp = z[(size_t)p * n], sp = sdf[(size_t)p * n * lds];
    for (int i = 0; i < n - 1; ++i) {
        const float zn = z[(size_t)p * n + i + 1], sn = sdf[((size_t)p * n + i + 1) * lds];
        const float dist = zn - zp;
        const float mid = (sp + sn) * 0.5f;
        float c = (sn - sp) / (dist + 1e-5f);
        const float mask = c < 0.f ? 1.f : 0.f;
        c = fminf(c, 0.f);
        const float pc = sigmoid_f((mid - c * dist * 0.5f) * inv_s), nc = sigmoid_f((mid + c * dist * 0.5f) * inv_s);
        const float alpha = (pc - nc + 1e-5f) / (pc + 1e-5f) * mask;
        const float w = alpha * (float)T;
        T *= (double)(1.0f - alpha + 1e-7f);
        if (w_out) w_out[(size_t)p * (n - 1) + i] = w;
        sum += w;
        zp = zn; sp = sn;
    }
    if (wsum) wsum[p] = sum;
}

}  // namespace

extern "C" {

int nero_occ_candidates(const float* x4, const float* sdf4, const float* grad, const int* idx, const float* d, int T, float thresh,
                        int n, unsigned char* flag, void* stream) {
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(occ_candidates_kernel, GRID1D(n), x4, sdf4, grad, idx, d, T, thresh, n, flag);
    return nero_check_launch("nero_occ_candidates");
}

int nero_occ_z(const float* o, const float* d, int P_, int n, float* z, void* stream) {
    if (P_ * n == 0) return NERO_OK;
    hipLaunchKernelGGL(occ_z_kernel, GRID1D(P_ * n), o, d, P_, n, z);
    return nero_check_launch("nero_occ_z");
}

int nero_section_weights(const float* z, const float* sdf, int lds, int n, const float* variance, int P_, float* w_out, float* wsum,
                         void* stream) {
    if (P_ == 0) return NERO_OK;
    hipLaunchKernelGGL(section_weights_kernel, dim3((P_ + 63) / 64), dim3(64), 0, (hipStream_t)stream, z, sdf, lds, n, variance, P_, w_out, wsum);
    return nero_check_launch("nero_section_weights");
}

}  // extern "C"
