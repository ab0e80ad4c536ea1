// step_glue.hip -- the Stage-I training glue BETWEEN the C-level driver calls (nero_stage1_sample / _render_fwd / _render_bwd) as a
// handful of launches instead of ~110 tiny tensor ops and a second host synchronisation (round 4: the step's torch
// glue cost 0.75 ms at every batch size -- 14 % of the reference's own 512-ray step, scripts/r04/glue_floor.py):
//   nero_near_far_sphere   near / far of the unit sphere along every ray                  (network/renderer.py:224-230)
//   nero_occ_select        the occlusion-loss candidate subset, on the device: one stable radix sort (hipCUB) of (key of the candidate's
//                          ordinal | +inf, sample index) over ALL inner samples, then the first min(#candidates, cap) indices in
//                          ascending order -- what torch.nonzero + argsort(keys, stable)[:cap] + sort did with a host read-back of
//                          the candidate count (network/renderer.py:535-541).  Launch sizes depend on n only.
//   nero_occ_gather        surface points / reflected directions of the kept candidates into a FIXED-capacity batch (unused slots
//                          get a harmless dummy ray: the march always runs `cap` rays, nothing waits for the count)
//   nero_occ_l1 (+ _backward)  the occlusion L1 mean alone and its seed d_occ, for callers that assemble the loss under autograd (drop-in renderer)
//   nero_shape_loss        loss_rgb (l2 / l1 / smooth_l1 / charbonier), the eikonal mean, the occlusion L1 mean, their weighted sum
//                          (train/trainer.py:127-137, network/loss.py:8-55) AND the seeds of the backward pass: d_rgb, d_gerr, d_occ
//   nero_var_grad          d loss / d variance from the driver's d loss / d inv_s        (inv_s = exp(10 variance), clipped)
// Reductions are per block and then by ONE block over the partials in a fixed order: values do not depend on scheduling.  Sub-gradient
// conventions follow the ATen kernels the reference runs (sign(0) = 0).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

constexpr int LB = 256;
constexpr int SB = 1024;                                 // elements per scan block (LB threads x 4)

__global__ __launch_bounds__(LB) void near_far_kernel(const float* __restrict__ o, const float* __restrict__ d, int R,
                                                      float* __restrict__ near, float* __restrict__ far) {
#pragma clang fp contract(off)                           // (products rounded before they are summed, like the tensor expression)
    const int r = blockIdx.x * LB + threadIdx.x;
    if (r >= R) return;
    const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2], dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
    // (x + z) + y: the order ATen's reduction adds three contiguous elements in on this device (two strided accumulators) -- with it
    // near / far are bit-identical to the tensor expression's (tests/test_step_glue.py)
    const float a = (dx * dx + dz * dz) + dy * dy;
    const float b = 2.0f * ((ox * dx + oz * dz) + oy * dy);
    const float mid = 0.5f * (-b) / a;
    near[r] = fmaxf(mid - 1.0f, 1e-3f);
    far[r] = mid + 1.0f;
}

// ---- occlusion-loss subset ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {       // LB threads; returns the exclusive prefix of v
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < LB; off <<= 1) {
        const int x = t >= off ? lds[t - off] : 0;
        __syncthreads();
        lds[t] += x;
        __syncthreads();
    }
    total = lds[LB - 1];
    const int ex = lds[t] - v;
    __syncthreads();
    return ex;
}
__global__ __launch_bounds__(LB) void occ_count_kernel(const unsigned char* __restrict__ flag, int n, int* __restrict__ blk) {
    __shared__ int lds[LB];
    const int base = blockIdx.x * SB + 4 * threadIdx.x;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (base + j < n && flag[base + j]) ? 1 : 0;
    int total;
    (void)block_exclusive_scan(c, lds, total);
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}
// ONE block: blk[] -> exclusive prefix in place; counts = (kept, total)
__global__ __launch_bounds__(LB) void occ_scan_kernel(int* __restrict__ blk, int nb, int cap, int* __restrict__ counts) {
    __shared__ int lds[LB];
    int run = 0;
    for (int b0 = 0; b0 < nb; b0 += LB) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? blk[b] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds, total);
        if (b < nb) blk[b] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) { counts[0] = run < cap ? run : cap; counts[1] = run; }
}
// sort records of the subset selection: every sample i gets (key, i) with key = keys[ordinal of i among the candidates] for a candidate
// and +inf otherwise -- a STABLE sort by key then lists the candidates by (key, ordinal), i.e. argsort(keys[:Pn], stable), with the
// non-candidates behind them
__global__ __launch_bounds__(LB) void occ_records_kernel(const unsigned char* __restrict__ flag, int n, const int* __restrict__ blk,
                                                         const float* __restrict__ keys, float* __restrict__ skey, int* __restrict__ sval) {
    __shared__ int lds[LB];
    const int base = blockIdx.x * SB + 4 * threadIdx.x;
    int f[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = (base + j < n && flag[base + j]) ? 1 : 0; c += f[j]; }
    int total;
    int p = blk[blockIdx.x] + block_exclusive_scan(c, lds, total);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) {
            skey[base + j] = f[j] ? keys[p] : INFINITY;
            sval[base + j] = base + j;
            p += f[j];
        }
}
// ONE block: the first `kept` sorted records -> their sample indices in ascending order (bitonic sort in LDS over the next power of two
// >= cap, at most 4096), -1 in the unused slots
constexpr int PICK_MAX = 4096;
__global__ __launch_bounds__(1024) void occ_pick_kernel(const int* __restrict__ counts, const int* __restrict__ sorted_idx, int n, int cap,
                                                        int m2, int* __restrict__ cand) {
    __shared__ int v[PICK_MAX];
    const int kept = counts[0];
    for (int k = threadIdx.x; k < m2; k += 1024) v[k] = (k < kept && k < n) ? sorted_idx[k] : 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= m2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (m2 >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1));           // index of the lower element of pair t at this stride
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int a = v[lo], b = v[hi];
                if ((a > b) == up) { v[lo] = b; v[hi] = a; }
            }
            __syncthreads();
        }
    for (int k = threadIdx.x; k < cap; k += 1024) cand[k] = v[k] == 0x7fffffff ? -1 : v[k];
}
__global__ __launch_bounds__(LB) void occ_gather_kernel(const float* __restrict__ x4, const float* __restrict__ geo,
                                                        const int* __restrict__ cand, int cap, float* __restrict__ pts,
                                                        float* __restrict__ dirs) {
    const int k = blockIdx.x * LB + threadIdx.x;
    if (k >= cap) return;
    const int i = cand[k];
    if (i < 0) {                                         // unused slot: a ray from the centre along +z (marched, never read)
        pts[3 * k] = pts[3 * k + 1] = pts[3 * k + 2] = 0.f;
        dirs[3 * k] = dirs[3 * k + 1] = 0.f; dirs[3 * k + 2] = 1.f;
        return;
    }
    pts[3 * k] = x4[4 * (size_t)i]; pts[3 * k + 1] = x4[4 * (size_t)i + 1]; pts[3 * k + 2] = x4[4 * (size_t)i + 2];
    dirs[3 * k] = geo[8 * (size_t)i + 4]; dirs[3 * k + 1] = geo[8 * (size_t)i + 5]; dirs[3 * k + 2] = geo[8 * (size_t)i + 6];
}

// ---- the occlusion L1 term alone, for callers that assemble the loss themselves (the drop-in renderer under autograd) -------------------------
// loss[0] = sum_k |occ_prob[cand[k]] - gt[k]| over the used slots / max(kept, 1); ONE block, fixed tree: the value does not depend on scheduling
__global__ __launch_bounds__(1024) void occ_l1_kernel(const float* __restrict__ occ_prob, const int* __restrict__ cand, const int* __restrict__ counts,
                                                      const float* __restrict__ gt, int cap, float* __restrict__ loss) {
    __shared__ float part[1024];
    float a = 0.f;
    for (int k = threadIdx.x; k < cap; k += 1024) {
        const int i = cand[k];
        if (i >= 0) a += fabsf(occ_prob[i] - gt[k]);
    }
    part[threadIdx.x] = a;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int kept = counts[0] > 1 ? counts[0] : 1;
        loss[0] = part[0] / (float)kept;
    }
}
__device__ __forceinline__ float sgn0(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
// d_occ[cand[k]] = d_loss[0] * sign(occ_prob[cand[k]] - gt[k]) / max(kept, 1) on a d_occ the caller zeroed (candidates are distinct samples)
__global__ __launch_bounds__(LB) void occ_l1_bwd_kernel(const float* __restrict__ d_loss, const float* __restrict__ occ_prob,
                                                        const int* __restrict__ cand, const int* __restrict__ counts, const float* __restrict__ gt,
                                                        int cap, float* __restrict__ d_occ) {
    const int k = blockIdx.x * LB + threadIdx.x;
    if (k >= cap) return;
    const int i = cand[k];
    if (i < 0) return;
    const int kept = counts[0] > 1 ? counts[0] : 1;
    d_occ[i] = d_loss[0] * sgn0(occ_prob[i] - gt[k]) / (float)kept;
}

// ---- loss + backward seeds ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float block_sum(float v, float* lds) {                        // LB threads, fixed order; result in every thread
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = LB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
        __syncthreads();
    }
    const float s = lds[0];
    __syncthreads();
    return s;
}
// thread i: ray i (rgb loss + d_rgb) and inner sample i (eikonal term: d_gerr, zeroed d_occ); partials[2 b] / [2 b + 1] = block sums
__global__ __launch_bounds__(LB) void loss_rows_kernel(int R, int kind, const float* __restrict__ rgb, const float* __restrict__ gt,
                                                       int n_in, const float* __restrict__ gerr, float eik_w,
                                                       const float* __restrict__ w, float* __restrict__ d_rgb,
                                                       float* __restrict__ d_gerr, float* __restrict__ d_occ,
                                                       float* __restrict__ partials) {
    __shared__ float lds[LB];
    const int i = blockIdx.x * LB + threadIdx.x;
    float lr = 0.f, lg = 0.f;
    if (i < R) {
        const float inv_r = 1.0f / (float)R;
        float df[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) df[c] = rgb[3 * i + c] - gt[3 * i + c];
        if (kind == NERO_RGB_L2) {
            lr = df[0] * df[0] + df[1] * df[1] + df[2] * df[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = 2.f * df[c];
        } else if (kind == NERO_RGB_L1) {
            lr = fabsf(df[0]) + fabsf(df[1]) + fabsf(df[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = sgn(df[c]);
        } else if (kind == NERO_RGB_SMOOTH_L1) {         // beta = 0.25 (network/loss.py: F.smooth_l1_loss(..., beta=0.25))
            const float beta = 0.25f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = fabsf(df[c]);
                if (a < beta) { lr += 0.5f * df[c] * df[c] / beta; g[c] = df[c] / beta; }
                else { lr += a - 0.5f * beta; g[c] = sgn(df[c]); }
            }
        } else {                                         // charbonier: sqrt(sum (gt - pr)^2 + 1e-3)
            lr = sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + 0.001f);
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = df[c] / lr;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) d_rgb[3 * i + c] = g[c] * inv_r;
    }
    if (i < n_in) {
        lg = gerr[i];
        d_gerr[i] = eik_w * (w ? w[0] : 1.f) / (float)n_in;
        if (d_occ) d_occ[i] = 0.f;
    }
    const float sr = block_sum(lr, lds), sg = block_sum(lg, lds);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sr; partials[2 * blockIdx.x + 1] = sg; }
}
// ONE block: the sums, the occlusion L1 term over the kept candidates (+ its seeds scattered into d_occ), the weighted total
__global__ __launch_bounds__(LB) void loss_final_kernel(int R, int n_in, float eik_w, int nb, const float* __restrict__ partials,
                                                        const float* __restrict__ occ_prob, const int* __restrict__ cand,
                                                        const int* __restrict__ counts, const float* __restrict__ gt_occ,
                                                        const float* __restrict__ w, float* __restrict__ d_occ,
                                                        float* __restrict__ losses) {
    __shared__ float lds[LB];
    float sr = 0.f, sg = 0.f;
    for (int b = threadIdx.x; b < nb; b += LB) { sr += partials[2 * b]; sg += partials[2 * b + 1]; }
    sr = block_sum(sr, lds);
    sg = block_sum(sg, lds);
    const float w_eik = w ? w[0] : 1.f, w_occ = w ? w[1] : 1.f;
    float so = 0.f;
    int kept = 0;
    if (cand) {
        kept = counts[0];
        for (int k = threadIdx.x; k < kept; k += LB) {
            const int i = cand[k];
            const float df = occ_prob[i] - gt_occ[k];
            so += fabsf(df);
            d_occ[i] = w_occ * sgn(df) / (float)kept;
        }
        so = block_sum(so, lds);
    }
    if (threadIdx.x == 0) {
        const float l_rgb = sr / (float)R;
        const float l_eik = n_in > 0 ? eik_w * sg / (float)n_in * w_eik : 0.f;
        const float l_occ = kept > 0 ? so / (float)kept * w_occ : 0.f;
        losses[0] = l_rgb + l_eik + l_occ;
        losses[1] = l_rgb;
        losses[2] = l_eik;
        losses[3] = l_occ;
    }
}
__global__ void var_grad_kernel(const float* __restrict__ dsum, const float* __restrict__ variance, float* __restrict__ grad) {
    const float inv_s = expf(variance[0] * 10.0f);
    const float live = (inv_s >= 1e-6f && inv_s <= 1e6f) ? 1.f : 0.f;
    grad[0] = dsum[0] * 10.0f * inv_s * live;
}

}  // namespace

extern "C" {

int nero_near_far_sphere(const float* o, const float* d, int R, float* near, float* far, void* stream) {
    if (R <= 0) return NERO_OK;
    if (!o || !d || !near || !far) return nero_fail(NERO_ERR_ARG, "nero_near_far_sphere: null pointer");
    hipLaunchKernelGGL(near_far_kernel, dim3((R + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, o, d, R, near, far);
    return nero_check_launch("nero_near_far_sphere");
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }
static size_t occ_sort_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs((void*)nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const int*)nullptr, (int*)nullptr, n);
    return bytes;
}
size_t nero_occ_select_workspace(int n) {
    if (n <= 0) return 0;
    const size_t nb = (size_t)(n + SB - 1) / SB;
    return align256((nb + 1) * 4) + 4 * align256((size_t)n * 4) + align256(occ_sort_temp_bytes(n));
}

int nero_occ_select(const unsigned char* flag, int n, const float* keys, int cap, int* cand, int* counts, void* ws, size_t ws_bytes,
                    void* stream) {
    if (n <= 0 || cap <= 0) return nero_fail(NERO_ERR_ARG, "nero_occ_select: n and cap must be positive");
    if (cap > PICK_MAX) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_occ_select: cap > 4096 (the in-LDS sort of the kept indices)");
    if (!flag || !keys || !cand || !counts || !ws) return nero_fail(NERO_ERR_ARG, "nero_occ_select: null pointer");
    if (ws_bytes < nero_occ_select_workspace(n)) return nero_fail(NERO_ERR_ARG, "nero_occ_select: workspace too small (nero_occ_select_workspace)");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (n + SB - 1) / SB;
    char* p = (char*)ws;
    int* blk = (int*)p;                 p += align256((size_t)(nb + 1) * 4);
    float* skey = (float*)p;            p += align256((size_t)n * 4);
    float* skey2 = (float*)p;           p += align256((size_t)n * 4);
    int* sval = (int*)p;                p += align256((size_t)n * 4);
    int* sval2 = (int*)p;               p += align256((size_t)n * 4);
    size_t temp_bytes = occ_sort_temp_bytes(n);
    hipLaunchKernelGGL(occ_count_kernel, dim3(nb), dim3(LB), 0, s, flag, n, blk);
    hipLaunchKernelGGL(occ_scan_kernel, dim3(1), dim3(LB), 0, s, blk, nb, cap, counts);
    hipLaunchKernelGGL(occ_records_kernel, dim3(nb), dim3(LB), 0, s, flag, n, blk, keys, skey, sval);
    if (hipcub::DeviceRadixSort::SortPairs((void*)p, temp_bytes, skey, skey2, sval, sval2, n, 0, 32, s) != hipSuccess)
        return nero_fail(NERO_ERR_LAUNCH, "nero_occ_select: radix sort failed");
    int m2 = 2;
    while (m2 < cap) m2 <<= 1;
    hipLaunchKernelGGL(occ_pick_kernel, dim3(1), dim3(1024), 0, s, counts, sval2, n, cap, m2, cand);
    return nero_check_launch("nero_occ_select");
}

int nero_occ_gather(const float* x4, const float* geo, const int* cand, int cap, float* pts, float* dirs, void* stream) {
    if (cap <= 0) return NERO_OK;
    if (!x4 || !geo || !cand || !pts || !dirs) return nero_fail(NERO_ERR_ARG, "nero_occ_gather: null pointer");
    hipLaunchKernelGGL(occ_gather_kernel, dim3((cap + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, x4, geo, cand, cap, pts, dirs);
    return nero_check_launch("nero_occ_gather");
}

int nero_occ_l1(const float* occ_prob, const int* cand, const int* counts, const float* gt, int cap, float* loss, void* stream) {
    if (cap <= 0 || cap > PICK_MAX) return nero_fail(NERO_ERR_ARG, "nero_occ_l1: cap must be in 1 ... 4096");
    if (!occ_prob || !cand || !counts || !gt || !loss) return nero_fail(NERO_ERR_ARG, "nero_occ_l1: null pointer");
    hipLaunchKernelGGL(occ_l1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, occ_prob, cand, counts, gt, cap, loss);
    return nero_check_launch("nero_occ_l1");
}

int nero_occ_l1_backward(const float* d_loss, const float* occ_prob, const int* cand, const int* counts, const float* gt, int cap, int n_in,
                         float* d_occ, void* stream) {
    if (cap <= 0 || n_in <= 0) return nero_fail(NERO_ERR_ARG, "nero_occ_l1_backward: cap and n_in must be positive");
    if (!d_loss || !occ_prob || !cand || !counts || !gt || !d_occ) return nero_fail(NERO_ERR_ARG, "nero_occ_l1_backward: null pointer");
    if (hipMemsetAsync(d_occ, 0, sizeof(float) * (size_t)n_in, (hipStream_t)stream) != hipSuccess)
        return nero_fail(NERO_ERR_LAUNCH, "nero_occ_l1_backward: memset failed");
    hipLaunchKernelGGL(occ_l1_bwd_kernel, dim3((cap + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, d_loss, occ_prob, cand, counts, gt, cap, d_occ);
    return nero_check_launch("nero_occ_l1_backward");
}

int nero_shape_loss_partials(int R, int n_in) {
    const int m = R > n_in ? R : n_in;
    return 2 * ((m + LB - 1) / LB);
}

int nero_shape_loss(int R, int rgb_kind, const float* rgb, const float* gt, int n_in, const float* gerr, float eik_weight,
                    const float* occ_prob, const int* cand, const int* counts, const float* gt_occ, const float* weights, float* losses,
                    float* d_rgb, float* d_gerr, float* d_occ, float* partials, void* stream) {
    if (R <= 0) return nero_fail(NERO_ERR_ARG, "nero_shape_loss: no rays");
    if (rgb_kind < NERO_RGB_L2 || rgb_kind > NERO_RGB_CHARBONIER) return nero_fail(NERO_ERR_ARG, "nero_shape_loss: unknown rgb loss kind");
    if (!rgb || !gt || !losses || !d_rgb || !partials) return nero_fail(NERO_ERR_ARG, "nero_shape_loss: null pointer");
    if (n_in > 0 && (!gerr || !d_gerr)) return nero_fail(NERO_ERR_ARG, "nero_shape_loss: gerr / d_gerr missing");
    if (cand && (n_in <= 0 || !counts || !gt_occ || !occ_prob || !d_occ)) return nero_fail(NERO_ERR_ARG, "nero_shape_loss: occlusion term incomplete");
    hipStream_t s = (hipStream_t)stream;
    const int nb = nero_shape_loss_partials(R, n_in) / 2;
    hipLaunchKernelGGL(loss_rows_kernel, dim3(nb), dim3(LB), 0, s, R, rgb_kind, rgb, gt, n_in, gerr, eik_weight, weights, d_rgb, d_gerr, d_occ,
                       partials);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(LB), 0, s, R, n_in, eik_weight, nb, partials, occ_prob, cand, counts, gt_occ, weights,
                       d_occ, losses);
    return nero_check_launch("nero_shape_loss");
}

int nero_var_grad(const float* dsum, const float* variance, float* grad, void* stream) {
    if (!dsum || !variance || !grad) return nero_fail(NERO_ERR_ARG, "nero_var_grad: null pointer");
    hipLaunchKernelGGL(var_grad_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dsum, variance, grad);
    return nero_check_launch("nero_var_grad");
}

}  // extern "C"
