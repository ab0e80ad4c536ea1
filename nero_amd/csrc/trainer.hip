// trainer.hip -- trainer-loop fusion for the Stage-I / Stage-II step.
//
// The reference reparametrises every weight-normed Linear (nn.utils.weight_norm, network/field.py:118-119, 323-331) and runs
// Adam (train/trainer.py:105-170, torch.optim.Adam over ~125 tensors) as ~750 tiny kernels per step.  Here:
//   nero_wn_forward_batch : W_l = g_l v_l / ||v_l||_row for EVERY weight-normed matrix of the model in ONE launch (row norms kept)
//   nero_wn_adam_batch    : given dL/dW_l (what the render step's backward produces, already all-reduced), the weight-norm
//                           backward  dg = <dW, v> / n,  dv = (g / n) (dW - <dW, v> v / n^2)  fused with the Adam update of g and v,
//                           plus plain Adam for every other tensor (biases, NeRF++ Linear weights, the variance scalar): ONE launch
// Adam follows torch.optim.Adam (betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) in the fused kernel's operation order:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps).
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

struct WnBatch { nero_wn_job job[NERO_MAX_WN_JOBS]; };
struct AdamBatch { nero_adam_job job[NERO_MAX_ADAM_JOBS]; };

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// one wave per matrix row; grid (ceil(max_rows / 4), n_jobs), 256 threads
__global__ __launch_bounds__(256) void wn_forward_kernel(WnBatch B) {
    const nero_wn_job& J = B.job[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= J.rows) return;
    const float* v = J.v + (size_t)row * J.cols;
    float s = 0.f;
    for (int k = lane; k < J.cols; k += 64) s = fmaf(v[k], v[k], s);
    s = wave_sum(s);
    const float n = sqrtf(s);
    const float sc = J.g[row] / n;
    float* w = J.w_eff + (size_t)row * J.cols;
    for (int k = lane; k < J.cols; k += 64) w[k] = v[k] * sc;
    if (lane == 0) J.inv_norm[row] = 1.f / n;
}

struct AdamHyper { float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, lr; };

// bias corrections of a job that keeps its own step counter (nero_adam_job.step > 0)
__device__ __forceinline__ AdamHyper hyper_for_step(AdamHyper H, int step) {
    const double bc1 = 1.0 - pow((double)H.beta1, (double)step), bc2 = 1.0 - pow((double)H.beta2, (double)step);
    H.lr_over_bc1 = (float)((double)H.lr / bc1);
    H.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return H;
}

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, const AdamHyper& H) {
    m = m + (g - m) * (1.f - H.beta1);
    v = H.beta2 * v + (1.f - H.beta2) * g * g;
    const float denom = sqrtf(v) * H.inv_sqrt_bc2 + H.eps;
    p = p - H.lr_over_bc1 * (m / denom);
}

// weight-norm backward + Adam on (g, v): one wave per row
__global__ __launch_bounds__(256) void wn_adam_kernel(WnBatch B, AdamHyper H) {
    const nero_wn_job& J = B.job[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= J.rows) return;
    float* v = J.v_rw + (size_t)row * J.cols;
    const float* dW = J.dW + (size_t)row * J.cols;
    float s = 0.f;
    for (int k = lane; k < J.cols; k += 64) s = fmaf(dW[k], v[k], s);
    s = wave_sum(s);
    const float inv = J.inv_norm[row], g = J.g_rw[row];
    const float c1 = g * inv, c2 = s * inv * inv;
    float* mv = J.m_v + (size_t)row * J.cols;
    float* vv = J.v_v + (size_t)row * J.cols;
    for (int k = lane; k < J.cols; k += 64) {
        const float dv = c1 * (dW[k] - c2 * v[k]);
        float p = v[k], m = mv[k], q = vv[k];
        adam_one(p, m, q, dv, H);
        v[k] = p; mv[k] = m; vv[k] = q;
    }
    if (lane == 0) {
        float p = g, m = J.m_g[row], q = J.v_g[row];
        adam_one(p, m, q, s * inv, H);
        J.g_rw[row] = p; J.m_g[row] = m; J.v_g[row] = q;
    }
}

// weight-norm backward alone (nero_wn_backward_batch): one wave per row
struct WnGradBatch { nero_wn_grad_job job[NERO_MAX_WN_JOBS]; };
__global__ __launch_bounds__(256) void wn_backward_kernel(WnGradBatch B) {
    const nero_wn_grad_job& J = B.job[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= J.rows) return;
    const float* v = J.v + (size_t)row * J.cols;
    const float* dW = J.dW + (size_t)row * J.cols;
    float s = 0.f;
    for (int k = lane; k < J.cols; k += 64) s = fmaf(dW[k], v[k], s);
    s = wave_sum(s);
    const float inv = J.inv_norm[row];
    const float c1 = J.g[row] * inv, c2 = s * inv * inv;
    float* dv = J.dv + (size_t)row * J.cols;
    for (int k = lane; k < J.cols; k += 64) dv[k] = c1 * (dW[k] - c2 * v[k]);
    if (lane == 0) J.dg[row] = s * inv;
}

// plain Adam: grid (blocks, n_jobs), grid-stride
__global__ __launch_bounds__(256) void adam_kernel(AdamBatch B, AdamHyper H) {
    const nero_adam_job& J = B.job[blockIdx.y];
    if (J.step < 0) return;                                   // no gradient this step: parameter, moments and step count untouched
    if (J.step > 0) H = hyper_for_step(H, J.step);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < J.n; idx += gridDim.x * 256) {
        float p = J.p[idx], m = J.m[idx], q = J.v[idx];
        adam_one(p, m, q, J.grad[idx], H);
        J.p[idx] = p; J.m[idx] = m; J.v[idx] = q;
    }
}

}  // namespace

extern "C" {

int nero_wn_forward_batch(const nero_wn_job* jobs, int n_jobs, void* stream) {
    if (n_jobs <= 0) return NERO_OK;
    if (!jobs || n_jobs > NERO_MAX_WN_JOBS) return nero_fail(NERO_ERR_ARG, "nero_wn_forward_batch: bad job count");
    WnBatch B;
    int max_rows = 1;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].v || !jobs[i].g || !jobs[i].w_eff || !jobs[i].inv_norm || jobs[i].rows <= 0 || jobs[i].cols <= 0)
            return nero_fail(NERO_ERR_ARG, "nero_wn_forward_batch: bad job");
        B.job[i] = jobs[i];
        max_rows = jobs[i].rows > max_rows ? jobs[i].rows : max_rows;
    }
    hipLaunchKernelGGL(wn_forward_kernel, dim3((max_rows + 3) / 4, n_jobs), dim3(256), 0, (hipStream_t)stream, B);
    return nero_check_launch("nero_wn_forward_batch");
}

int nero_wn_backward_batch(const nero_wn_grad_job* jobs, int n_jobs, void* stream) {
    if (n_jobs <= 0) return NERO_OK;
    if (!jobs || n_jobs > NERO_MAX_WN_JOBS) return nero_fail(NERO_ERR_ARG, "nero_wn_backward_batch: bad job count");
    WnGradBatch B;
    int max_rows = 1;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].v || !jobs[i].g || !jobs[i].inv_norm || !jobs[i].dW || !jobs[i].dv || !jobs[i].dg || jobs[i].rows <= 0 || jobs[i].cols <= 0)
            return nero_fail(NERO_ERR_ARG, "nero_wn_backward_batch: bad job");
        B.job[i] = jobs[i];
        max_rows = jobs[i].rows > max_rows ? jobs[i].rows : max_rows;
    }
    hipLaunchKernelGGL(wn_backward_kernel, dim3((max_rows + 3) / 4, n_jobs), dim3(256), 0, (hipStream_t)stream, B);
    return nero_check_launch("nero_wn_backward_batch");
}

int nero_wn_adam_batch(const nero_wn_job* wn, int n_wn, const nero_adam_job* plain, int n_plain, float lr, float beta1, float beta2,
                       float eps, int step, void* stream) {
    if (n_wn < 0 || n_plain < 0 || n_wn > NERO_MAX_WN_JOBS || step < 1) return nero_fail(NERO_ERR_ARG, "nero_wn_adam_batch: bad argument");
    AdamHyper H;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    H.lr_over_bc1 = (float)(lr / bc1);
    H.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    H.beta1 = beta1; H.beta2 = beta2; H.eps = eps; H.lr = lr;
    if (n_wn > 0) {
        WnBatch B;
        int max_rows = 1;
        for (int i = 0; i < n_wn; ++i) {
            if (!wn[i].v_rw || !wn[i].g_rw || !wn[i].dW || !wn[i].inv_norm || !wn[i].m_v || !wn[i].v_v || !wn[i].m_g || !wn[i].v_g)
                return nero_fail(NERO_ERR_ARG, "nero_wn_adam_batch: bad weight-norm job");
            B.job[i] = wn[i];
            max_rows = wn[i].rows > max_rows ? wn[i].rows : max_rows;
        }
        hipLaunchKernelGGL(wn_adam_kernel, dim3((max_rows + 3) / 4, n_wn), dim3(256), 0, (hipStream_t)stream, B, H);
    }
    for (int i0 = 0; i0 < n_plain; i0 += NERO_MAX_ADAM_JOBS) {
        const int n = n_plain - i0 < NERO_MAX_ADAM_JOBS ? n_plain - i0 : NERO_MAX_ADAM_JOBS;
        AdamBatch A;
        int max_n = 1;
        for (int i = 0; i < n; ++i) {
            const nero_adam_job& J = plain[i0 + i];
            if (!J.p || !J.grad || !J.m || !J.v || J.n <= 0) return nero_fail(NERO_ERR_ARG, "nero_wn_adam_batch: bad Adam job");
            A.job[i] = J;
            max_n = J.n > max_n ? J.n : max_n;
        }
        int bx = (max_n + 255) / 256;
        bx = bx > 64 ? 64 : bx;
        hipLaunchKernelGGL(adam_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, A, H);
    }
    return nero_check_launch("nero_wn_adam_batch");
}

}  // extern "C"
