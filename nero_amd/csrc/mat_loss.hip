// mat_loss.hip -- the Stage-II training glue around the shading kernels as five launches instead of ~230 tiny tensor ops:
//   nero_mat_reg_points   tangent-plane perturbation of the surface points        (network/field.py:1066-1076, 756-766)
//   nero_mat_head_fwd/bwd sigmoid heads of the material MLP, roughness affine      (network/field.py:915-922)
//   nero_mat_loss_fwd/bwd loss_rgb (Charbonnier / L1 on sRGB), loss_mat_reg (smoothness + the saturation hinge of the first 2000
//                         steps), loss_diffuse_light; the sum of their means        (network/renderer.py:837-844, utils/raw_utils.py:4-10,
//                         network/field.py:1061-1087, train/trainer.py:134-137)
// One thread per surface point.  The loss is reduced per block and then by ONE block over the partials in a fixed order: the value
// does not depend on scheduling.  Sub-gradient conventions follow the ATen kernels the reference runs: sign(0) = 0 for |x|,
// clamp passes the gradient on its closed interval, the `where` of the sRGB curve differentiates the branch taken.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

constexpr int LB = 256;
// (constants as the Python glue of the reference forms them: double arithmetic, then one rounding to fp32 when they meet a tensor)
constexpr float R_MIN = (float)(0.04 * 0.04);            // roughness range [0.04^2, 1]
constexpr float R_SPAN = (float)(1.0 - 0.04 * 0.04);
constexpr float H_R_HI = (float)(0.98 * 0.98), H_R_LO = (float)(0.02 * 0.02), H_M_HI = 0.98f, H_M_LO = 0.02f;
constexpr float SRGB_EPS = 1.1920928955078125e-07f;     // torch.finfo(float32).eps

__device__ __forceinline__ float srgb(float x) {
    return x <= 0.0031308f ? (323.f / 25.f) * x : (211.f * powf(fmaxf(x, SRGB_EPS), 5.f / 12.f) - 11.f) / 200.f;
}
__device__ __forceinline__ float srgb_grad(float x) {
    if (x <= 0.0031308f) return 323.f / 25.f;
    return (211.f / 200.f) * (5.f / 12.f) * powf(fmaxf(x, SRGB_EPS), -7.f / 12.f);       // (x > 0.0031308 > eps: the clamp is inactive)
}
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(LB) void reg_points_kernel(int P, const float* __restrict__ pts, const float* __restrict__ nrm,
                                                        const float* __restrict__ ang01, const float* __restrict__ eps, float eps_const,
                                                        float* __restrict__ out) {
    const int p = blockIdx.x * LB + threadIdx.x;
    if (p >= P) return;
    const float px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
    float nx = nrm[3 * p], ny = nrm[3 * p + 1], nz = nrm[3 * p + 2];
    const float nn = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
    nx /= nn; ny /= nn; nz /= nn;
    // get_orthogonal_directions: the longer of (n1, -n0, 0) and (-n2, 0, n0), normalised; y = n x x
    const float l0 = sqrtf(ny * ny + nx * nx), l1 = sqrtf(nz * nz + nx * nx);
    float xx, xy, xz;
    if (l0 > l1) { xx = ny; xy = -nx; xz = 0.f; } else { xx = -nz; xy = 0.f; xz = nx; }
    const float xl = fmaxf(l0 > l1 ? l0 : l1, 1e-12f);
    xx /= xl; xy /= xl; xz /= xl;
    const float yx = ny * xz - nz * xy, yy = nz * xx - nx * xz, yz = nx * xy - ny * xx;
    const float a = ang01[p] * 3.14159274101257324f * 2.f;
    const float c = cosf(a), s = sinf(a), e = eps ? eps[p] : eps_const;
    out[3 * p] = px; out[3 * p + 1] = py; out[3 * p + 2] = pz;
    float* o = out + (size_t)3 * (P + p);
    o[0] = px + (c * xx + s * yx) * e;
    o[1] = py + (c * xy + s * yy) * e;
    o[2] = pz + (c * xz + s * yz) * e;
}

__global__ __launch_bounds__(LB) void head_fwd_kernel(int n5, const float* __restrict__ raw, float* __restrict__ mat) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= n5) return;
    const float s = sigmoidf(raw[i]);
    // (separate multiply and add, as the tensor ops of the reference round them: a contracted fma moves the roughness by one ulp, and
    //  the specular sample directions -- hence every light-MLP gradient -- with it)
    float r;
    {
#pragma clang fp contract(off)
        const float t = s * R_SPAN;
        r = t + R_MIN;
    }
    mat[i] = (i % 5 == 1) ? r : s;
}
__global__ __launch_bounds__(LB) void head_bwd_kernel(int n5, const float* __restrict__ raw, const float* __restrict__ d_mat, float* __restrict__ d_raw) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= n5) return;
    const float s = sigmoidf(raw[i]);
    const float g = d_mat[i] * ((i % 5 == 1) ? R_SPAN : 1.f);
    d_raw[i] = g * (1.f - s) * s;
}

struct PointLoss { float rgb, reg, hinge, dl; };

// the per-point loss terms and (GRAD) their gradients scaled by w = grad_out / P (hinge: grad_out * hinge_weight)
template <bool GRAD>
__device__ __forceinline__ PointLoss point_terms(const nero_mat_loss_cfg& c, int P, int has_reg, int p, const float* __restrict__ mat,
                                                 const float* __restrict__ rgb_lin, const float* __restrict__ dl, const float* __restrict__ gt,
                                                 float* __restrict__ rgb_pr, float w, float wh, float* __restrict__ d_mat,
                                                 float* __restrict__ d_rgb, float* __restrict__ d_dl) {
    PointLoss L = {0.f, 0.f, 0.f, 0.f};
    // ---- loss_rgb on the sRGB colour
    float x[3], sr[3], df[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[k] = rgb_lin[3 * p + k];
        sr[k] = srgb(x[k]);
        df[k] = sr[k] - gt[3 * p + k];
        if (!GRAD && rgb_pr) rgb_pr[3 * p + k] = sr[k];
    }
    if (c.rgb_l1) {
        L.rgb = fabsf(df[0]) + fabsf(df[1]) + fabsf(df[2]);
        if (GRAD)
#pragma unroll
            for (int k = 0; k < 3; ++k) d_rgb[3 * p + k] = w * sgn(df[k]) * srgb_grad(x[k]);
    } else {
        L.rgb = sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + 0.001f);
        if (GRAD)
#pragma unroll
            for (int k = 0; k < 3; ++k) d_rgb[3 * p + k] = w * (df[k] / L.rgb) * srgb_grad(x[k]);
    }
    // ---- loss_mat_reg: smoothness against the materials at the perturbed point + the saturation hinge
    float dm[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (c.reg_mat) {
        const float m = mat[5 * p], r = mat[5 * p + 1];
        if (c.reg_change && has_reg) {
            const float* m2 = mat + (size_t)5 * (P + p);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float d = m2[k] - mat[5 * p + k];
                const float wk = k < 2 ? 1.f : (1.f / 3.f);
                acc += fabsf(d) * wk;
                if (GRAD) {
                    const float g = w * c.reg_lambda1 * wk * sgn(d);
                    d_mat[(size_t)5 * (P + p) + k] = g;
                    dm[k] = -g;
                }
            }
            L.reg = acc * c.reg_lambda1;
        }
        if (c.hinge_weight > 0.f) {
            L.hinge = fmaxf(r - H_R_HI, 0.f) + fmaxf(H_R_LO - r, 0.f) + fmaxf(m - H_M_HI, 0.f) + fmaxf(H_M_LO - m, 0.f);
            if (GRAD) {
                dm[1] += wh * ((r - H_R_HI >= 0.f ? 1.f : 0.f) - (H_R_LO - r >= 0.f ? 1.f : 0.f));
                dm[0] += wh * ((m - H_M_HI >= 0.f ? 1.f : 0.f) - (H_M_LO - m >= 0.f ? 1.f : 0.f));
            }
        }
    }
    if (GRAD)
#pragma unroll
        for (int k = 0; k < 5; ++k) d_mat[5 * p + k] = dm[k];
    // ---- loss_diffuse_light: colour neutrality of the clamped sRGB diffuse light
    if (c.reg_diffuse) {
        float y[3], v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            y[k] = srgb(dl[3 * p + k]);
            v[k] = fminf(fmaxf(y[k], 0.f), 1.f);
        }
        const float mean = (v[0] + v[1] + v[2]) / 3.f;
        const float s0 = sgn(v[0] - mean), s1 = sgn(v[1] - mean), s2 = sgn(v[2] - mean);
        L.dl = (fabsf(v[0] - mean) + fabsf(v[1] - mean) + fabsf(v[2] - mean)) * c.reg_diffuse_lambda;
        if (GRAD) {
            const float ss = (s0 + s1 + s2) / 3.f;
            const float sk[3] = {s0, s1, s2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float pass = (y[k] >= 0.f && y[k] <= 1.f) ? 1.f : 0.f;
                d_dl[3 * p + k] = w * c.reg_diffuse_lambda * (sk[k] - ss) * pass * srgb_grad(dl[3 * p + k]);
            }
        }
    } else if (GRAD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d_dl[3 * p + k] = 0.f;
    }
    return L;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {          // fixed-order tree over the 256 threads
    sh[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = LB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(LB) void loss_fwd_kernel(nero_mat_loss_cfg c, int P, int has_reg, const float* __restrict__ mat,
                                                      const float* __restrict__ rgb_lin, const float* __restrict__ dl, const float* __restrict__ gt,
                                                      float* __restrict__ rgb_pr, float* __restrict__ partials) {
    __shared__ float sh[LB];
    const int p = blockIdx.x * LB + threadIdx.x;
    PointLoss L = {0.f, 0.f, 0.f, 0.f};
    if (p < P) L = point_terms<false>(c, P, has_reg, p, mat, rgb_lin, dl, gt, rgb_pr, 0.f, 0.f, nullptr, nullptr, nullptr);
    const float a = block_sum(L.rgb, sh), b = block_sum(L.reg, sh), h = block_sum(L.hinge, sh), d = block_sum(L.dl, sh);
    if (threadIdx.x == 0) {
        float* o = partials + 4 * blockIdx.x;
        o[0] = a; o[1] = b; o[2] = h; o[3] = d;
    }
}
// loss[0] = total, [1] = mean loss_rgb, [2] = mean loss_mat_reg (smoothness + hinge_weight * hinge sum), [3] = mean loss_diffuse_light
__global__ __launch_bounds__(LB) void loss_sum_kernel(nero_mat_loss_cfg c, int P, int nb, const float* __restrict__ partials, float* __restrict__ loss) {
    __shared__ float sh[LB];
    float a = 0.f, b = 0.f, h = 0.f, d = 0.f;
    for (int i = threadIdx.x; i < nb; i += LB) {
        a += partials[4 * i]; b += partials[4 * i + 1]; h += partials[4 * i + 2]; d += partials[4 * i + 3];
    }
    a = block_sum(a, sh); b = block_sum(b, sh); h = block_sum(h, sh); d = block_sum(d, sh);
    if (threadIdx.x == 0) {
        const float inv = 1.f / (float)P;
        loss[1] = a * inv;
        loss[2] = b * inv + c.hinge_weight * h;
        loss[3] = d * inv;
        loss[0] = loss[1] + loss[2] + loss[3];
    }
}
__global__ __launch_bounds__(LB) void loss_bwd_kernel(nero_mat_loss_cfg c, int P, int has_reg, const float* __restrict__ mat,
                                                      const float* __restrict__ rgb_lin, const float* __restrict__ dl, const float* __restrict__ gt,
                                                      const float* __restrict__ grad_out, float* __restrict__ d_mat, float* __restrict__ d_rgb,
                                                      float* __restrict__ d_dl) {
    const int p = blockIdx.x * LB + threadIdx.x;
    if (p >= P) return;
    const float g = grad_out ? grad_out[0] : 1.f;
    point_terms<true>(c, P, has_reg, p, mat, rgb_lin, dl, gt, nullptr, g / (float)P, g * c.hinge_weight, d_mat, d_rgb, d_dl);
}

}  // namespace

extern "C" {

int nero_mat_reg_points(int P, const float* pts, const float* normals, const float* ang01, const float* eps, float eps_const, float* out, void* stream) {
    if (P < 0 || !pts || !normals || !ang01 || !out) return nero_fail(NERO_ERR_ARG, "nero_mat_reg_points: bad argument");
    if (P == 0) return NERO_OK;
    hipLaunchKernelGGL(reg_points_kernel, dim3((P + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, P, pts, normals, ang01, eps, eps_const, out);
    return nero_check_launch("nero_mat_reg_points");
}

int nero_mat_head_fwd(int n, const float* raw, float* mat, void* stream) {
    if (n < 0 || !raw || !mat) return nero_fail(NERO_ERR_ARG, "nero_mat_head_fwd: bad argument");
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((5 * n + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, 5 * n, raw, mat);
    return nero_check_launch("nero_mat_head_fwd");
}

int nero_mat_head_bwd(int n, const float* raw, const float* d_mat, float* d_raw, void* stream) {
    if (n < 0 || !raw || !d_mat || !d_raw) return nero_fail(NERO_ERR_ARG, "nero_mat_head_bwd: bad argument");
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((5 * n + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, 5 * n, raw, d_mat, d_raw);
    return nero_check_launch("nero_mat_head_bwd");
}

int nero_mat_loss_partials(int P) { return P > 0 ? 4 * ((P + LB - 1) / LB) : 4; }

int nero_mat_loss_fwd(const nero_mat_loss_cfg* cfg, int P, int has_reg, const float* mat, const float* rgb_lin, const float* dl, const float* gt,
                      float* rgb_pr, float* partials, float* loss, void* stream) {
    if (!cfg || P <= 0 || !mat || !rgb_lin || !dl || !gt || !partials || !loss) return nero_fail(NERO_ERR_ARG, "nero_mat_loss_fwd: bad argument");
    const int nb = (P + LB - 1) / LB;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(nb), dim3(LB), 0, (hipStream_t)stream, *cfg, P, has_reg, mat, rgb_lin, dl, gt, rgb_pr, partials);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(LB), 0, (hipStream_t)stream, *cfg, P, nb, partials, loss);
    return nero_check_launch("nero_mat_loss_fwd");
}

int nero_mat_loss_bwd(const nero_mat_loss_cfg* cfg, int P, int has_reg, const float* mat, const float* rgb_lin, const float* dl, const float* gt,
                      const float* grad_out, float* d_mat, float* d_rgb_lin, float* d_dl, void* stream) {
    if (!cfg || P <= 0 || !mat || !rgb_lin || !dl || !gt || !d_mat || !d_rgb_lin || !d_dl)
        return nero_fail(NERO_ERR_ARG, "nero_mat_loss_bwd: bad argument");
    if (!(cfg->reg_mat && cfg->reg_change && has_reg) && has_reg)      // rows P..2P-1 of d_mat receive nothing from the kernel
        (void)hipMemsetAsync(d_mat + (size_t)5 * P, 0, sizeof(float) * 5 * (size_t)P, (hipStream_t)stream);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((P + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, *cfg, P, has_reg, mat, rgb_lin, dl, gt, grad_out,
                       d_mat, d_rgb_lin, d_dl);
    return nero_check_launch("nero_mat_loss_bwd");
}

}  // extern "C"
