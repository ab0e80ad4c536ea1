// encode.hip -- positional encodings and their Jacobian products (gfx950).  HBM-bound elementwise kernels: one thread per
// row, rows are written as whole padded records so the MLP-chain kernels can load them with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

// out[r] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x), 0-pad]   (network/field.py:14-58)
__global__ void encode_pe_kernel(const float* __restrict__ x, int ldx, int dim, int n_freq, int n, int n_pad,
                                 float* __restrict__ out, int ldo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float* o = out + (size_t)r * ldo;
    if (r >= n) { for (int c = 0; c < ldo; ++c) o[c] = 0.f; return; }
    float v[4];
    for (int c = 0; c < dim; ++c) { v[c] = x[(size_t)r * ldx + c]; o[c] = v[c]; }
    int p = dim;
    float f = 1.f;
    for (int k = 0; k < n_freq; ++k) {
        for (int c = 0; c < dim; ++c) o[p + c] = sinf(v[c] * f);
        for (int c = 0; c < dim; ++c) o[p + dim + c] = cosf(v[c] * f);
        p += 2 * dim;
        f *= 2.f;
    }
    for (; p < ldo; ++p) o[p] = 0.f;
}

// normal = J_e^T (ebar0 + ebar1):  n_c = e_c + sum_k 2^k (cos(2^k x_c) e_sin[k,c] - sin(2^k x_c) e_cos[k,c])
__global__ void pe_vjp_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ e0, int ld0,
                              const float* __restrict__ e1, int ld1, int n_freq, int n, float* __restrict__ out, int ldo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* a = e0 + (size_t)r * ld0;
    const float* b = e1 ? e1 + (size_t)r * ld1 : nullptr;
    // the <= 40 values of the row(s) are fetched first, as 16-byte loads where the layout allows: interleaved with the
    // trigonometric code below, the 4-byte loads of a 160-byte row were spread over so many cycles that every cache line was
    // re-fetched from HBM ~10 times (rocprofv3 FETCH_SIZE: 959 MB per launch for 96 MB of input)
    const int nv = 3 + 6 * n_freq;                     // <= 40 (checked by the host wrapper)
    float e[40];
    const bool vec = ((ld0 | (e1 ? ld1 : 0)) & 3) == 0 && (((uintptr_t)e0 | (uintptr_t)e1) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (4 * j < nv) {
                float4 va = *reinterpret_cast<const float4*>(a + 4 * j);
                if (b) { const float4 vb = *reinterpret_cast<const float4*>(b + 4 * j); va.x += vb.x; va.y += vb.y; va.z += vb.z; va.w += vb.w; }
                e[4 * j] = va.x; e[4 * j + 1] = va.y; e[4 * j + 2] = va.z; e[4 * j + 3] = va.w;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 40; ++j)
            if (j < nv) e[j] = a[j] + (b ? b[j] : 0.f);
    }
    for (int c = 0; c < 3; ++c) {
        const float xc = x[(size_t)r * ldx + c];
        float acc = e[c];
        float f = 1.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k < n_freq) {
                const float es = e[3 + 6 * k + c], ec = e[3 + 6 * k + 3 + c];
                acc += f * (cosf(xc * f) * es - sinf(xc * f) * ec);
                f *= 2.f;
            }
        }
        out[(size_t)r * ldo + c] = acc;
    }
}

// ehat = J_e nbar:  [nbar_c, 2^k cos(2^k x_c) nbar_c, -2^k sin(2^k x_c) nbar_c, ... , 0-pad]; rows >= n are zero
__global__ void pe_jvp_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ t, int ldt, int n_freq,
                              int n, int n_pad, float* __restrict__ out, int ldo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float* o = out + (size_t)r * ldo;
    if (r >= n) { for (int c = 0; c < ldo; ++c) o[c] = 0.f; return; }
    float xv[3], tv[3];
    for (int c = 0; c < 3; ++c) { xv[c] = x[(size_t)r * ldx + c]; tv[c] = t[(size_t)r * ldt + c]; o[c] = tv[c]; }
    int p = 3;
    float f = 1.f;
    for (int k = 0; k < n_freq; ++k) {
        for (int c = 0; c < 3; ++c) o[p + c] = f * cosf(xv[c] * f) * tv[c];
        for (int c = 0; c < 3; ++c) o[p + 3 + c] = -f * sinf(xv[c] * f) * tv[c];
        p += 6;
        f *= 2.f;
    }
    for (; p < ldo; ++p) o[p] = 0.f;
}

}  // namespace

extern "C" {

int nero_encode_pe(const float* x, int ldx, int dim, int n_freq, int n, float* out, int ldo, void* stream) {
    if (!x || !out || dim < 1 || dim > 4 || dim * (1 + 2 * n_freq) > ldo) return nero_fail(NERO_ERR_ARG, "nero_encode_pe: bad argument");
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(encode_pe_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, dim, n_freq, n, n_pad, out, ldo);
    return nero_check_launch("nero_encode_pe");
}

int nero_pe_vjp(const float* x, int ldx, const float* e0, int ld0, const float* e1, int ld1, int n_freq, int n,
                float* out, int ldo, void* stream) {
    if (!x || !e0 || !out || n_freq < 0 || n_freq > 6) return nero_fail(NERO_ERR_ARG, "nero_pe_vjp: bad argument (n_freq <= 6)");
    if (n == 0) return NERO_OK;
    hipLaunchKernelGGL(pe_vjp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, e0, ld0, e1, ld1, n_freq, n, out, ldo);
    return nero_check_launch("nero_pe_vjp");
}

int nero_pe_jvp(const float* x, int ldx, const float* t, int ldt, int n_freq, int n, float* out, int ldo, void* stream) {
    if (!x || !t || !out || 3 * (1 + 2 * n_freq) > ldo) return nero_fail(NERO_ERR_ARG, "nero_pe_jvp: bad argument");
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(pe_jvp_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, t, ldt, n_freq, n, n_pad, out, ldo);
    return nero_check_launch("nero_pe_jvp");
}

}  // extern "C"
