// encode.hip -- positional encodings and their Jacobian products (gfx950).  HBM-bound elementwise kernels: one thread per
// row, rows are written as whole padded records so the MLP-chain kernels can load them with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "rows.h"

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
__global__ __launch_bounds__(ROW_BLOCK) void pe_vjp_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ e0, int ld0,
                              const float* __restrict__ e1, int ld1, int n_freq, int n, float* __restrict__ out, int ldo) {
    // the <= 40 values of the block's rows (e0 [+ e1]) come in through LDS, consecutive lanes on consecutive floats (rows.h): a thread
    // fetching its own 160-byte row had every cache line re-fetched from HBM -- 959 MB per launch for 96 MB of input with 4-byte
    // loads, still 603 MB with 16-byte loads (rocprofv3 FETCH_SIZE)
    __shared__ float stage[ROW_BLOCK * 41];
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int r = row0 + threadIdx.x;
    const int nv = 3 + 6 * n_freq;                     // <= 40 (checked by the host wrapper)
    rows_load<40>(stage, e0, ld0, 0, e1, ld1, 0, nv, row0, n);
    if (r >= n) return;
    float e[40];
#pragma unroll
    for (int j = 0; j < 40; ++j) e[j] = j < nv ? rows_at<40>(stage, j) : 0.f;
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
__global__ __launch_bounds__(ROW_BLOCK) void pe_jvp_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ t, int ldt, int n_freq,
                              int n, int n_pad, float* __restrict__ out, int ldo) {
    __shared__ float stage[ROW_BLOCK * 41];            // rows leave through LDS, row-major (rows.h); ldo <= 40 (host wrapper)
    const int row0 = blockIdx.x * ROW_BLOCK;
    const int r = row0 + threadIdx.x;
    const bool live = r < n;
    float xv[3] = {0.f, 0.f, 0.f}, tv[3] = {0.f, 0.f, 0.f};
    if (live)
        for (int c = 0; c < 3; ++c) { xv[c] = x[(size_t)r * ldx + c]; tv[c] = t[(size_t)r * ldt + c]; }
    float o[40];
#pragma unroll
    for (int c = 0; c < 40; ++c) o[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = tv[c];
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < n_freq) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 + 6 * k + c] = f * cosf(xv[c] * f) * tv[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 + 6 * k + 3 + c] = -f * sinf(xv[c] * f) * tv[c];
            f *= 2.f;
        }
    }
    rows_put<40, 0, 40>(stage, o, 1.f);                // (rows n .. n_pad-1: tv = 0 -> all-zero rows)
    __syncthreads();
    const int rows = n_pad - row0 < ROW_BLOCK ? n_pad - row0 : ROW_BLOCK;
    for (int idx = threadIdx.x; idx < rows * ldo; idx += ROW_BLOCK) {
        const int rr = idx / ldo, c = idx - rr * ldo;
        out[(size_t)(row0 + rr) * ldo + c] = stage[rr * 41 + c];
    }
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
    hipLaunchKernelGGL(pe_vjp_kernel, dim3((n + ROW_BLOCK - 1) / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, x, ldx, e0, ld0, e1, ld1, n_freq, n, out, ldo);
    return nero_check_launch("nero_pe_vjp");
}

int nero_pe_jvp(const float* x, int ldx, const float* t, int ldt, int n_freq, int n, float* out, int ldo, void* stream) {
    if (!x || !t || !out || n_freq < 0 || n_freq > 6 || 3 * (1 + 2 * n_freq) > ldo || ldo > 40)
        return nero_fail(NERO_ERR_ARG, "nero_pe_jvp: bad argument (n_freq <= 6, 3 (1 + 2 n_freq) <= ldo <= 40)");
    const int n_pad = NERO_ROW_PAD(n);
    if (n_pad == 0) return NERO_OK;
    hipLaunchKernelGGL(pe_jvp_kernel, dim3((n_pad + ROW_BLOCK - 1) / ROW_BLOCK), dim3(ROW_BLOCK), 0, (hipStream_t)stream, x, ldx, t, ldt, n_freq, n, n_pad, out, ldo);
    return nero_check_launch("nero_pe_jvp");
}

}  // extern "C"
