// rows.h -- coalesced global-memory access for kernels in which a thread owns a whole ROW of a row-major matrix.
#pragma once
#include <hip/hip_runtime.h>

// ---- coalesced row stores for thread-per-row encoders ------------------------------------------------------------------------------
// A thread that owns a whole row of NC floats and writes it with scalar stores puts 64 lanes on 64 different cache lines per
// instruction (row pitch 288 ... 512 B); with the unrolled IDE the stores of several output matrices also interleave, and the L2 wrote
// 923 MB for 650 MB of encodings (WRITE_SIZE, shade_encode_kernel).  Instead the block (ROW_BLOCK threads = rows) parks the rows in
// LDS at an odd pitch and writes them back row-major: consecutive lanes -> consecutive floats.  All threads of the block must call it.
constexpr int ROW_BLOCK = 64;
// rows_put<NC, OFF, N>: this thread's row gets v[0..N) at columns OFF.. of an NC-wide staged matrix (pitch NC + 1); rows_flush<NC>:
// barrier, cooperative row-major write of the block's rows to g[(row0 + r) * ld + col0 + c], barrier (the buffer is free again).
template <int NC, int OFF, int N>
__device__ __forceinline__ void rows_put(float* __restrict__ lds, const float (&v)[N], float scale) {
    static_assert(OFF + N <= NC, "rows_put: columns out of range");
#pragma unroll
    for (int c = 0; c < N; ++c) lds[threadIdx.x * (NC + 1) + OFF + c] = v[c] * scale;
}
template <int NC, int OFF, int N>
__device__ __forceinline__ void rows_zero(float* __restrict__ lds) {
#pragma unroll
    for (int c = 0; c < N; ++c) lds[threadIdx.x * (NC + 1) + OFF + c] = 0.f;
}
template <int NC>
__device__ __forceinline__ void rows_flush(const float* __restrict__ lds, float* __restrict__ g, int ld, int col0, int row0, int n_rows_total) {
    __syncthreads();
    const int rows = n_rows_total - row0 < ROW_BLOCK ? n_rows_total - row0 : ROW_BLOCK;
    for (int idx = threadIdx.x; idx < rows * NC; idx += ROW_BLOCK) {
        const int r = idx / NC, c = idx - r * NC;
        g[(size_t)(row0 + r) * ld + col0 + c] = lds[r * (NC + 1) + c];
    }
    __syncthreads();
}

// (the same with fewer columns than the staging pitch holds: nc <= NC)
template <int NC>
__device__ __forceinline__ void rows_flush_cols(const float* __restrict__ lds, float* __restrict__ g, int ld, int col0, int nc, int row0, int n_rows_total) {
    __syncthreads();
    const int rows = n_rows_total - row0 < ROW_BLOCK ? n_rows_total - row0 : ROW_BLOCK;
    for (int idx = threadIdx.x; idx < rows * nc; idx += ROW_BLOCK) {
        const int r = idx / nc, c = idx - r * nc;
        g[(size_t)(row0 + r) * ld + col0 + c] = lds[r * (NC + 1) + c];
    }
    __syncthreads();
}

// The read side: rows_load<NC> brings columns col0 .. col0+NC-1 of the block's rows into the staging buffer with consecutive lanes on
// consecutive floats (optionally adding a second matrix of the same shape), barrier; a thread then reads ITS row with rows_at.  A
// thread reading its own 160 ... 576-byte row with vector loads re-fetched every cache line several times over
// (rocprofv3 FETCH_SIZE: pe_vjp_kernel 603 MB per launch for 96 MB of input).
template <int NC>
__device__ __forceinline__ void rows_load(float* __restrict__ lds, const float* __restrict__ g, int ld, int col0, const float* __restrict__ g2,
                                          int ld2, int col0_2, int nc, int row0, int n_rows_total) {
    __syncthreads();                                   // (earlier readers of the buffer are done)
    const int rows = n_rows_total - row0 < ROW_BLOCK ? n_rows_total - row0 : ROW_BLOCK;
    for (int idx = threadIdx.x; idx < rows * nc; idx += ROW_BLOCK) {
        const int r = idx / nc, c = idx - r * nc;
        float v = g[(size_t)(row0 + r) * ld + col0 + c];
        if (g2) v += g2[(size_t)(row0 + r) * ld2 + col0_2 + c];
        lds[r * (NC + 1) + c] = v;
    }
    __syncthreads();
}
template <int NC>
__device__ __forceinline__ float rows_at(const float* __restrict__ lds, int c) { return lds[threadIdx.x * (NC + 1) + c]; }
