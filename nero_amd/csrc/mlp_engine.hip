// mlp_engine.hip -- fused MLP-chain kernels for gfx950 (MI355X): forward, tangent (forward-mode), reverse and
// weight-gradient GEMM, all on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s dense peak).
//
// Design: one 256-thread workgroup owns a tile of 64 rows (ray samples) and walks the whole layer list
// of one network with the activations resident in LDS ([64][260] fp32, row stride 260 -> conflict-free ds_read_b128
// A fragments).  The four waves split the output columns (2 column tiles of 32 each); B fragments (weights) are read
// straight from the L2-resident packed weight image with one coalesced 16-byte load per lane per 8 k-values -- weights
// are never shared between waves, so staging them through LDS would buy nothing.  Two workgroups per CU overlap one
// tile's VALU epilogue (bias, softplus, stores) with the other's MFMA stream.
//
// k-ordering inside a chunk of 8: MFMA t (t=0..3) consumes k = 8c + 4h + t from lane half h, which makes the A fragment
// 4 contiguous floats per lane (one ds_read_b128) and the packed B image a plain float4 per lane.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDA = 260;          // activation tile row stride (floats)
constexpr int LDX_NARROW = 44;    // aux tile row stride, <= 40 columns
constexpr int LDX_WIDE = 92;      // aux tile row stride, <= 88 columns
constexpr float BETA = 100.0f;

struct WaveTiles { int ct0, n; };

__device__ __forceinline__ WaveTiles wave_tiles(int nt, int wave) {
    int per = nt > 4 ? 2 : 1;
    int ct0 = wave * per;
    int n = nt - ct0;
    n = n < 0 ? 0 : (n > per ? per : n);
    return {ct0, n};
}

// acc[rt][ct] += A[64 x 8*nchunks] (LDS, stride lds) * B (packed), for this wave's column tiles.
// Software pipeline: operands of chunk c+2 (B from the L2-resident packed image, A from LDS) are requested before the
// 16 MFMAs of chunk c are issued, so neither the ~500-cycle L2 latency nor the LDS latency sits on the MFMA stream
// (three rotating register buffers, static indexing by a 3x unrolled body).
struct OpBuf { float4 a0, a1, b0, b1; };

#define NERO_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0)

template <int NCT>
__device__ __forceinline__ void op_load(OpBuf& o, const float* a0p, const float* a1p, const float4* bp, size_t bstride, int c) {
    o.b0 = bp[c * bstride];
    if (NCT == 2) o.b1 = bp[c * bstride + 64];
    o.a0 = *reinterpret_cast<const float4*>(a0p + 8 * c);
    o.a1 = *reinterpret_cast<const float4*>(a1p + 8 * c);
}

template <int NCT>
__device__ __forceinline__ void op_compute(f32x16 (&acc)[2][2], const OpBuf& o) {
    NERO_MFMA(acc[0][0], o.a0.x, o.b0.x); if (NCT == 2) NERO_MFMA(acc[0][1], o.a0.x, o.b1.x);
    NERO_MFMA(acc[1][0], o.a1.x, o.b0.x); if (NCT == 2) NERO_MFMA(acc[1][1], o.a1.x, o.b1.x);
    NERO_MFMA(acc[0][0], o.a0.y, o.b0.y); if (NCT == 2) NERO_MFMA(acc[0][1], o.a0.y, o.b1.y);
    NERO_MFMA(acc[1][0], o.a1.y, o.b0.y); if (NCT == 2) NERO_MFMA(acc[1][1], o.a1.y, o.b1.y);
    NERO_MFMA(acc[0][0], o.a0.z, o.b0.z); if (NCT == 2) NERO_MFMA(acc[0][1], o.a0.z, o.b1.z);
    NERO_MFMA(acc[1][0], o.a1.z, o.b0.z); if (NCT == 2) NERO_MFMA(acc[1][1], o.a1.z, o.b1.z);
    NERO_MFMA(acc[0][0], o.a0.w, o.b0.w); if (NCT == 2) NERO_MFMA(acc[0][1], o.a0.w, o.b1.w);
    NERO_MFMA(acc[1][0], o.a1.w, o.b0.w); if (NCT == 2) NERO_MFMA(acc[1][1], o.a1.w, o.b1.w);
}

template <int NCT>
__device__ __forceinline__ void gemm_pipe(f32x16 (&acc)[2][2], const float* a0p, const float* a1p, const float4* bp,
                                          size_t bstride, int n) {
    OpBuf u, v, w;
    const int last = n - 1;
    op_load<NCT>(u, a0p, a1p, bp, bstride, 0);
    op_load<NCT>(v, a0p, a1p, bp, bstride, 1 < last ? 1 : last);
    int c = 0;
    // sched_barrier(0): hipcc otherwise sinks each prefetch down to its first use and re-serialises the loop
    __builtin_amdgcn_sched_barrier(0);
    for (; c + 3 <= n; c += 3) {
        op_load<NCT>(w, a0p, a1p, bp, bstride, c + 2 < last ? c + 2 : last);
        __builtin_amdgcn_sched_barrier(0);
        op_compute<NCT>(acc, u);
        __builtin_amdgcn_sched_barrier(0);
        op_load<NCT>(u, a0p, a1p, bp, bstride, c + 3 < last ? c + 3 : last);
        __builtin_amdgcn_sched_barrier(0);
        op_compute<NCT>(acc, v);
        __builtin_amdgcn_sched_barrier(0);
        op_load<NCT>(v, a0p, a1p, bp, bstride, c + 4 < last ? c + 4 : last);
        __builtin_amdgcn_sched_barrier(0);
        op_compute<NCT>(acc, w);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c < n) op_compute<NCT>(acc, u);
    if (c + 1 < n) op_compute<NCT>(acc, v);
}

__device__ __forceinline__ void gemm_part(f32x16 (&acc)[2][2], const float* tile, int lds, int nchunks,
                                          const float* __restrict__ wpack, int nt, WaveTiles wt, int lane) {
    if (nchunks <= 0 || wt.n == 0) return;
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = tile + i * lds + 4 * h;
    const float* a1p = a0p + 32 * lds;
    const float4* bp = reinterpret_cast<const float4*>(wpack) + (size_t)wt.ct0 * 64 + lane;
    const size_t bstride = (size_t)nt * 64;          // float4 per chunk
    if (wt.n == 2) gemm_pipe<2>(acc, a0p, a1p, bp, bstride, nchunks);
    else gemm_pipe<1>(acc, a0p, a1p, bp, bstride, nchunks);
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
}

// C/D layout of the 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
#define FOR_EACH_ACC(WT, BODY)                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {          \
        if (ct < (WT).n) {                                                                       \
            const int col = 32 * ((WT).ct0 + ct) + (lane & 31);                                  \
            _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                     \
                const int row = 32 * rt + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);              \
                float x = acc[rt][ct][v];                                                        \
                BODY                                                                             \
                acc[rt][ct][v] = x;                                                              \
            }                                                                                    \
        }                                                                                        \
    }

// softplus(beta=100, threshold=20) = max(x,0) + log1p(exp(-|beta x|)) / beta  on the hardware exp2/log2 units.
// u = exp(-|bx|) in (0,1]; log1p(u) by a 5-term series below 2^-4 (rel. error < 2e-8) and log(1+u) above (1+u is exact
// enough: u >= 2^-4).  Measured against torch.nn.functional.softplus in fp64: rel. error <= 3e-7 over [-0.5, 0.5].
__device__ __forceinline__ float log1p_small(float u) {
    return u * (1.f + u * (-0.5f + u * (0.33333334f + u * (-0.25f + u * 0.2f))));
}
__device__ __forceinline__ float softplus100(float x) {
    const float bx = BETA * x;
    if (bx > 20.f) return x;
    const float u = __expf(-fabsf(bx));
    const float l = u < 0.0625f ? log1p_small(u) : __logf(1.f + u);
    return fmaxf(x, 0.f) + l * (1.0f / BETA);
}
// sigma'(z) recovered from a = softplus(z):  1 - exp(-beta a)  (== 1 in torch's linear region beta z > 20)
__device__ __forceinline__ float softplus100_grad_from_out(float a) {
    const float ba = BETA * a;
    if (ba > 20.f) return 1.f;
    if (ba < 0.03125f) return ba * (1.f + ba * (-0.5f + ba * (0.16666667f + ba * (-0.041666668f))));   // -expm1(-ba)
    return 1.f - __expf(-ba);
}

// copy [64][k] from a row-major global matrix into an LDS tile (k multiple of 4, rows clamped to n_rows-1)
__device__ __forceinline__ void load_tile(float* tile, int lds, const float* __restrict__ src, int ld, int k, int row0,
                                          int n_rows, int tid) {
    const int k4 = k >> 2;
    for (int idx = tid; idx < 64 * k4; idx += 256) {
        int r = idx / k4, c4 = idx - r * k4;
        int gr = row0 + r;
        gr = gr < n_rows ? gr : (n_rows - 1);
        float4 v = *reinterpret_cast<const float4*>(src + (size_t)gr * ld + 4 * c4);
        *reinterpret_cast<float4*>(tile + r * lds + 4 * c4) = v;
    }
}

// VALU head on the current activation tile: out[r][j] = b[j] + sum_k tile[r][k] * W[j][k], k < hk
__device__ __forceinline__ void eval_head(const float* tile, const float* __restrict__ w, const float* __restrict__ b,
                                          float* __restrict__ out, int n_head, int hk, int row0, int tid) {
    const int r = tid >> 2, q = tid & 3;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* arow = tile + r * LDA + q;
    for (int i = 0; i < (hk >> 2); ++i) {
        float a = arow[4 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_head) s[j] = fmaf(a, w[j * NERO_HID + 4 * i + q], s[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] += __shfl_xor(s[j], 1);
        s[j] += __shfl_xor(s[j], 2);
    }
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_head) out[(size_t)(row0 + r) * 4 + j] = s[j] + (b ? b[j] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue scheme shared by the three chain kernels: (1) every wave drops its raw accumulators into the LDS tile (C
// layout, immediate-offset ds_write_b32), (2) barrier, (3) a ROLLED row-major pass -- thread t owns columns 4(t&63)..+3 of
// rows it*4 + (t>>6) -- applies the elementwise math with 16-byte LDS accesses and fully coalesced 16-byte global loads /
// stores (one 1 KiB row per wave instruction).  Keeps the epilogue at ~40 VGPRs instead of unrolling 64 scalar
// load/compute/store strands per lane.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dump_acc(f32x16 (&acc)[2][2], float* tile, WaveTiles wt, int lane, const float* __restrict__ bias) {
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {
        if (ct < wt.n) {
            const int col = 32 * (wt.ct0 + ct) + (lane & 31);
            const float b = bias ? bias[col] : 0.f;
            _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)
                _Pragma("unroll") for (int v = 0; v < 16; ++v) {
                    const int row = 32 * rt + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                    tile[row * LDA + col] = acc[rt][ct][v] + b;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward chain
// ---------------------------------------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(nero_fwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDX = WIDE ? LDX_WIDE : LDX_NARROW;
    float* act = smem;
    float* aux = smem + 64 * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 64;
    if (ch.init) load_tile(act, LDA, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
    if (ch.aux) load_tile(aux, LDX, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_fwd_layer& L = ch.layer[l];
        if (L.n_head > 0) eval_head(act, L.head_w, L.head_b, L.head_out, L.n_head, L.head_k, row0, tid);
        if (L.n_tiles == 0) continue;
        {
            f32x16 acc[2][2];
            zero_acc(acc);
            const WaveTiles wt = wave_tiles(L.n_tiles, wave);
            gemm_part(acc, act, LDA, L.k_main >> 3, L.w_main, L.n_tiles, wt, lane);
            gemm_part(acc, aux, LDX, L.k_aux >> 3, L.w_aux, L.n_tiles, wt, lane);
            __syncthreads();                               // every wave is done reading the input tile
            dump_acc(acc, act, wt, lane, L.bias);
        }
        __syncthreads();
        const int ncols = 32 * L.n_tiles, c4 = 4 * lane;
        float* __restrict__ save = L.save;
        const int actk = L.act;
        if (c4 < ncols && (actk != NERO_ACT_NONE || save)) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = 4 * it + wave;
                float4 y = *reinterpret_cast<float4*>(act + row * LDA + c4);
                if (actk == NERO_ACT_RELU) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                else if (actk == NERO_ACT_SOFTPLUS100) { y.x = softplus100(y.x); y.y = softplus100(y.y); y.z = softplus100(y.z); y.w = softplus100(y.w); }
                if (actk != NERO_ACT_NONE) *reinterpret_cast<float4*>(act + row * LDA + c4) = y;
                if (save) *reinterpret_cast<float4*>(save + (size_t)(row0 + row) * NERO_HID + c4) = y;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tangent chain (softplus networks):  adot_l = s_l * (W_l adot_{l-1}),  inj_l = gbar_l * beta (1-s_l) * zdot_l
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tan_elem(float a, float zd, float gb, bool live, float& ad, float& ij) {
    const float s = softplus100_grad_from_out(a);
    ad = s * zd;
    // sigma''/sigma' = beta (1 - s); zero in the linear region (torch: softplus double-backward is 0 there)
    const float r2 = (BETA * a > 20.f) ? 0.f : BETA * (1.f - s);
    ij = live ? gb * r2 * zd : 0.f;
}

template <bool WIDE>
__global__ __launch_bounds__(256, 2) void mlp_tan_kernel(nero_tan_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDX = WIDE ? LDX_WIDE : LDX_NARROW;
    float* act = smem;
    float* aux = smem + 64 * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 64;
    if (ch.init) load_tile(act, LDA, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
    if (ch.aux) load_tile(aux, LDX, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_tan_layer& L = ch.layer[l];
        // saved activations / normal-pass gradients for the row-major epilogue: first halves requested before the GEMM (latency
        // hidden behind the MFMA stream), second halves at the top of the epilogue; row bases are wave-uniform
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const size_t rbase = (size_t)(row0 + wv) * NERO_HID;
        const int ncols = 32 * L.n_tiles, c4 = 4 * lane;
        const float* __restrict__ asv = L.a_saved + rbase + c4;
        const float* __restrict__ gbp = L.gbar + rbase + c4;
        float4 pa[8], pg[8];
        if (c4 < ncols) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                pa[it] = *reinterpret_cast<const float4*>(asv + it * 4 * NERO_HID);
                pg[it] = *reinterpret_cast<const float4*>(gbp + it * 4 * NERO_HID);
            }
        }
        {
            f32x16 acc[2][2];
            zero_acc(acc);
            const WaveTiles wt = wave_tiles(L.n_tiles, wave);
            gemm_part(acc, act, LDA, L.k_main >> 3, L.w_main, L.n_tiles, wt, lane);
            gemm_part(acc, aux, LDX, L.k_aux >> 3, L.w_aux, L.n_tiles, wt, lane);
            __syncthreads();
            dump_acc(acc, act, wt, lane, nullptr);
        }
        __syncthreads();
        float* __restrict__ adot = L.adot + rbase + c4;
        float* __restrict__ inj = L.inj + rbase + c4;
        float* actw = act + wv * LDA + c4;
        if (c4 < ncols) {
            float4 qa[8], qg[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                qa[it] = *reinterpret_cast<const float4*>(asv + (it + 8) * 4 * NERO_HID);
                qg[it] = *reinterpret_cast<const float4*>(gbp + (it + 8) * 4 * NERO_HID);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const float4 a = it < 8 ? pa[it & 7] : qa[it & 7];
                const float4 gb = it < 8 ? pg[it & 7] : qg[it & 7];
                const float4 zd = *reinterpret_cast<const float4*>(actw + it * 4 * LDA);
                const bool live = (row0 + wv + 4 * it) < n_rows;
                float4 ad, ij;
                tan_elem(a.x, zd.x, gb.x, live, ad.x, ij.x);
                tan_elem(a.y, zd.y, gb.y, live, ad.y, ij.y);
                tan_elem(a.z, zd.z, gb.z, live, ad.z, ij.z);
                tan_elem(a.w, zd.w, gb.w, live, ad.w, ij.w);
                *reinterpret_cast<float4*>(actw + it * 4 * LDA) = ad;
                *reinterpret_cast<float4*>(adot + it * 4 * NERO_HID) = live ? ad : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(inj + it * 4 * NERO_HID) = ij;
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse chain
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float a, float g, int actp) {
    if (actp == NERO_ACT_RELU) return a > 0.f ? g : 0.f;
    if (actp == NERO_ACT_SOFTPLUS100) return g * softplus100_grad_from_out(a);
    return g;
}

template <bool WIDE>
__global__ __launch_bounds__(256, 2) void mlp_bwd_kernel(nero_bwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 64;
    if (ch.dy) load_tile(act, LDA, ch.dy, ch.ld_dy, ch.k_dy, row0, n_rows, tid);
    __syncthreads();
    for (int l = ch.n_layers - 1; l >= 0; --l) {
        const nero_bwd_layer& L = ch.layer[l];
        const bool first = (L.a_prev == nullptr);
        if (first && ch.d_init == nullptr && !(ch.d_aux && L.w_aux_t)) break;
        const int nt = L.k_main_tiles;
        // the saved activations this thread's row-major epilogue needs (16 rows x 16 bytes).  The first half is requested here,
        // before the GEMM, so that its HBM latency hides behind the MFMA stream; the second half at the top of the epilogue
        // (once the accumulators are dead) and is consumed after the first half.  Row bases are wave-uniform (SGPR), the
        // per-thread part is just the column offset.
        float4 apf[8];
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const size_t rbase = (size_t)(row0 + wv) * NERO_HID;
        if (!first && 4 * lane < 32 * nt) {
            const float* __restrict__ ap0 = L.a_prev + rbase;
#pragma unroll
            for (int it = 0; it < 8; ++it) apf[it] = *reinterpret_cast<const float4*>(ap0 + it * 4 * NERO_HID + 4 * lane);
        }
        float4 ijf[4];
        const bool has_inj = !first && L.inj != nullptr;
        if (has_inj && 4 * lane < 32 * nt) {
            const float* __restrict__ ij0 = L.inj + rbase;
#pragma unroll
            for (int it = 0; it < 4; ++it) ijf[it] = *reinterpret_cast<const float4*>(ij0 + it * 4 * NERO_HID + 4 * lane);
        }
        if (L.n_out > 0) {
            f32x16 acc[2][2];
            // gradient w.r.t. the aux columns of this layer (skip connections), written straight out; done first so that
            // only one accumulator set is ever live
            if (ch.d_aux && L.w_aux_t) {
                zero_acc(acc);
                const WaveTiles wx = wave_tiles(L.k_aux_tiles, wave);
                gemm_part(acc, act, LDA, L.n_out >> 3, L.w_aux_t, L.k_aux_tiles, wx, lane);
                // through the (otherwise unused) aux LDS region, then coalesced rows to d_aux
                float* auxr = smem + 64 * LDA;
                FOR_EACH_ACC(wx, { if (col < LDX_NARROW) auxr[row * LDX_NARROW + col] = x; })
                __syncthreads();
                const int l4 = ch.ld_daux >> 2;
                for (int idx = tid; idx < 64 * l4; idx += 256) {
                    const int r = idx / l4, c4 = (idx - r * l4) * 4;
                    *reinterpret_cast<float4*>(ch.d_aux + (size_t)(row0 + r) * ch.ld_daux + c4) = *reinterpret_cast<const float4*>(auxr + r * LDX_NARROW + c4);
                }
            }
            zero_acc(acc);
            const WaveTiles wt = wave_tiles(nt, wave);
            gemm_part(acc, act, LDA, L.n_out >> 3, L.w_main_t, nt, wt, lane);
            __syncthreads();
            dump_acc(acc, act, wt, lane, nullptr);
            __syncthreads();
            if (first) {
                if (ch.d_init) {
                    const int ldi = ch.ld_dinit, l4 = ldi >> 2;
                    const bool accum = ch.accumulate_dinit != 0;
                    for (int idx = tid; idx < 64 * l4; idx += 256) {
                        const int r = idx / l4, c4 = (idx - r * l4) * 4;
                        if (c4 < 32 * nt) {
                            float4 v = *reinterpret_cast<const float4*>(act + r * LDA + c4);
                            float4* dst = reinterpret_cast<float4*>(ch.d_init + (size_t)(row0 + r) * ldi + c4);
                            if (accum) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            *dst = v;
                        }
                    }
                }
                break;
            }
        } else if (!ch.dy) {
            // head-only pseudo layer without an upstream tile: start from zero
            for (int idx = tid; idx < 64 * 64; idx += 256) {
                const int r = idx >> 6, c4 = (idx & 63) * 4;
                *reinterpret_cast<float4*>(act + r * LDA + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
        }
        if (first) break;
        // row-major pass: delta_prev = (g [+ dy_head W_head]) * act'(a_prev) [+ inj], rows >= n_rows zeroed
        const int ncols = 32 * nt, c4 = 4 * lane;
        const float* __restrict__ apb = L.a_prev + rbase;
        const float* __restrict__ injb = L.inj ? L.inj + rbase : nullptr;
        float* __restrict__ dpb = L.delta_prev ? L.delta_prev + rbase : nullptr;
        const float* __restrict__ hdy = L.head_dy ? L.head_dy + (size_t)(row0 + wv) * 4 : nullptr;
        float* actw = act + wv * LDA;
        const int nh = L.n_head, actp = L.act_prev;
        if (c4 < ncols) {
            float4 hw[4];
            for (int j = 0; j < 4; ++j) hw[j] = (j < nh) ? *reinterpret_cast<const float4*>(L.head_w + j * NERO_HID + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 apg[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) apg[it] = *reinterpret_cast<const float4*>(apb + (it + 8) * 4 * NERO_HID + c4);
            float4 ijg[12];
            if (has_inj) {
#pragma unroll
                for (int it = 0; it < 12; ++it) ijg[it] = *reinterpret_cast<const float4*>(injb + (it + 4) * 4 * NERO_HID + c4);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const float4 a = it < 8 ? apf[it & 7] : apg[it & 7];
                float4 gs = *reinterpret_cast<const float4*>(actw + it * 4 * LDA + c4);
                if (nh > 0) {
                    const float4 dyh = *reinterpret_cast<const float4*>(hdy + it * 16);
                    const float dj[4] = {dyh.x, dyh.y, dyh.z, dyh.w};
                    for (int j = 0; j < 4; ++j) {
                        if (j < nh) {
                            gs.x = fmaf(dj[j], hw[j].x, gs.x); gs.y = fmaf(dj[j], hw[j].y, gs.y);
                            gs.z = fmaf(dj[j], hw[j].z, gs.z); gs.w = fmaf(dj[j], hw[j].w, gs.w);
                        }
                    }
                }
                float4 d;
                d.x = act_grad(a.x, gs.x, actp); d.y = act_grad(a.y, gs.y, actp);
                d.z = act_grad(a.z, gs.z, actp); d.w = act_grad(a.w, gs.w, actp);
                if (has_inj) {
                    const float4 ij = it < 4 ? ijf[it & 3] : ijg[it < 4 ? 0 : it - 4];
                    d.x += ij.x; d.y += ij.y; d.z += ij.z; d.w += ij.w;
                }
                if (row0 + wv + 4 * it >= n_rows) d = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(actw + it * 4 * LDA + c4) = d;
                if (dpb) *reinterpret_cast<float4*>(dpb + it * 4 * NERO_HID + c4) = d;
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the unrolled pass from hoisting all 16 rows' loads
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight-gradient GEMM:  C[n][k] = sum_r D[r][n] B[r][k]   (split over row slices, fixed-order reduction)
// 512 threads = 8 waves in a 4(n) x 2(k) grid; each wave owns a 64x128 block = 2x4 tiles of 32x32 (128 acc VGPRs).
// Row chunks of 32 are staged through LDS with coalesced 16-byte loads.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DW_RC = 32;           // rows per staged chunk
constexpr int DW_LD = 256;
constexpr int DW_BUF = DW_RC * DW_LD;   // floats per staged matrix chunk

// rows [r0, r0+32) of src as 4 float4 per thread (512 threads x 4 x 4 floats = 32 x 256); columns >= cols and rows >= r1 are zero
__device__ __forceinline__ void dw_fetch(float4 (&v)[4], const float* __restrict__ src, int ld, int cols, int r0, int r1, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 512 * q;
        const int r = idx >> 6, c4 = (idx & 63) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gr = r0 + r;
        if (gr < r1 && c4 < cols) {
            const float* p = src + (size_t)gr * ld + c4;
            if (c4 + 3 < cols) x = *reinterpret_cast<const float4*>(p);
            else { x.x = p[0]; if (c4 + 1 < cols) x.y = p[1]; if (c4 + 2 < cols) x.z = p[2]; }
        }
        v[q] = x;
    }
}
__device__ __forceinline__ void dw_put(float* dst, const float4 (&v)[4], int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 512 * q;
        *reinterpret_cast<float4*>(dst + (idx >> 6) * DW_LD + (idx & 63) * 4) = v[q];
    }
}

// 512 threads = 8 waves in a 4(n) x 2(k) grid, 64x128 block per wave (2x4 tiles, 128 accumulator VGPRs).  Row chunks of 32
// are double-buffered in LDS: the next chunk's 16-byte global loads are issued before the 128 MFMAs of the current chunk
// and stored to the other buffer afterwards -- one barrier per chunk, HBM latency hidden behind the MFMA stream.
// NARROW (k_pad <= 128: first layers, skip/aux column blocks): the 8 waves split n eight ways (one 32-row tile each) and
// multiply only the k-tiles that exist, and only 128 columns of B are staged -- such jobs are then bound by streaming D, not
// by MFMAs on zero padding.
__device__ __forceinline__ void dw_fetch_narrow(float4 (&v)[4], const float* __restrict__ src, int ld, int cols, int r0, int r1, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + 512 * q;
        const int r = idx >> 5, c4 = (idx & 31) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gr = r0 + r;
        if (gr < r1 && c4 < cols) {
            const float* p = src + (size_t)gr * ld + c4;
            if (c4 + 3 < cols) x = *reinterpret_cast<const float4*>(p);
            else { x.x = p[0]; if (c4 + 1 < cols) x.y = p[1]; if (c4 + 2 < cols) x.z = p[2]; }
        }
        v[q] = x;
    }
}
__device__ __forceinline__ void dw_put_narrow(float* dst, const float4 (&v)[4], int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + 512 * q;
        *reinterpret_cast<float4*>(dst + (idx >> 5) * DW_LD + (idx & 31) * 4) = v[q];
    }
}

template <bool NARROW>
__global__ __launch_bounds__(512, 2) void dw_gemm_kernel(nero_dw_job job, int n_rows, int rows_per_slice, float* __restrict__ partials,
                                                          int n_pad, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = NARROW ? wave : wave >> 1, wk = NARROW ? 0 : wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int r_begin = blockIdx.x * rows_per_slice;
    int r_end = r_begin + rows_per_slice;
    r_end = r_end < n_rows ? r_end : n_rows;
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float bsum = 0.f;                    // bias gradient: thread tid < 256 owns column tid of D0
    const int n_tiles = n_pad >> 5, k_tiles = k_pad >> 5;
    const bool wave_live = NARROW ? (wn < n_tiles) : ((2 * wn < n_tiles) && (4 * wk < k_tiles));
    const int nch = r_end > r_begin ? (r_end - r_begin + DW_RC - 1) / DW_RC : 0;
    const int total = nch * (job.d1 ? 2 : 1);
    float4 pd[4], pb[4];
    if (total > 0) {
        dw_fetch(pd, job.d0, job.ldd0, job.n_out, r_begin, r_end, tid);
        if (NARROW) dw_fetch_narrow(pb, job.b0, job.ldb0, job.k_cols, r_begin, r_end, tid);
        else dw_fetch(pb, job.b0, job.ldb0, job.k_cols, r_begin, r_end, tid);
        dw_put(smem, pd, tid);
        if (NARROW) dw_put_narrow(smem + DW_BUF, pb, tid);
        else dw_put(smem + DW_BUF, pb, tid);
    }
    __syncthreads();
    for (int q = 0; q < total; ++q) {
        float* sD = smem + (q & 1) * 2 * DW_BUF;
        float* sB = sD + DW_BUF;
        const bool more = q + 1 < total;
        if (more) {
            const int q1 = q + 1;
            const bool second = q1 >= nch;
            const int r0 = r_begin + (second ? q1 - nch : q1) * DW_RC;
            dw_fetch(pd, second ? job.d1 : job.d0, second ? job.ldd1 : job.ldd0, job.n_out, r0, r_end, tid);
            if (NARROW) dw_fetch_narrow(pb, second ? job.b1 : job.b0, second ? job.ldb1 : job.ldb0, job.k_cols, r0, r_end, tid);
            else dw_fetch(pb, second ? job.b1 : job.b0, second ? job.ldb1 : job.ldb0, job.k_cols, r0, r_end, tid);
        }
        if (q < nch && tid < 256) {
#pragma unroll 8
            for (int r = 0; r < DW_RC; ++r) bsum += sD[r * DW_LD + tid];
        }
        if (wave_live) {
            if (NARROW) {
#pragma unroll 4
                for (int j = 0; j < DW_RC / 2; ++j) {
                    const float* dr = sD + (2 * j + h) * DW_LD + 32 * wn + i;
                    const float* br = sB + (2 * j + h) * DW_LD + i;
                    const float a0 = dr[0];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, br[0], acc[0][0], 0, 0, 0);
                    if (k_tiles > 1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, br[32], acc[0][1], 0, 0, 0);
                    if (k_tiles > 2) acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, br[64], acc[0][2], 0, 0, 0);
                    if (k_tiles > 3) acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, br[96], acc[0][3], 0, 0, 0);
                }
            } else {
#pragma unroll 4
                for (int j = 0; j < DW_RC / 2; ++j) {
                    const float* dr = sD + (2 * j + h) * DW_LD + 64 * wn + i;
                    const float* br = sB + (2 * j + h) * DW_LD + 128 * wk + i;
                    float a0 = dr[0], a1 = dr[32];
                    float b0 = br[0], b1 = br[32], b2 = br[64], b3 = br[96];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[0][2], 0, 0, 0);
                    acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b3, acc[0][3], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[1][2], 0, 0, 0);
                    acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b3, acc[1][3], 0, 0, 0);
                }
            }
        }
        if (more) {
            float* nD = smem + ((q + 1) & 1) * 2 * DW_BUF;
            dw_put(nD, pd, tid);
            if (NARROW) dw_put_narrow(nD + DW_BUF, pb, tid);
            else dw_put(nD + DW_BUF, pb, tid);
        }
        __syncthreads();
    }
    // write this slice's partial C (row-major [n_pad][k_pad]) and bias partial
    float* __restrict__ P = partials + (size_t)blockIdx.x * ((size_t)n_pad * k_pad + n_pad);
    if (wave_live) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int nt = NARROW ? wn : 2 * wn + a, kt = 4 * wk + b;
                if ((!NARROW || a == 0) && nt < n_tiles && kt < k_tiles) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int row = 32 * nt + (v & 3) + 8 * (v >> 2) + 4 * h;
                        P[(size_t)row * k_pad + 32 * kt + i] = acc[a][b][v];
                    }
                }
            }
    }
    if (tid < n_pad) P[(size_t)n_pad * k_pad + tid] = bsum;
}

// partial reduction: block = 64 outputs x 4 slice groups (fixed order inside a group, groups combined in fixed order ->
// deterministic), so ~1000 workgroups keep enough loads in flight to run at HBM speed
__device__ __forceinline__ void dw_reduce_body(const nero_dw_job& job, const float* __restrict__ partials, int n_slices, int n_pad, int k_pad) {
    __shared__ float red[4][64];
    const size_t per = (size_t)n_pad * k_pad + n_pad;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    const int total = job.n_out * job.k_cols;
    const bool is_w = idx < total, is_b = !is_w && idx < total + job.n_out && job.db != nullptr;
    float s = 0.f;
    if (is_w || is_b) {
        size_t off;
        if (is_w) { const int n = idx / job.k_cols, k = idx - n * job.k_cols; off = (size_t)n * k_pad + k; }
        else off = (size_t)n_pad * k_pad + (idx - total);
        const int per_grp = (n_slices + 3) / 4;
        const int s0 = grp * per_grp, s1 = min(n_slices, s0 + per_grp);
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        int sl = s0;
        for (; sl + 4 <= s1; sl += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc4[u] += partials[(size_t)(sl + u) * per + off];
        }
        for (; sl < s1; ++sl) acc4[0] += partials[(size_t)sl * per + off];
        s = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && (is_w || is_b)) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (is_w) {
            const int n = idx / job.k_cols, k = idx - n * job.k_cols;
            float* o = job.dW + (size_t)n * job.ldw + job.col0 + k;
            *o = job.accumulate ? *o + t * job.scale : t * job.scale;
        } else {
            const int n = idx - total;
            job.db[n] = job.accumulate ? job.db[n] + t : t;
        }
    }
}
__global__ __launch_bounds__(256) void dw_reduce_kernel(nero_dw_job job, const float* __restrict__ partials, int n_slices, int n_pad, int k_pad) {
    dw_reduce_body(job, partials, n_slices, n_pad, k_pad);
}
// blockIdx.y = job of a batch (nero_dw_gemm_batch); the grid is sized for the largest job, the others' surplus blocks fall through
__global__ __launch_bounds__(256) void dw_reduce_batch_kernel(nero_dw_batch B, const float* __restrict__ partials, int n_slices) {
    const int q = blockIdx.y;
    if ((int)blockIdx.x * 64 >= B.j[q].n_out * B.j[q].k_cols + B.j[q].n_out) return;
    dw_reduce_body(B.j[q], partials + B.poff[q], n_slices, B.n_pad[q], B.k_pad[q]);
}

// head weight gradient: dWh[j][k] = sum_r dy[r][j] a[r][k] (+ extra[r][k] for j == 0).  Thread t owns columns 4(t&63)..+3 and the
// rows r = r_begin + (t>>6) + 4i of the block's slice (16-byte loads, 4 independent row strands), strands combined through LDS.
// one row of a strand: s[j] += dy[r][j] a[r][c4..c4+3] (+ extra for j == 0)
__device__ __forceinline__ void head_dw_row(float4 (&s)[4], float (&sb)[4], const float4 d, const float4 av, const float4 e, bool has_extra) {
    const float dj[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j].x = fmaf(dj[j], av.x, s[j].x); s[j].y = fmaf(dj[j], av.y, s[j].y);
        s[j].z = fmaf(dj[j], av.z, s[j].z); s[j].w = fmaf(dj[j], av.w, s[j].w);
        sb[j] += dj[j];
    }
    if (has_extra) { s[0].x += e.x; s[0].y += e.y; s[0].z += e.z; s[0].w += e.w; }
}
// EXTRA is a template parameter so that the row loop has no branch in it: with `if (extra)` inside, hipcc waited for every row's loads
// before it requested the next row (one kilobyte in flight per wavefront: 2.9 TB/s); now eight rows per strand are requested together
template <bool EXTRA>
__device__ __forceinline__ void head_dw_rows(float4 (&s)[4], float (&sb)[4], const float* __restrict__ dy, const float* __restrict__ a,
                                             const float* __restrict__ extra, int r, int r_end, int c4) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = EXTRA ? 4 : 8;                       // rows of a strand requested together (with three operands per row hipcc
    for (; r + 4 * (U - 1) < r_end; r += 4 * U) {          // serialises a batch of eight again)
        float4 d[U], av[U], e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = *reinterpret_cast<const float4*>(dy + (size_t)(r + 4 * u) * 4);
            av[u] = *reinterpret_cast<const float4*>(a + (size_t)(r + 4 * u) * NERO_HID + c4);
            e[u] = EXTRA ? *reinterpret_cast<const float4*>(extra + (size_t)(r + 4 * u) * NERO_HID + c4) : z;
        }
        __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise sinks the loads between the FMAs again)
#pragma unroll
        for (int u = 0; u < U; ++u) head_dw_row(s, sb, d[u], av[u], e[u], EXTRA);
    }
    for (; r < r_end; r += 4) {
        const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * 4);
        const float4 av = *reinterpret_cast<const float4*>(a + (size_t)r * NERO_HID + c4);
        const float4 e = EXTRA ? *reinterpret_cast<const float4*>(extra + (size_t)r * NERO_HID + c4) : z;
        head_dw_row(s, sb, d, av, e, EXTRA);
    }
}
__global__ __launch_bounds__(256) void head_dw_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                      const float* __restrict__ extra, int n_head, int n_rows,
                                                      int rows_per_slice, float* __restrict__ partials) {
    __shared__ float red[4][4][NERO_HID];
    __shared__ float redb[4][4];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c4 = 4 * lane;
    const int r_begin = blockIdx.x * rows_per_slice;
    int r_end = r_begin + rows_per_slice;
    r_end = r_end < n_rows ? r_end : n_rows;
    float4 s[4];
    float sb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (extra) head_dw_rows<true>(s, sb, dy, a, extra, r_begin + grp, r_end, c4);
    else head_dw_rows<false>(s, sb, dy, a, extra, r_begin + grp, r_end, c4);
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&red[grp][j][c4]) = s[j];
    if (lane == 0) for (int j = 0; j < 4; ++j) redb[grp][j] = sb[j];
    __syncthreads();
    float* P = partials + (size_t)blockIdx.x * (4 * NERO_HID + 4);
    for (int e = threadIdx.x; e < 4 * NERO_HID; e += 256) {
        const int j = e >> 8, k = e & 255;
        P[e] = (red[0][j][k] + red[1][j][k]) + (red[2][j][k] + red[3][j][k]);
    }
    if (threadIdx.x < 4) P[4 * NERO_HID + threadIdx.x] = (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
    (void)n_head;
}

// (ld_out / k_cols: dWh is written as n_head rows of k_cols columns with row pitch ld_out -- the chain drivers point it straight at the
//  destination tensor instead of copying a [4,256] temporary)
__global__ __launch_bounds__(256) void head_dw_reduce_kernel(const float* __restrict__ partials, int n_slices, int n_head, float* __restrict__ dWh,
                                                             float* __restrict__ dbh, int accumulate, int ld_out, int k_cols) {
    __shared__ float red[4][64];
    const size_t per = 4 * NERO_HID + 4;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;                   // [0, n_head*256) weights, then n_head biases
    const int nw = n_head * NERO_HID;
    const bool is_w = idx < nw, is_b = !is_w && idx < nw + n_head && dbh != nullptr;
    float s = 0.f;
    if (is_w || is_b) {
        const size_t off = is_w ? (size_t)idx : (size_t)(4 * NERO_HID + (idx - nw));
        const int per_grp = (n_slices + 3) / 4;
        const int s0 = grp * per_grp, s1 = min(n_slices, s0 + per_grp);
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        int sl = s0;
        for (; sl + 4 <= s1; sl += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc4[u] += partials[(size_t)(sl + u) * per + off];
        }
        for (; sl < s1; ++sl) acc4[0] += partials[(size_t)sl * per + off];
        s = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && (is_w || is_b)) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (is_w) {
            const int j = idx >> 8, k = idx & (NERO_HID - 1);
            if (k < k_cols) {
                float* o = dWh + (size_t)j * ld_out + k;
                *o = accumulate ? *o + t : t;
            }
        } else dbh[idx - nw] = accumulate ? dbh[idx - nw] + t : t;
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ W, int nrows, int ld, int col0, int ncols, int transpose,
                                   float scale, int kpad, int nt_count, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = (kpad >> 3) * nt_count * 64 * 4;
    if (idx >= total) return;
    const int t = idx & 3, lane = (idx >> 2) & 63;
    const int rest = idx >> 8;
    const int nt = rest % nt_count, c = rest / nt_count;
    const int k = 8 * c + 4 * (lane >> 5) + t, n = 32 * nt + (lane & 31);
    float v = 0.f;
    if (!transpose) { if (k < ncols && n < nrows) v = W[(size_t)n * ld + col0 + k]; }
    else            { if (k < nrows && n < ncols) v = W[(size_t)k * ld + col0 + n]; }
    out[idx] = nero_mul_rn(v, scale);
}

constexpr int DW_MAX_SLICES = 256;          // one row slice per CU
constexpr int DW_BATCH_SLICES = 1024 + NERO_DW_BATCH_MAX;     // upper bound of slices x jobs of one batched launch (>= DW_MAX_SLICES)
constexpr int DW_BATCH_ROWS = 131072;                          // batch only below this row count: above it a job alone fills the chip

inline int dw_rows_per_slice(int n_rows) {
    int rps = (n_rows + DW_MAX_SLICES - 1) / DW_MAX_SLICES;
    rps = (rps + DW_RC - 1) / DW_RC * DW_RC;
    return rps < DW_RC ? DW_RC : rps;
}

}  // namespace

extern "C" {

int nero_pack_weight(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale, int kpad,
                     int nt_count, float* out, void* stream) {
    if (!W || !out || kpad % 8 || nt_count <= 0) return nero_fail(NERO_ERR_ARG, "nero_pack_weight: bad argument");
    const int total = (kpad >> 3) * nt_count * 256;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, nrows, ld, col0,
                       ncols, transpose, scale, kpad, nt_count, out);
    return nero_check_launch("nero_pack_weight");
}

int nero_pack_weight_split(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale, int kpad,
                           int nt_count, void* out, void* stream) {
    if (!W || !out || nt_count <= 0) return nero_fail(NERO_ERR_ARG, "nero_pack_weight_split: bad argument");
    return nero_split_pack(W, nrows, ld, col0, ncols, transpose, scale, kpad, nt_count, out, (hipStream_t)stream);
}

int nero_pack_batch(const nero_pack_job* jobs, int n_jobs, void* stream) {
    if (!jobs && n_jobs > 0) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: bad argument");
    return nero_split_pack_batch(jobs, n_jobs, (hipStream_t)stream);
}

static int lds_bytes(int wide) { return (64 * LDA + 64 * (wide ? LDX_WIDE : LDX_NARROW)) * (int)sizeof(float); }

int nero_mlp_forward(const nero_fwd_chain* ch, int n_rows, void* stream) {
    if (!ch || n_rows < 0 || ch->n_layers > NERO_MAX_LAYERS) return nero_fail(NERO_ERR_ARG, "nero_mlp_forward: bad argument");
    if (n_rows == 0) return NERO_OK;
    if (ch->k_aux > (ch->aux_wide ? 88 : 40) || ch->k_init > 256 || (ch->k_init & 3) || (ch->k_aux & 3))
        return nero_fail(NERO_ERR_ARG, "nero_mlp_forward: init/aux width out of range");
    const dim3 grid((n_rows + 63) / 64), block(256);
    nero_prof_begin(NERO_K_FWD, 2.0 * ch->macs_per_row * n_rows, (hipStream_t)stream);
    nero_prof_note(n_rows, (unsigned)ch->n_layers | (ch->aux_wide ? 1u : 0u) << 8 | (unsigned)ch->k_init << 16);
    if (ch->gemm_mode == NERO_GEMM_BF16X6 || ch->gemm_mode == NERO_GEMM_F16X3) {
        const int rc = ch->gemm_mode == NERO_GEMM_F16X3 ? nero_f16_forward(ch, n_rows, (hipStream_t)stream)
                                                        : nero_split_forward(ch, n_rows, (hipStream_t)stream);
        nero_prof_end(NERO_K_FWD, (hipStream_t)stream);
        return rc != NERO_OK ? rc : nero_check_launch("nero_mlp_forward(split)");
    }
    if (ch->aux_wide) {
        NERO_ONCE(hipFuncSetAttribute((const void*)mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1)));
        hipLaunchKernelGGL(mlp_fwd_kernel<true>, grid, block, lds_bytes(1), (hipStream_t)stream, *ch, n_rows);
    } else {
        NERO_ONCE(hipFuncSetAttribute((const void*)mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(0)));
        hipLaunchKernelGGL(mlp_fwd_kernel<false>, grid, block, lds_bytes(0), (hipStream_t)stream, *ch, n_rows);
    }
    nero_prof_end(NERO_K_FWD, (hipStream_t)stream);
    return nero_check_launch("nero_mlp_forward");
}

int nero_mlp_tangent(const nero_tan_chain* ch, int n_rows, void* stream) {
    if (!ch || n_rows < 0 || ch->n_layers > NERO_MAX_LAYERS) return nero_fail(NERO_ERR_ARG, "nero_mlp_tangent: bad argument");
    if (n_rows == 0) return NERO_OK;
    const dim3 grid((n_rows + 63) / 64), block(256);
    nero_prof_begin(NERO_K_TAN, 2.0 * ch->macs_per_row * n_rows, (hipStream_t)stream);
    if (ch->gemm_mode == NERO_GEMM_BF16X6 || ch->gemm_mode == NERO_GEMM_F16X3) {
        const int rc = ch->gemm_mode == NERO_GEMM_F16X3 ? nero_f16_tangent(ch, n_rows, (hipStream_t)stream)
                                                        : nero_split_tangent(ch, n_rows, (hipStream_t)stream);
        nero_prof_end(NERO_K_TAN, (hipStream_t)stream);
        return rc != NERO_OK ? rc : nero_check_launch("nero_mlp_tangent(bf16x6)");
    }
    if (ch->aux_wide) {
        NERO_ONCE(hipFuncSetAttribute((const void*)mlp_tan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1)));
        hipLaunchKernelGGL(mlp_tan_kernel<true>, grid, block, lds_bytes(1), (hipStream_t)stream, *ch, n_rows);
    } else {
        NERO_ONCE(hipFuncSetAttribute((const void*)mlp_tan_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(0)));
        hipLaunchKernelGGL(mlp_tan_kernel<false>, grid, block, lds_bytes(0), (hipStream_t)stream, *ch, n_rows);
    }
    nero_prof_end(NERO_K_TAN, (hipStream_t)stream);
    return nero_check_launch("nero_mlp_tangent");
}

int nero_mlp_backward(const nero_bwd_chain* ch, int n_rows, void* stream) {
    if (!ch || n_rows < 0 || ch->n_layers > NERO_MAX_LAYERS) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward: bad argument");
    if (n_rows == 0) return NERO_OK;
    const dim3 grid((n_rows + 63) / 64), block(256);
    NERO_ONCE(hipFuncSetAttribute((const void*)mlp_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(0)));
    nero_prof_begin(NERO_K_BWD, 2.0 * ch->macs_per_row * n_rows, (hipStream_t)stream);
    if (nero_prof_is_on()) {                           // signature: layers | inj 8 | mask 9 | softplus 10 | dy 11 | d_aux 12 | heads 13 | saves 14 | d_init 15
        unsigned sig = (unsigned)ch->n_layers;
        for (int l = 0; l < ch->n_layers; ++l) {
            const nero_bwd_layer& Lr = ch->layer[l];
            sig |= (Lr.inj ? 1u : 0u) << 8 | (Lr.mask_prev ? 1u : 0u) << 9 | (Lr.act_prev == NERO_ACT_SOFTPLUS100 ? 1u : 0u) << 10;
            sig |= (Lr.n_head > 0 ? 1u : 0u) << 13 | (Lr.delta_prev ? 1u : 0u) << 14;
        }
        sig |= (ch->dy ? 1u : 0u) << 11 | (ch->d_aux ? 1u : 0u) << 12 | (ch->d_init ? 1u : 0u) << 15;
        nero_prof_note(n_rows, sig);
    }
    if (ch->gemm_mode == NERO_GEMM_BF16X6 || ch->gemm_mode == NERO_GEMM_F16X3) {
        const int rc = ch->gemm_mode == NERO_GEMM_F16X3 ? nero_f16_backward(ch, n_rows, (hipStream_t)stream)
                                                        : nero_split_backward(ch, n_rows, (hipStream_t)stream);
        nero_prof_end(NERO_K_BWD, (hipStream_t)stream);
        return rc != NERO_OK ? rc : nero_check_launch("nero_mlp_backward(bf16x6)");
    }
    hipLaunchKernelGGL(mlp_bwd_kernel<false>, grid, block, lds_bytes(0), (hipStream_t)stream, *ch, n_rows);
    nero_prof_end(NERO_K_BWD, (hipStream_t)stream);
    return nero_check_launch("nero_mlp_backward");
}

// the batching policy of nero_dw_gemm_batch (shared with the workspace query below)
struct DwBatchPolicy { int rows, big_total, big_group, small_total; };
static const DwBatchPolicy& dw_batch_policy() {
    static DwBatchPolicy P = {-1, 256, 8, 256};
    if (P.rows < 0) {
        const char* e = getenv("NERO_DW_BATCH_ROWS"); P.rows = e ? atoi(e) : DW_BATCH_ROWS;
        e = getenv("NERO_DW_BATCH_TOTAL"); if (e) P.big_total = atoi(e);
        e = getenv("NERO_DW_BATCH_GROUP"); if (e) P.big_group = atoi(e);
        e = getenv("NERO_DW_SMALL_TOTAL"); if (e) P.small_total = atoi(e);
        P.small_total = P.small_total < 64 ? 64 : (P.small_total > 1024 ? 1024 : P.small_total);
        P.big_group = P.big_group < 0 ? 0 : (P.big_group > NERO_DW_BATCH_MAX ? NERO_DW_BATCH_MAX : P.big_group);
        P.big_total = P.big_total < 16 ? 16 : (P.big_total > 1024 ? 1024 : P.big_total);
    }
    return P;
}

int nero_dw_workspace_floats(int n_rows) {
    // Partial matrices ([n_pad][k_pad] + n_pad floats, at most 256 x 256 + 256) of one launch: per-job launches write 256, small-row batches
    // up to 1024 + 12, large-row batches `big_total` + a few.  The bound must be MONOTONIC in n_rows -- callers size one buffer for the
    // largest row count of a step and hand it to launches over fewer rows, which may sit in the small-row regime -- so it is the
    // maximum over the regimes whatever n_rows is (a round-4 attempt to size it by regime let a 100 k-row launch overrun a buffer sized
    // for 300 k rows: found by tests/test_determinism.py).  One 4 x 256 (+4) partial per 128-row block for nero_head_dw.
    const int rows = n_rows < 1 ? 1 : n_rows;
    const DwBatchPolicy& P = dw_batch_policy();
    long mats = DW_BATCH_SLICES;
    if (P.big_total + NERO_DW_BATCH_MAX + 8 > mats) mats = P.big_total + NERO_DW_BATCH_MAX + 8;
    const long head_blocks = (rows + 127) / 128;
    const long a = mats * (256 * 256 + 256), b = head_blocks * (4 * NERO_HID + 4);
    return (int)(a > b ? a : b);
}

int nero_dw_gemm(const nero_dw_job* job, int n_rows, float* partials, void* stream) {
    if (!job || !job->d0 || !job->b0 || !job->dW || !partials || job->n_out > 256 || job->k_cols > 256 || job->n_out <= 0 || job->k_cols <= 0)
        return nero_fail(NERO_ERR_ARG, "nero_dw_gemm: bad argument");
    const int n_pad = (job->n_out + 31) / 32 * 32, k_pad = (job->k_cols + 31) / 32 * 32;
    const int rows = n_rows < 1 ? 1 : n_rows;
    const int rps = dw_rows_per_slice(rows);
    const int slices = (rows + rps - 1) / rps;
    const int lds = 4 * DW_BUF * (int)sizeof(float);
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    nero_prof_begin(NERO_K_DW, 2.0 * job->n_out * job->k_cols * (job->d1 ? 2.0 : 1.0) * n_rows, (hipStream_t)stream);
    if (job->gemm_mode == NERO_GEMM_BF16X6)
        nero_split_dw(job, n_rows, rps, slices, partials, n_pad, k_pad, (hipStream_t)stream);
    else if (job->gemm_mode == NERO_GEMM_F16X3)
        nero_f16_dw(job, n_rows, rps, slices, partials, n_pad, k_pad, (hipStream_t)stream);
    else if (k_pad <= 128)
        hipLaunchKernelGGL(dw_gemm_kernel<true>, dim3(slices), dim3(512), lds, (hipStream_t)stream, *job, n_rows, rps, partials, n_pad, k_pad);
    else
        hipLaunchKernelGGL(dw_gemm_kernel<false>, dim3(slices), dim3(512), lds, (hipStream_t)stream, *job, n_rows, rps, partials, n_pad, k_pad);
    nero_prof_end(NERO_K_DW, (hipStream_t)stream);
    const int total = job->n_out * job->k_cols + job->n_out;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, (hipStream_t)stream, *job, partials, slices, n_pad, k_pad);
    return nero_check_launch("nero_dw_gemm");
}

int nero_dw_gemm_batch(const nero_dw_job* jobs, int n_jobs, int n_rows, float* partials, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs) || !partials) return nero_fail(NERO_ERR_ARG, "nero_dw_gemm_batch: bad argument");
    const int rows = n_rows < 1 ? 1 : n_rows;
    bool f16 = true;
    for (int i = 0; i < n_jobs; ++i) {
        const nero_dw_job& J = jobs[i];
        if (!J.d0 || !J.b0 || !J.dW || J.n_out > 256 || J.k_cols > 256 || J.n_out <= 0 || J.k_cols <= 0)
            return nero_fail(NERO_ERR_ARG, "nero_dw_gemm_batch: bad job");
        f16 = f16 && J.gemm_mode == NERO_GEMM_F16X3;
    }
    // Two batching regimes (both: blockIdx.y = job, one reduction launch per group).  Below DW_BATCH_ROWS rows a job alone neither fills
    // the chip nor amortises its partial matrices: groups of up to NERO_DW_BATCH_MAX jobs, and -- round 4 -- 256 slices over the whole
    // group here too (it was ~1024: at the reference's 512-ray batch the partial matrices and their reduction cost more than the rows they
    // cover; 6.30 -> 5.95 ms per step at 512 rays, 9.75 -> 8.90 at 1024, Stage II 6.90 -> 6.75; 128 and 64 slices are slower again:
    // profiles/r04_dw_batch_sweep.txt).  At or above it (round 4) the
    // jobs of a chain still run in groups, of 8, with ONE slice per CU over the whole group (256 slices in all, 32 per job): every
    // workgroup streams ~9 k rows of one job, and the group writes 8 x 32 partial matrices instead of 8 x 256 -- a job alone spent a
    // fifth of its HBM traffic (67 MB written + read back for 610 MB of operands) and ~11 us of dirty-line write-back at the kernel
    // boundary on them: 31.28 -> 30.75 ms per step at 4096 rays (same box, profiles/r04_dw_batch_sweep.txt).
    // Experiment switches: NERO_DW_BATCH_ROWS (the regime boundary), NERO_DW_BATCH_TOTAL / NERO_DW_BATCH_GROUP (slices per group / jobs
    // per group of the large-row regime; 0 rows = the old per-job launches).
    const DwBatchPolicy& POL = dw_batch_policy();
    const int batch_rows = POL.rows, big_total = POL.big_total, big_group = POL.big_group;
    const bool big = rows >= batch_rows;
    const int batch_total = big ? big_total : POL.small_total, batch_group = big ? big_group : NERO_DW_BATCH_MAX;
    if (!f16 || n_jobs < 2 || batch_group < 1) {                 // the per-job path: one launch (+ reduction) per job
        for (int i = 0; i < n_jobs; ++i) {
            const int rc = nero_dw_gemm(jobs + i, n_rows, partials, stream);
            if (rc != NERO_OK) return rc;
        }
        return NERO_OK;
    }
    if (n_jobs > 256) return nero_fail(NERO_ERR_ARG, "nero_dw_gemm_batch: at most 256 jobs per call");
    // narrow (k_pad <= 128) and wide jobs are different kernels; inside a kind, groups of NERO_DW_BATCH_MAX in the caller's order.
    // The partial buffer is reused from group to group (stream order).
    for (int narrow = 0; narrow < 2; ++narrow) {
        int idx[256], n = 0;
        for (int i = 0; i < n_jobs; ++i)
            if ((((jobs[i].k_cols + 31) / 32 * 32) <= 128) == (narrow != 0)) idx[n++] = i;
        for (int g0 = 0; g0 < n; g0 += batch_group) {
            const int ng = n - g0 < batch_group ? n - g0 : batch_group;
            // ~1024 slices over the group's jobs, at least 128 rows (8 chunks) per slice, at most one slice per CU and job
            int target = batch_total / ng;
            target = target < 4 ? 4 : (target > DW_MAX_SLICES ? DW_MAX_SLICES : target);
            int rps = (rows + target - 1) / target;
            rps = (rps + 15) / 16 * 16;
            rps = rps < 128 ? 128 : rps;
            const int slices = (rows + rps - 1) / rps;
            nero_dw_batch B;
            memset(&B, 0, sizeof(B));
            size_t off = 0;
            double flops = 0.0;
            int max_total = 0;
            for (int k = 0; k < ng; ++k) {
                const nero_dw_job& J = jobs[idx[g0 + k]];
                B.j[k] = J;
                B.n_pad[k] = (short)((J.n_out + 31) / 32 * 32);
                B.k_pad[k] = (short)((J.k_cols + 31) / 32 * 32);
                B.poff[k] = off;
                off += (size_t)slices * ((size_t)B.n_pad[k] * B.k_pad[k] + B.n_pad[k]);
                flops += 2.0 * J.n_out * J.k_cols * (J.d1 ? 2.0 : 1.0) * n_rows;
                const int total = J.n_out * J.k_cols + J.n_out;
                max_total = total > max_total ? total : max_total;
            }
            if (off > (size_t)nero_dw_workspace_floats(n_rows)) return nero_fail(NERO_ERR_ARG, "nero_dw_gemm_batch: partial buffer too small");
            nero_prof_begin(NERO_K_DW, flops, (hipStream_t)stream);
            nero_f16_dw_batch(&B, ng, narrow, n_rows, rps, slices, partials, (hipStream_t)stream);
            nero_prof_end(NERO_K_DW, (hipStream_t)stream);
            hipLaunchKernelGGL(dw_reduce_batch_kernel, dim3((max_total + 63) / 64, ng), dim3(256), 0, (hipStream_t)stream, B, partials, slices);
        }
    }
    return nero_check_launch("nero_dw_gemm_batch");
}

int nero_head_dw(const float* dy, const float* a, const float* extra, int n_head, int n_rows, float* dWh, float* dbh,
                 float* partials, int accumulate, void* stream) {
    return nero_head_dw_ld(dy, a, extra, n_head, n_rows, dWh, NERO_HID, NERO_HID, dbh, partials, accumulate, stream);
}

int nero_head_dw_ld(const float* dy, const float* a, const float* extra, int n_head, int n_rows, float* dWh, int ld_dwh, int k_cols, float* dbh,
                    float* partials, int accumulate, void* stream) {
    if (!dy || !a || !dWh || !partials || n_head < 1 || n_head > 4 || k_cols < 1 || k_cols > NERO_HID || ld_dwh < k_cols)
        return nero_fail(NERO_ERR_ARG, "nero_head_dw: bad argument");
    const int rows = n_rows < 1 ? 1 : n_rows;
    // rows per block: 512 at the step's 300k-row launches (measured 92 us against 106 at 256, 114 at 768, 141 at 1024), fewer
    // rows for smaller inputs so that ~2 blocks per CU remain
    int rps = rows / 512 / 128 * 128;
    rps = rps < 128 ? 128 : (rps > 512 ? 512 : rps);
    {   // a whole number of blocks per CU where the rows allow it: k x 256 slices of equal length, ~512 rows each (297 k rows: 580 blocks
        // of 512 rows = 2.3 per CU took 73.6 us, 768 of 388 rows 61.8)
        int k = (rows + 256 * 512 / 2) / (256 * 512);
        k = k < 1 ? 1 : k;
        int q = (rows + 256 * k - 1) / (256 * k);
        q = (q + 3) / 4 * 4;
        if (q >= 128) rps = q;
    }
    const int slices = (rows + rps - 1) / rps;
    hipLaunchKernelGGL(head_dw_kernel, dim3(slices), dim3(256), 0, (hipStream_t)stream, dy, a, extra, n_head, n_rows, rps, partials);
    const int total = n_head * NERO_HID + n_head;
    hipLaunchKernelGGL(head_dw_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, (hipStream_t)stream, partials, slices, n_head, dWh, dbh, accumulate,
                       ld_dwh, k_cols);
    return nero_check_launch("nero_head_dw");
}

}  // extern "C"
