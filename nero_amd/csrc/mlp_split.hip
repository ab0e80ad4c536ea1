// mlp_split.hip -- the fused MLP-chain passes of mlp_engine.hip on the bf16 matrix pipe with fp32-grade arithmetic
// (gemm_mode NERO_GEMM_BF16X6).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA (2.5 PFLOP/s vs 157 TFLOP/s), so every fp32
// operand x is carried as THREE bf16 planes  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1),
// round-to-nearest: |x1| <= 2^-9 |x|, |x2| <= 2^-18 |x|, and the sum is exact because 3 x (8+1) bits >= 24), and a product
// sum_k w x is evaluated with the six plane products of weight 2^0, 2^-9, 2^-9, 2^-18, 2^-18, 2^-18:
//      w0 x0 + (w0 x1 + w1 x0) + (w0 x2 + w1 x1 + w2 x0),                dropped: w1 x2 + w2 x1 + w2 x2 <= 3 * 2^-27 |w x|
// accumulated in fp32 inside the MFMA.  The result differs from an fp32 fmaf chain by less than fp32's own rounding
// (2^-24 per term) -- the same error class as a different summation order -- at 1/6 of the bf16 rate = 417 TFLOP/s
// fp32-equivalent peak, 2.65x the f32 MFMA.
//
// Kernel shape: one 512-thread workgroup owns 64 rows and walks the whole layer list; the activations live
// in LDS as three bf16 planes [64][256+8] (row stride 528 B -> conflict-free 16-byte fragment reads); the product is
// computed TRANSPOSED, Y^T[f][row] = W[f][k] X^T[k][row]: the weight planes are the A operand (streamed from the
// L2-resident packed image, one 16-byte load per lane per plane per 16 k), the activation planes the B operand, so that
// the accumulator layout (col = row of the batch, 4 consecutive features per register quad) lets the epilogue run
// straight out of the registers: bias + activation + 3-way split + 8-byte LDS plane writes + 16-byte global saves,
// without the LDS round trip of the f32 engine.  Wave w owns feature tile w (32 features) for both 32-row halves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float BETA = 100.0f;
constexpr int SA = 528;                 // bytes per row of a main plane: 256 bf16 + 16 B pad
constexpr int PLANE_A = 64 * SA;        // 33792 B
constexpr int SX_N = 112, SX_W = 208;   // aux plane row strides: 48 / 96 columns + 16 B pad (16 x odd -> conflict-free)

// ---- fp32 <-> bf16 planes -------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (RN): a -> low half, b -> high half
    f32x2 v = {a, b};
    bf16x2 c = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, c);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    float ra = a - bf_lo(p0), rb = b - bf_hi(p0);
    p1 = pk_bf16(ra, rb);
    ra -= bf_lo(p1);
    rb -= bf_hi(p1);
    p2 = pk_bf16(ra, rb);
}
// four consecutive columns -> one 8-byte store per plane
__device__ __forceinline__ void store_planes4(char* dst, int plane_bytes, float4 v) {
    unsigned a0, a1, a2, b0, b1, b2;
    split2(v.x, v.y, a0, a1, a2);
    split2(v.z, v.w, b0, b1, b2);
    *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(dst + plane_bytes) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(dst + 2 * plane_bytes) = make_uint2(a2, b2);
}
__device__ __forceinline__ float4 load_planes4(const char* src, int plane_bytes) {
    const uint2 a = *reinterpret_cast<const uint2*>(src);
    const uint2 b = *reinterpret_cast<const uint2*>(src + plane_bytes);
    const uint2 c = *reinterpret_cast<const uint2*>(src + 2 * plane_bytes);
    float4 v;
    v.x = (bf_lo(a.x) + bf_lo(b.x)) + bf_lo(c.x);
    v.y = (bf_hi(a.x) + bf_hi(b.x)) + bf_hi(c.x);
    v.z = (bf_lo(a.y) + bf_lo(b.y)) + bf_lo(c.y);
    v.w = (bf_hi(a.y) + bf_hi(b.y)) + bf_hi(c.y);
    return v;
}

// ---- accumulator-layout <-> row-major global I/O through a wave-private LDS scratch -----------------------------------------
// In the accumulator layout a wave instruction touches 32 rows x 32 bytes of a [rows][256] matrix, which is issue bound in the
// texture path (~4k cycles per 64x256 stream).  Routing the 32x32 block through a private [32][36]-float scratch turns every
// global access into 8 rows x 128 contiguous bytes.  Row-major mapping: pass p (0..3) -> row 8p + (lane>>3), columns 4(lane&7)..+3.
constexpr int SCR_LD = 36;                           // floats; 144-byte rows -> conflict-free 16-byte column accesses
constexpr int SCR_BYTES = 32 * SCR_LD * 4;           // 4608 B per wave
__device__ __forceinline__ void rm_prefetch(float4 (&raw)[4], const float* __restrict__ gblock, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) raw[p] = *reinterpret_cast<const float4*>(gblock + (size_t)(8 * p + (lane >> 3)) * NERO_HID + 4 * (lane & 7));
}
// row-major registers -> accumulator-layout quads q[g] = features 8g + 4h + 0..3 of row i
__device__ __forceinline__ void rm_to_acc(float* scr, const float4 (&raw)[4], float4 (&q)[4], int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(scr + (8 * p + (lane >> 3)) * SCR_LD + 4 * (lane & 7)) = raw[p];
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = *reinterpret_cast<const float4*>(scr + i * SCR_LD + 8 * g + 4 * h);
    __builtin_amdgcn_wave_barrier();
}
// accumulator-layout quads -> row-major global store
__device__ __forceinline__ void acc_to_global(float* scr, const float4 (&q)[4], float* __restrict__ gblock, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(scr + i * SCR_LD + 8 * g + 4 * h) = q[g];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 4; ++p)
        *reinterpret_cast<float4*>(gblock + (size_t)(8 * p + (lane >> 3)) * NERO_HID + 4 * (lane & 7)) =
            *reinterpret_cast<const float4*>(scr + (8 * p + (lane >> 3)) * SCR_LD + 4 * (lane & 7));
    __builtin_amdgcn_wave_barrier();
}

// softplus(beta=100, threshold=20) and its derivative recovered from the output: identical to mlp_engine.hip
__device__ __forceinline__ float log1p_small(float u) {
    return u * (1.f + u * (-0.5f + u * (0.33333334f + u * (-0.25f + u * 0.2f))));
}
// (written with selects only: the epilogues run them on 16 values per lane and must stay branch-free)
__device__ __forceinline__ float softplus100(float x) {
    const float bx = BETA * x;
    const float u = __expf(-fabsf(bx));
    const float ls = log1p_small(u), lg = __logf(1.f + u);
    const float l = u < 0.0625f ? ls : lg;
    const float r = fmaxf(x, 0.f) + l * (1.0f / BETA);
    return bx > 20.f ? x : r;
}
__device__ __forceinline__ float softplus100_grad_from_out(float a) {
    const float ba = BETA * a;
    const float ps = ba * (1.f + ba * (-0.5f + ba * (0.16666667f + ba * (-0.041666668f))));
    const float pe = 1.f - __expf(-ba);
    const float r = ba < 0.03125f ? ps : pe;
    return ba > 20.f ? 1.f : r;
}
template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
    if (ACT == NERO_ACT_RELU) return fmaxf(x, 0.f);
    if (ACT == NERO_ACT_SOFTPLUS100) return softplus100(x);
    return x;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float a, float g) {
    if (ACT == NERO_ACT_RELU) return a > 0.f ? g : 0.f;
    if (ACT == NERO_ACT_SOFTPLUS100) return g * softplus100_grad_from_out(a);
    return g;
}

// ---- GEMM core ------------------------------------------------------------------------------------------------------
// acc[r] (r = 32-row half) += sum over `n` k-steps of 16: the six significant plane products.
// Software pipeline: the weight planes (A operand, streamed from L2: 200-800 cycles under load) run THREE steps ahead in a
// ring of four register sets; the activation planes (B operand, LDS: ~100 cycles) one step ahead in a double buffer.
struct WF { uint4 w0, w1, w2; };                 // lane (i, h): 8 consecutive k of feature row i, per plane
struct XF { uint4 x00, x01, x02, x10, x11, x12; };   // lane (i, h): 8 consecutive k of batch row i (rows 0..31 / 32..63), per plane

__device__ __forceinline__ void load_w(WF& o, const uint4* wp, int c) {
    const uint4* w = wp + (size_t)c * 192;
    o.w0 = w[0];
    o.w1 = w[64];
    o.w2 = w[128];
}
__device__ __forceinline__ void load_x(XF& o, const char* xp, int half_bytes, int plane_bytes, int c) {
    const char* x = xp + c * 32;
    o.x00 = *reinterpret_cast<const uint4*>(x);
    o.x01 = *reinterpret_cast<const uint4*>(x + plane_bytes);
    o.x02 = *reinterpret_cast<const uint4*>(x + 2 * plane_bytes);
    x += half_bytes;
    o.x10 = *reinterpret_cast<const uint4*>(x);
    o.x11 = *reinterpret_cast<const uint4*>(x + plane_bytes);
    o.x12 = *reinterpret_cast<const uint4*>(x + 2 * plane_bytes);
}

#define NERO_MF(ACC, A, B) \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

__device__ __forceinline__ void ops_compute(f32x16 (&acc)[2], const WF& w, const XF& x) {
    NERO_MF(acc[0], w.w2, x.x00); NERO_MF(acc[1], w.w2, x.x10);
    NERO_MF(acc[0], w.w1, x.x01); NERO_MF(acc[1], w.w1, x.x11);
    NERO_MF(acc[0], w.w0, x.x02); NERO_MF(acc[1], w.w0, x.x12);
    NERO_MF(acc[0], w.w1, x.x00); NERO_MF(acc[1], w.w1, x.x10);
    NERO_MF(acc[0], w.w0, x.x01); NERO_MF(acc[1], w.w0, x.x11);
    NERO_MF(acc[0], w.w0, x.x00); NERO_MF(acc[1], w.w0, x.x10);
}

#define NERO_FENCE() __builtin_amdgcn_sched_barrier(0)     /* keeps hipcc from sinking the prefetches down to their first use */

__device__ __forceinline__ void gemm_split(f32x16 (&acc)[2], const uint4* wp, const char* xp, int half_bytes, int plane_bytes, int n) {
    if (n <= 0) return;
    WF wa, wb, wc, wd;
    XF xa, xb;
    const int last = n - 1;
#define NERO_CL(c) ((c) < last ? (c) : last)
    load_w(wa, wp, 0);
    load_w(wb, wp, NERO_CL(1));
    load_w(wc, wp, NERO_CL(2));
    load_x(xa, xp, half_bytes, plane_bytes, 0);
    NERO_FENCE();
    for (int c = 0; c < n; c += 4) {
        load_w(wd, wp, NERO_CL(c + 3)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 1)); NERO_FENCE();
        ops_compute(acc, wa, xa); NERO_FENCE();
        if (c + 1 < n) {
            load_w(wa, wp, NERO_CL(c + 4)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 2)); NERO_FENCE();
            ops_compute(acc, wb, xb); NERO_FENCE();
        }
        if (c + 2 < n) {
            load_w(wb, wp, NERO_CL(c + 5)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 3)); NERO_FENCE();
            ops_compute(acc, wc, xa); NERO_FENCE();
        }
        if (c + 3 < n) {
            load_w(wc, wp, NERO_CL(c + 6)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 4)); NERO_FENCE();
            ops_compute(acc, wd, xb); NERO_FENCE();
        }
    }
#undef NERO_CL
}

__device__ __forceinline__ void zero2(f32x16 (&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
}

// rows [row0, row0+64) x first k columns (k multiple of 4) of a row-major fp32 matrix -> three planes; columns [k, kpad16)
// are zero filled, rows are clamped to n_rows-1
__device__ __forceinline__ void load_planes(char* planes, int stride, int plane_bytes, const float* __restrict__ src, int ld, int k,
                                            int row0, int n_rows, int tid) {
    const int k16 = (k + 15) & ~15, q4 = k16 >> 2;
    for (int idx = tid; idx < 64 * q4; idx += 512) {
        const int r = idx / q4, c4 = (idx - r * q4) * 4;
        int gr = row0 + r;
        gr = gr < n_rows ? gr : n_rows - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < k) v = *reinterpret_cast<const float4*>(src + (size_t)gr * ld + c4);
        store_planes4(planes + r * stride + c4 * 2, plane_bytes, v);
    }
}

// VALU head on the current activation planes: out[r][j] = b[j] + sum_k x[r][k] W[j][k], k < hk (8 threads per row)
__device__ __forceinline__ void eval_head_split(const char* planes, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ out, int n_head, int hk, int row0, int tid) {
    const int r = tid >> 3, q = tid & 7;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = 4 * q; c4 < hk; c4 += 32) {
        const float4 x = load_planes4(planes + r * SA + c4 * 2, PLANE_A);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_head) {
                const float4 ww = *reinterpret_cast<const float4*>(w + j * NERO_HID + c4);
                s[j] = fmaf(x.x, ww.x, fmaf(x.y, ww.y, fmaf(x.z, ww.z, fmaf(x.w, ww.w, s[j]))));
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] += __shfl_xor(s[j], 1);
        s[j] += __shfl_xor(s[j], 2);
        s[j] += __shfl_xor(s[j], 4);
    }
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_head) out[(size_t)(row0 + r) * 4 + j] = s[j] + (b ? b[j] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward chain
// ---------------------------------------------------------------------------------------------------------------------
// epilogue straight out of the accumulators: lane (i, h) of wave w holds, for batch rows i and 32+i, the features
// 32w + 8g + 4h + {0..3}, g = 0..3 -> bias, activation, optional 16-byte global save, 3-way split, 8-byte plane stores
// sv: save pointer in accumulator layout (&save[row0+i][32w+4h]); sblock/scr: row-major block pointer (&save[row0][32w]) and the
// wave's scratch, used instead of sv when scr != nullptr
template <int ACT>
__device__ __forceinline__ void fwd_epilogue(const f32x16 (&acc)[2], const float4 (&bq)[4], char* dst, float* sv, float* sblock,
                                             float* scr, int lane) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float4 q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = act_fwd<ACT>(acc[r][4 * g] + bq[g].x);
            v.y = act_fwd<ACT>(acc[r][4 * g + 1] + bq[g].y);
            v.z = act_fwd<ACT>(acc[r][4 * g + 2] + bq[g].z);
            v.w = act_fwd<ACT>(acc[r][4 * g + 3] + bq[g].w);
            if (sv && !scr) *reinterpret_cast<float4*>(sv + (size_t)r * 32 * NERO_HID + 8 * g) = v;
            store_planes4(dst + r * 32 * SA + 16 * g, PLANE_A, v);
            q[g] = v;
        }
        if (sv && scr) acc_to_global(scr, q, sblock + (size_t)r * 32 * NERO_HID, lane);
    }
}

template <bool WIDE>
__global__ __launch_bounds__(512, 1) void fwd_split_kernel(nero_fwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SX = WIDE ? SX_W : SX_N;
    constexpr int PLANE_X = 64 * SX;
    char* actp = smem;
    char* auxp = smem + 3 * PLANE_A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 64;
    if (ch.init) load_planes(actp, SA, PLANE_A, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
    if (ch.aux) load_planes(auxp, SX, PLANE_X, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_fwd_layer& L = ch.layer[l];
        if (L.n_head > 0) eval_head_split(actp, L.head_w, L.head_b, L.head_out, L.n_head, L.head_k, row0, tid);
        if (L.n_tiles == 0) continue;
        const bool live_wave = wave < L.n_tiles;
        f32x16 acc[2];
        zero2(acc);
        float4 bq[4];                                      // this lane's 16 bias values (features 32w + 8g + 4h + 0..3), fetched under the GEMM
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bq[g] = (live_wave && L.bias) ? *reinterpret_cast<const float4*>(L.bias + 32 * wave + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (live_wave) {
            const int sm = L.k_main >> 4, sx = L.k_aux >> 4;
            gemm_split(acc, reinterpret_cast<const uint4*>(L.w_main) + (size_t)wave * sm * 192 + lane, actp + i * SA + 16 * h,
                       32 * SA, PLANE_A, sm);
            gemm_split(acc, reinterpret_cast<const uint4*>(L.w_aux) + (size_t)wave * sx * 192 + lane, auxp + i * SX + 16 * h,
                       32 * SX, PLANE_X, sx);
        }
        __syncthreads();                                   // every wave is done reading the input planes
        if (live_wave) {
            char* dst = actp + i * SA + (32 * wave + 4 * h) * 2;
            float* sv = L.save ? L.save + (size_t)(row0 + i) * NERO_HID + 32 * wave + 4 * h : nullptr;
            float* sblock = L.save ? L.save + (size_t)row0 * NERO_HID + 32 * wave : nullptr;
            // (the wide-aux variant has no LDS left for the scratch: it keeps the direct accumulator-layout stores)
            float* scr = WIDE ? nullptr : reinterpret_cast<float*>(smem + 3 * PLANE_A + 3 * PLANE_X + wave * SCR_BYTES);
            if (L.act == NERO_ACT_RELU) fwd_epilogue<NERO_ACT_RELU>(acc, bq, dst, sv, sblock, scr, lane);
            else if (L.act == NERO_ACT_SOFTPLUS100) fwd_epilogue<NERO_ACT_SOFTPLUS100>(acc, bq, dst, sv, sblock, scr, lane);
            else fwd_epilogue<NERO_ACT_NONE>(acc, bq, dst, sv, sblock, scr, lane);
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
    const float r2 = (BETA * a > 20.f) ? 0.f : BETA * (1.f - s);     // sigma''/sigma' (0 in torch's linear region)
    ij = live ? gb * r2 * zd : 0.f;
}

template <bool WIDE>
__global__ __launch_bounds__(512, 1) void tan_split_kernel(nero_tan_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SX = WIDE ? SX_W : SX_N;
    constexpr int PLANE_X = 64 * SX;
    char* actp = smem;
    char* auxp = smem + 3 * PLANE_A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 64;
    if (ch.init) load_planes(actp, SA, PLANE_A, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
    if (ch.aux) load_planes(auxp, SX, PLANE_X, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_tan_layer& L = ch.layer[l];
        const bool live_wave = wave < L.n_tiles;
        const size_t goff = (size_t)(row0 + i) * NERO_HID + 32 * wave + 4 * h;     // + r*32*HID + 8g
        // saved activations of this lane's 32 outputs, requested before the GEMM
        float4 pa[2][4], pg[2][4];
        if (live_wave) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pa[r][g] = *reinterpret_cast<const float4*>(L.a_saved + goff + (size_t)r * 32 * NERO_HID + 8 * g);
                    pg[r][g] = *reinterpret_cast<const float4*>(L.gbar + goff + (size_t)r * 32 * NERO_HID + 8 * g);
                }
        }
        f32x16 acc[2];
        zero2(acc);
        if (live_wave) {
            const int sm = L.k_main >> 4, sx = L.k_aux >> 4;
            gemm_split(acc, reinterpret_cast<const uint4*>(L.w_main) + (size_t)wave * sm * 192 + lane, actp + i * SA + 16 * h,
                       32 * SA, PLANE_A, sm);
            gemm_split(acc, reinterpret_cast<const uint4*>(L.w_aux) + (size_t)wave * sx * 192 + lane, auxp + i * SX + 16 * h,
                       32 * SX, PLANE_X, sx);
        }
        __syncthreads();
        if (live_wave) {
            char* dst = actp + i * SA + (32 * wave + 4 * h) * 2;
            float* scr = reinterpret_cast<float*>(smem + 3 * PLANE_A + 3 * PLANE_X + wave * SCR_BYTES);
            const size_t boff = (size_t)row0 * NERO_HID + 32 * wave;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool live = (row0 + 32 * r + i) < n_rows;
                float4 adq[4], ijq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = pa[r][g], gb = pg[r][g];
                    float4 ad, ij;
                    tan_elem(a.x, acc[r][4 * g], gb.x, live, ad.x, ij.x);
                    tan_elem(a.y, acc[r][4 * g + 1], gb.y, live, ad.y, ij.y);
                    tan_elem(a.z, acc[r][4 * g + 2], gb.z, live, ad.z, ij.z);
                    tan_elem(a.w, acc[r][4 * g + 3], gb.w, live, ad.w, ij.w);
                    store_planes4(dst + r * 32 * SA + 16 * g, PLANE_A, ad);
                    adq[g] = live ? ad : make_float4(0.f, 0.f, 0.f, 0.f);
                    ijq[g] = ij;
                }
                // row-major (128-byte segment) stores through the wave's LDS scratch
                acc_to_global(scr, adq, L.adot + boff + (size_t)r * 32 * NERO_HID, lane);
                acc_to_global(scr, ijq, L.inj + boff + (size_t)r * 32 * NERO_HID, lane);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse chain:  delta_{l-1} = (delta_l W_l [+ dy_head W_head]) * act'(a_{l-1}) [+ inj_{l-1}]
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT, bool HEAD>
__device__ __forceinline__ void bwd_epilogue(const f32x16 (&acc)[2], const float4 (&pa)[2][4], const float4 (&pi)[2][4], bool has_inj,
                                             const nero_bwd_layer& L, char* dst, size_t boff, float* scr, int lane, int row0, int i,
                                             int fbase, int n_rows) {
    const int nh = L.n_head;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int grow = row0 + 32 * r + i;
        const bool live = grow < n_rows;
        float dj[4] = {0.f, 0.f, 0.f, 0.f};
        float4 dq[4];
        if (HEAD) {
            const float4 dyh = *reinterpret_cast<const float4*>(L.head_dy + (size_t)grow * 4);
            dj[0] = dyh.x; dj[1] = dyh.y; dj[2] = dyh.z; dj[3] = dyh.w;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 gs = make_float4(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3]);
            if (HEAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nh) {
                        const float4 hw = *reinterpret_cast<const float4*>(L.head_w + j * NERO_HID + fbase + 8 * g);
                        gs.x = fmaf(dj[j], hw.x, gs.x); gs.y = fmaf(dj[j], hw.y, gs.y);
                        gs.z = fmaf(dj[j], hw.z, gs.z); gs.w = fmaf(dj[j], hw.w, gs.w);
                    }
                }
            }
            const float4 a = pa[r][g];
            float4 d;
            d.x = act_grad<ACT>(a.x, gs.x); d.y = act_grad<ACT>(a.y, gs.y);
            d.z = act_grad<ACT>(a.z, gs.z); d.w = act_grad<ACT>(a.w, gs.w);
            if (has_inj) { d.x += pi[r][g].x; d.y += pi[r][g].y; d.z += pi[r][g].z; d.w += pi[r][g].w; }
            if (!live) d = make_float4(0.f, 0.f, 0.f, 0.f);
            store_planes4(dst + r * 32 * SA + 16 * g, PLANE_A, d);
            dq[g] = d;
        }
        // row-major (128-byte segment) store through the wave's LDS scratch
        if (L.delta_prev) acc_to_global(scr, dq, L.delta_prev + boff + (size_t)r * 32 * NERO_HID, lane);
    }
}

template <int ACT>
__device__ __forceinline__ void bwd_epilogue_h(const f32x16 (&acc)[2], const float4 (&pa)[2][4], const float4 (&pi)[2][4], bool has_inj,
                                               const nero_bwd_layer& L, char* dst, size_t boff, float* scr, int lane, int row0, int i,
                                               int fbase, int n_rows) {
    if (L.n_head > 0) bwd_epilogue<ACT, true>(acc, pa, pi, has_inj, L, dst, boff, scr, lane, row0, i, fbase, n_rows);
    else bwd_epilogue<ACT, false>(acc, pa, pi, has_inj, L, dst, boff, scr, lane, row0, i, fbase, n_rows);
}

__global__ __launch_bounds__(512, 1) void bwd_split_kernel(nero_bwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* actp = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 64;
    if (ch.dy) load_planes(actp, SA, PLANE_A, ch.dy, ch.ld_dy, ch.k_dy, row0, n_rows, tid);
    else {
        for (int idx = tid; idx < 3 * PLANE_A / 16; idx += 512) reinterpret_cast<uint4*>(actp)[idx] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    for (int l = ch.n_layers - 1; l >= 0; --l) {
        const nero_bwd_layer& L = ch.layer[l];
        const bool first = (L.a_prev == nullptr);
        if (first && ch.d_init == nullptr && !(ch.d_aux && L.w_aux_t)) break;
        const int nt = L.k_main_tiles;
        const bool live_wave = wave < nt;
        const int fbase = 32 * wave + 4 * h;
        const size_t goff = (size_t)(row0 + i) * NERO_HID + fbase;
        const int steps = L.n_out >> 4;
        // saved activations (for act') and the optional additive term of this lane's 32 outputs: requested before the GEMM
        float4 pa[2][4], pi[2][4];
        const bool has_inj = !first && L.inj != nullptr;
        if (!first && live_wave) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) pa[r][g] = *reinterpret_cast<const float4*>(L.a_prev + goff + (size_t)r * 32 * NERO_HID + 8 * g);
            if (has_inj) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) pi[r][g] = *reinterpret_cast<const float4*>(L.inj + goff + (size_t)r * 32 * NERO_HID + 8 * g);
            }
        }
        f32x16 acc[2];
        if (L.n_out > 0) {
            // gradient w.r.t. the aux columns of this layer (skip connections), written straight out
            if (ch.d_aux && L.w_aux_t) {
                zero2(acc);
                if (wave < L.k_aux_tiles) {
                    gemm_split(acc, reinterpret_cast<const uint4*>(L.w_aux_t) + (size_t)wave * steps * 192 + lane, actp + i * SA + 16 * h,
                               32 * SA, PLANE_A, steps);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int f = fbase + 8 * g;
                            if (f < ch.ld_daux)
                                *reinterpret_cast<float4*>(ch.d_aux + (size_t)(row0 + 32 * r + i) * ch.ld_daux + f) =
                                    make_float4(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3]);
                        }
                }
            }
            zero2(acc);
            if (live_wave)
                gemm_split(acc, reinterpret_cast<const uint4*>(L.w_main_t) + (size_t)wave * steps * 192 + lane, actp + i * SA + 16 * h,
                           32 * SA, PLANE_A, steps);
            if (first) {
                if (ch.d_init && live_wave) {
                    const int ldi = ch.ld_dinit;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int f = fbase + 8 * g;
                            if (f < ldi) {
                                float4 v = make_float4(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3]);
                                float4* dstp = reinterpret_cast<float4*>(ch.d_init + (size_t)(row0 + 32 * r + i) * ldi + f);
                                if (ch.accumulate_dinit) { const float4 o = *dstp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                                *dstp = v;
                            }
                        }
                }
                break;
            }
            __syncthreads();                               // every wave is done reading the delta planes
        } else {
            // head-only pseudo layer: the incoming gradient is the current content of the planes (exact reconstruction)
            if (first) break;
            if (live_wave) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = load_planes4(actp + (32 * r + i) * SA + (fbase + 8 * g) * 2, PLANE_A);
                        acc[r][4 * g] = v.x; acc[r][4 * g + 1] = v.y; acc[r][4 * g + 2] = v.z; acc[r][4 * g + 3] = v.w;
                    }
            }
        }
        if (live_wave) {
            char* dst = actp + i * SA + fbase * 2;
            float* scr = reinterpret_cast<float*>(smem + 3 * PLANE_A + wave * SCR_BYTES);
            const size_t boff = (size_t)row0 * NERO_HID + 32 * wave;
            if (L.act_prev == NERO_ACT_RELU) bwd_epilogue_h<NERO_ACT_RELU>(acc, pa, pi, has_inj, L, dst, boff, scr, lane, row0, i, fbase, n_rows);
            else if (L.act_prev == NERO_ACT_SOFTPLUS100) bwd_epilogue_h<NERO_ACT_SOFTPLUS100>(acc, pa, pi, has_inj, L, dst, boff, scr, lane, row0, i, fbase, n_rows);
            else bwd_epilogue_h<NERO_ACT_NONE>(acc, pa, pi, has_inj, L, dst, boff, scr, lane, row0, i, fbase, n_rows);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight-gradient GEMM:  C[n][k] = sum_r D[r][n] B[r][k]  (contraction over the batch rows; split over row slices)
//
// Both operands are row-major fp32 [rows][<=256] matrices in HBM and the MFMA wants, for every column, 8 consecutive batch
// rows per lane.  A chunk of 16 rows is staged per step: thread (col = tid & 255, rh = tid >> 8) loads rows 8rh..8rh+7 of
// its column of D and of B (dword loads, one 256-byte row segment per wave instruction), splits them into the three bf16
// planes and writes ONE 16-byte LDS store per plane into the layout [rh][col] x 16 B -- which is exactly the fragment
// order (lane (i, h) reads col 32t+i, half h), so operand reads are conflict-free ds_read_b128.  Chunks are double
// buffered (global loads of chunk q+1 in flight during the 48 MFMAs per wave of chunk q).
// 8 waves: FULL: wave (wn = w>>1, wk = w&1) owns n-tiles {2wn, 2wn+1} x k-tiles {4wk..4wk+3} (8 accumulators);
//          NARROW (k_pad <= 128): wave w owns n-tile w x k-tiles {0..3}.
// At 6 plane products per tile step the kernel needs ~6.5 TB/s of operand traffic to saturate the matrix pipe: it is HBM
// bound for full 256x256 jobs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DWS_PLANE = 2 * 256 * 16;              // one plane of one matrix: [2][256] x 16 B = 8 KB
constexpr int DWS_MAT = 3 * DWS_PLANE;               // 24 KB
constexpr int DWS_STAGE = 2 * DWS_MAT;               // D + B = 48 KB

// (branch free: out-of-range rows / columns read a clamped, valid address and are zeroed by a select, so that the whole chunk
// loop stays one basic block and the scheduler can interleave it with the MFMA stream; requires r1 >= 1, cols >= 1)
__device__ __forceinline__ void dws_fetch(float (&v)[8], const float* __restrict__ src, int ld, int cols, int r0, int r1, int col, int rh) {
    const int cc = col < cols ? col : cols - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gr = r0 + 8 * rh + j;
        const int gc = gr < r1 ? gr : r1 - 1;
        const float m = (col < cols && gr < r1) ? 1.f : 0.f;
        v[j] = src[(size_t)gc * ld + cc] * m;           // (a multiply, not a select: keeps the load unconditional)
    }
}
__device__ __forceinline__ void dws_put(char* mat, const float (&v)[8], int col, int rh) {
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    char* dst = mat + (rh * 256 + col) * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(dst + q * DWS_PLANE) = make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]);
}
struct Fr3 { uint4 p0, p1, p2; };
__device__ __forceinline__ Fr3 dws_frag(const char* mat, int tile, int i, int h) {
    const char* s = mat + (h * 256 + 32 * tile + i) * 16;
    Fr3 f;
    f.p0 = *reinterpret_cast<const uint4*>(s);
    f.p1 = *reinterpret_cast<const uint4*>(s + DWS_PLANE);
    f.p2 = *reinterpret_cast<const uint4*>(s + 2 * DWS_PLANE);
    return f;
}
__device__ __forceinline__ void mf6(f32x16& acc, const Fr3& a, const Fr3& b) {
    NERO_MF(acc, a.p2, b.p0);
    NERO_MF(acc, a.p1, b.p1);
    NERO_MF(acc, a.p0, b.p2);
    NERO_MF(acc, a.p1, b.p0);
    NERO_MF(acc, a.p0, b.p1);
    NERO_MF(acc, a.p0, b.p0);
}

template <bool NARROW>
__global__ __launch_bounds__(512, 1) void dw_split_kernel(nero_dw_job job, int n_rows, int rows_per_slice, float* __restrict__ partials,
                                                          int n_pad, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int col = tid & 255, rh = tid >> 8;
    const int r_begin = blockIdx.x * rows_per_slice;
    int r_end = r_begin + rows_per_slice;
    r_end = r_end < n_rows ? r_end : n_rows;
    const int n_tiles = n_pad >> 5, k_tiles = k_pad >> 5;
    constexpr int NA = NARROW ? 1 : 2;                   // n-tiles per wave
    const int nt0 = NARROW ? wave : 2 * (wave >> 1), kt0 = NARROW ? 0 : 4 * (wave & 1);
    f32x16 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float bsum = 0.f;                                    // bias gradient partial of column `col` (rows of this thread's half)
    const int nch = r_end > r_begin ? (r_end - r_begin + 15) / 16 : 0;
    const int total = nch * (job.d1 ? 2 : 1);
    // chunk q -> (operand pair, first row)
    auto fetch = [&](float (&vd)[8], float (&vb)[8], int q) {
        const bool second = q >= nch;
        const int r0 = r_begin + (second ? q - nch : q) * 16;
        dws_fetch(vd, second ? job.d1 : job.d0, second ? job.ldd1 : job.ldd0, job.n_out, r0, r_end, col, rh);
        dws_fetch(vb, second ? job.b1 : job.b0, second ? job.ldb1 : job.ldb0, job.k_cols, r0, r_end, col, rh);
    };
    // software pipeline: chunk q is multiplied out of LDS stage q&1 while the registers of chunk q+1 are split and stored into
    // the other stage (the compiler interleaves that VALU work with the MFMA stream) and the loads of chunk q+2 are in flight
    float ad[8], ab[8], cd[8], cb[8];                    // chunk q+1 (landed) and chunk q+2 (in flight)
    if (total > 0) {
        fetch(ad, ab, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += ad[j];
        dws_put(smem, ad, col, rh);
        dws_put(smem + DWS_MAT, ab, col, rh);
        if (total > 1) fetch(ad, ab, 1);
    }
    __syncthreads();
    for (int q = 0; q < total; ++q) {
        const char* sD = smem + (q & 1) * DWS_STAGE;
        const char* sB = sD + DWS_MAT;
        // (indices past the end are clamped: the extra fetch / store is harmless -- nobody reads that stage any more)
        fetch(cd, cb, q + 2 < total ? q + 2 : total - 1);
        {
            const float keep = (q + 1 < nch) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum = fmaf(keep, ad[j], bsum);
            char* nD = smem + ((q + 1) & 1) * DWS_STAGE;
            dws_put(nD, ad, col, rh);
            dws_put(nD + DWS_MAT, ab, col, rh);
        }
        if (NARROW) {
            if (nt0 < n_tiles) {
                const Fr3 fa = dws_frag(sD, nt0, i, h);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b < k_tiles) mf6(acc[0][b], fa, dws_frag(sB, b, i, h));
            }
        } else {
            // all 2 x 4 tiles of this wave, unconditionally (missing columns are zero planes)
            Fr3 fa[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) fa[a] = dws_frag(sD, nt0 + a, i, h);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const Fr3 fb = dws_frag(sB, kt0 + b, i, h);
#pragma unroll
                for (int a = 0; a < NA; ++a) mf6(acc[a][b], fa[a], fb);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ad[j] = cd[j]; ab[j] = cb[j]; }
        __syncthreads();
    }
    // this slice's partial C (row-major [n_pad][k_pad]) and bias partial, in the layout dw_reduce_kernel expects
    float* __restrict__ P = partials + (size_t)blockIdx.x * ((size_t)n_pad * k_pad + n_pad);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int nt = nt0 + a, kt = kt0 + b;
            if (nt < n_tiles && kt < k_tiles) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = 32 * nt + (v & 3) + 8 * (v >> 2) + 4 * h;
                    P[(size_t)row * k_pad + 32 * kt + i] = acc[a][b][v];
                }
            }
        }
    // bias: the two row-halves of a column live in threads col and col+256
    float* red = reinterpret_cast<float*>(smem);
    if (rh == 1) red[col] = bsum;
    __syncthreads();
    if (rh == 0 && col < n_pad) P[(size_t)n_pad * k_pad + col] = bsum + red[col];
}

// ---------------------------------------------------------------------------------------------------------------------
// operand packing: three bf16 planes in A-fragment order
//   out[(((t*nsteps + c)*3 + p)*64 + lane)*8 + j] = plane_p( A[32t + (lane&31)][16c + 8(lane>>5) + j] * scale )
//   transpose == 0:  A[m][k] = W[m][col0 + k]   (m < nrows, k < ncols)       forward operand
//   transpose == 1:  A[m][k] = W[k][col0 + m]   (m < ncols, k < nrows)       reverse operand
// ---------------------------------------------------------------------------------------------------------------------
__global__ void pack_split_kernel(const float* __restrict__ W, int nrows, int ld, int col0, int ncols, int transpose, float scale,
                                  int nsteps, int ntiles, uint4* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ntiles * nsteps * 64) return;
    const int lane = idx & 63, tc = idx >> 6;
    const int c = tc % nsteps, t = tc / nsteps;
    const int m = 32 * t + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float x = 0.f;
        if (!transpose) { if (m < nrows && k < ncols) x = W[(size_t)m * ld + col0 + k]; }
        else            { if (m < ncols && k < nrows) x = W[(size_t)k * ld + col0 + m]; }
        v[j] = nero_mul_rn(x, scale);                   // (rounded product: never contracted into the split's x - x0, which must see the fp32 value)
    }
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
    for (int q = 0; q < 3; ++q) out[((size_t)tc * 3 + q) * 64 + lane] = make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]);
}

// ---- batched packing: one launch for all operand images / bias copies of a network ---------------------------------------
struct PackBatch { nero_pack_job job[NERO_MAX_PACK_JOBS]; };

__device__ __forceinline__ void pack_split_one(const nero_pack_job& J, int idx) {
    const int nsteps = J.kpad >> 4;
    if (idx >= J.nt_count * nsteps * 64) return;
    const float* __restrict__ W = J.W;
    const int lane = idx & 63, tc = idx >> 6;
    const int c = tc % nsteps, t = tc / nsteps;
    const int m = 32 * t + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float x = 0.f;
        if (!J.transpose) { if (m < J.nrows && k < J.ncols) x = W[(size_t)m * J.ld + J.col0 + k]; }
        else              { if (m < J.ncols && k < J.nrows) x = W[(size_t)k * J.ld + J.col0 + m]; }
        v[j] = nero_mul_rn(x, J.scale);
    }
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    uint4* out = reinterpret_cast<uint4*>(J.out);
#pragma unroll
    for (int q = 0; q < 3; ++q) out[((size_t)tc * 3 + q) * 64 + lane] = make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]);
}
// the f32-MFMA B-operand order of mlp_engine.hip's pack_weight_kernel: out[((c*NT + nt)*64 + lane)*4 + t]
__device__ __forceinline__ void pack_f32_one(const nero_pack_job& J, int idx) {
    const int total = (J.kpad >> 3) * J.nt_count * 256;
    if (idx >= total) return;
    const int t = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
    const int nt = rest % J.nt_count, c = rest / J.nt_count;
    const int k = 8 * c + 4 * (lane >> 5) + t, n = 32 * nt + (lane & 31);
    float x = 0.f;
    if (!J.transpose) { if (n < J.nrows && k < J.ncols) x = J.W[(size_t)n * J.ld + J.col0 + k]; }
    else              { if (k < J.nrows && n < J.ncols) x = J.W[(size_t)k * J.ld + J.col0 + n]; }
    reinterpret_cast<float*>(J.out)[idx] = nero_mul_rn(x, J.scale);
}
__global__ __launch_bounds__(256) void pack_batch_kernel(PackBatch B) {
    const nero_pack_job& J = B.job[blockIdx.y];
    for (int idx = blockIdx.x * 256 + threadIdx.x;; idx += gridDim.x * 256) {
        if (J.kind == 0) { if (idx >= J.nt_count * (J.kpad >> 4) * 64) break; pack_split_one(J, idx); }
        else if (J.kind == 1) { if (idx >= (J.kpad >> 3) * J.nt_count * 256) break; pack_f32_one(J, idx); }
        else if (J.kind == 3) break;                      // fp16 two-plane operands: nero_f16_pack_batch
        else {
            if (idx >= J.nrows * J.ncols) break;
            const int r = idx / J.ncols, c = idx - r * J.ncols;
            reinterpret_cast<float*>(J.out)[(size_t)r * J.kpad + c] = J.W[(size_t)r * J.ld + c];
        }
    }
}

inline int split_lds_bytes(int wide) { return 3 * PLANE_A + 3 * 64 * (wide ? SX_W : SX_N) + (wide ? 0 : 8 * SCR_BYTES); }

}  // namespace

// ---- host side (called from mlp_engine.hip's C-ABI entry points) ------------------------------------------------------
int nero_split_pack(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale, int kpad, int nt_count,
                    void* out, hipStream_t stream) {
    if (kpad % 16) return nero_fail(NERO_ERR_ARG, "nero_pack_weight_split: kpad must be a multiple of 16");
    const int nsteps = kpad / 16, total = nt_count * nsteps * 64;
    if (total <= 0) return NERO_OK;
    hipLaunchKernelGGL(pack_split_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, W, nrows, ld, col0, ncols, transpose, scale,
                       nsteps, nt_count, reinterpret_cast<uint4*>(out));
    return nero_check_launch("nero_pack_weight_split");
}

int nero_split_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream) {
    const dim3 grid((n_rows + 63) / 64), block(512);
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(bf16x6): k_main / k_aux must be multiples of 16");
    if (ch->aux_wide) {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, split_lds_bytes(1)));
        hipLaunchKernelGGL(fwd_split_kernel<true>, grid, block, split_lds_bytes(1), stream, *ch, n_rows);
    } else {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, split_lds_bytes(0)));
        hipLaunchKernelGGL(fwd_split_kernel<false>, grid, block, split_lds_bytes(0), stream, *ch, n_rows);
    }
    return NERO_OK;
}

int nero_split_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream) {
    const dim3 grid((n_rows + 63) / 64), block(512);
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_tangent(bf16x6): k_main / k_aux must be multiples of 16");
    // (the wide-aux LDS layout has no room for the epilogue's store scratch; only the SDF network -- narrow aux -- has a tangent pass)
    if (ch->aux_wide) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_mlp_tangent(bf16x6): aux_wide chains are not supported");
    NERO_ONCE(hipFuncSetAttribute((const void*)tan_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, split_lds_bytes(0)));
    hipLaunchKernelGGL(tan_split_kernel<false>, grid, block, split_lds_bytes(0), stream, *ch, n_rows);
    return NERO_OK;
}

int nero_split_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream) {
    const dim3 grid((n_rows + 63) / 64), block(512);
    for (int l = 0; l < ch->n_layers; ++l)
        if (ch->layer[l].n_out & 15) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(bf16x6): n_out must be a multiple of 16");
    if (ch->d_aux && (ch->ld_daux & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(bf16x6): ld_daux must be a multiple of 4");
    if (ch->d_init && (ch->ld_dinit & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(bf16x6): ld_dinit must be a multiple of 4");
    const int lds = 3 * PLANE_A + 8 * SCR_BYTES;
    NERO_ONCE(hipFuncSetAttribute((const void*)bwd_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(bwd_split_kernel, grid, block, lds, stream, *ch, n_rows);
    return NERO_OK;
}

int nero_split_dw(const nero_dw_job* job, int n_rows, int rows_per_slice, int slices, float* partials, int n_pad, int k_pad,
                  hipStream_t stream) {
    const int lds = 2 * DWS_STAGE;
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (k_pad <= 128)
        hipLaunchKernelGGL(dw_split_kernel<true>, dim3(slices), dim3(512), lds, stream, *job, n_rows, rows_per_slice, partials, n_pad, k_pad);
    else
        hipLaunchKernelGGL(dw_split_kernel<false>, dim3(slices), dim3(512), lds, stream, *job, n_rows, rows_per_slice, partials, n_pad, k_pad);
    return NERO_OK;
}

int nero_split_pack_batch(const nero_pack_job* jobs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return NERO_OK;
    if (n_jobs > NERO_MAX_PACK_JOBS) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: too many jobs");
    PackBatch B;
    int max_work = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const nero_pack_job& J = jobs[i];
        if (!J.W || !J.out) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: null pointer");
        if (J.kind == 0 && (J.kpad & 15)) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: split kpad must be a multiple of 16");
        if (J.kind == 1 && (J.kpad & 7)) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: kpad must be a multiple of 8");
        B.job[i] = J;
        const int work = J.kind == 0 ? J.nt_count * (J.kpad >> 4) * 64 : J.kind == 1 ? (J.kpad >> 3) * J.nt_count * 256 : J.kind == 3 ? 0 : J.nrows * J.ncols;
        max_work = work > max_work ? work : max_work;
    }
    int bx = (max_work + 255) / 256;
    bx = bx > 32 ? 32 : bx;                              // grid-stride inside the kernel
    hipLaunchKernelGGL(pack_batch_kernel, dim3(bx, n_jobs), dim3(256), 0, stream, B);
    const int rc = nero_f16_pack_batch(jobs, n_jobs, stream);
    return rc != NERO_OK ? rc : nero_check_launch("nero_pack_batch");
}
