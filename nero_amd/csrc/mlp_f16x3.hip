// mlp_f16x3.hip -- the fused MLP-chain passes on the fp16 matrix pipe with fp32-grade arithmetic and HALF the MFMA count of
// the bf16x6 engine (gemm_mode NERO_GEMM_F16X3).
//
// An fp32 operand is carried as TWO fp16 planes of its block-scaled value xs = x * 2^-e:
//        xs = h + 2^-11 * l,      h = fp16(xs),   l = fp16((xs - h) * 2^11)            (round to nearest)
// fp16 keeps 11 significant bits, so |xs - h| <= 2^-12 |xs|, the scaled remainder has the magnitude of xs/2 (no subnormal
// trouble of its own) and |xs - h - 2^-11 l| <= 2^-24 |xs|: the pair represents xs to fp32's own half-ulp.  A product is
//        w x = [ hw hx  +  2^-11 (hw lx + lw hx) ] * 2^(ew + ex),          dropped: 2^-22 lw lx <= 2^-24 |w x|
// i.e. THREE MFMAs in two accumulator sets (H: hw hx, L: hw lx + lw hx), combined as H + 2^-11 L in the epilogue.
// ROUND 5 (default; mlp_f16_util.h, -DF16_TWO_ACC = the text above): the block scale puts the maximum at the TOP of fp16's range,
// l = fp16(xs - h) at its true scale, and all three products accumulate into ONE fp32 accumulator set.
// Range: fp16 overflows at 65504 and loses precision below 2^-14, so every operand is scaled by an exact power of two:
// activations per ROW (64 per tile; exponent of the row maximum, kept in LDS next to the planes), weights per MATRIX (exponent in
// the packed image's header).  Elements more than 2^14 below their row's maximum lose relative -- not absolute -- precision,
// which a dot product cannot see.  Scales are applied back exactly in the epilogue.
//
// Kernel shape: as mlp_split.hip (512 threads own 64 rows, transposed product, wave w = feature tile w for both 32-row halves,
// epilogue straight out of the accumulators) with: two planes instead of three (LDS 68 KB), 6 MFMAs per k-step of 16 instead
// of 12, and one more barrier per layer for the row-maximum exchange between the 8 waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

#include "mlp_f16_util.h"


// ---- accumulator-layout -> row-major global store through a wave-private LDS scratch (as mlp_split.hip) -----------------
__device__ __forceinline__ void acc_to_global(float* scr, const float4 (&q)[4], float* __restrict__ gblock, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(scr + i * SCR_LD + 8 * g + 4 * h) = q[g];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 4; ++p)
        store_ws4(gblock + (size_t)(8 * p + (lane >> 3)) * NERO_HID + 4 * (lane & 7),
                  *reinterpret_cast<const float4*>(scr + (8 * p + (lane >> 3)) * SCR_LD + 4 * (lane & 7)));
    __builtin_amdgcn_wave_barrier();
}

// rows [row0, row0+64) x first k columns (k multiple of 4, <= 256) of a row-major fp32 matrix -> scaled plane pairs + the per-row
// scale rs[row] = 2^e.  8 threads per row: each keeps its float4s in registers, the row maximum is a 3-step shuffle.  In two halves, so
// that a persistent workgroup can REQUEST the next tile while the current one still computes and convert it when it is done:
// tile_request (global loads into NV float4 registers per thread: NV = 8 covers 256 columns) and tile_commit (maximum, scale, planes).
template <int NV>
__device__ __forceinline__ void tile_request(float4 (&v)[NV], const float* __restrict__ src, int ld, int k, int row0, int n_rows, int tid) {
    const int r = tid >> 3, q = tid & 7;
    int gr = row0 + r;
    gr = gr < n_rows ? gr : n_rows - 1;
    const float* rowp = src + (size_t)gr * ld;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c4 = 4 * (q + 8 * j);
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < k) v[j] = *reinterpret_cast<const float4*>(rowp + c4);
    }
}
template <int NV>
__device__ __forceinline__ void tile_commit(char* planes, int stride, int plane_bytes, float* rs, const float4 (&v)[NV], int k, int tid) {
    const int r = tid >> 3, q = tid & 7;
    const int k16 = (k + 15) & ~15, q4 = k16 >> 2;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
    m = max_8lanes(m);
    const int e = scale_exp(m);
    const float inv = pow2i(-e);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c4 = 4 * (q + 8 * j);
        if (q + 8 * j < q4) {
            const float4 s = make_float4(v[j].x * inv, v[j].y * inv, v[j].z * inv, v[j].w * inv);
            store_planes4h(planes + r * stride + c4 * 2, plane_bytes, s);
        }
    }
    if (q == 0) rs[r] = pow2i(e);
}
__device__ __forceinline__ void load_planes_scaled(char* planes, int stride, int plane_bytes, float* rs, const float* __restrict__ src,
                                                   int ld, int k, int row0, int n_rows, int tid) {
    float4 v[8];
    tile_request<8>(v, src, ld, k, row0, n_rows, tid);
    tile_commit<8>(planes, stride, plane_bytes, rs, v, k, tid);
}

// VALU head on the current activation planes: out[r][j] = b[j] + sum_k x[r][k] W[j][k], k < hk (8 threads per row)
__device__ __forceinline__ void eval_head_f16(const char* planes, const float* rs, const float* __restrict__ w, const float* __restrict__ b,
                                              float* __restrict__ out, int n_head, int hk, int row0, int tid) {
    const int r = tid >> 3, q = tid & 7;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    float bj[4] = {0.f, 0.f, 0.f, 0.f};                    // (the four biases together, ahead of the sums: they were four serial loads at the end)
    if (b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = b[j < n_head ? j : 0];
    }
    for (int c4 = 4 * q; c4 < hk; c4 += 32) {
        const float4 x = load_planes4h(planes + r * SA + c4 * 2, PLANE_A);
        float4 ww[4];                                      // all four rows requested together (a missing head re-reads row 0): with the
#pragma unroll                                             // load inside `if (j < n_head)` every row was waited for separately
        for (int j = 0; j < 4; ++j) ww[j] = *reinterpret_cast<const float4*>(w + (j < n_head ? j : 0) * NERO_HID + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = fmaf(x.x, ww[j].x, fmaf(x.y, ww[j].y, fmaf(x.z, ww[j].z, fmaf(x.w, ww[j].w, s[j]))));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = sum_8lanes(s[j]);
    if (q == 0) {
        const float sc = rs[r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_head) out[(size_t)(row0 + r) * 4 + j] = s[j] * sc + bj[j];
    }
}

// LDS carve-up shared by the chain kernels
struct Lds {
    char* actp; char* auxp; float* rs_main; float* rs_aux; float* rmax; float* wsc; char* scr;
};
template <int SX>
__device__ __forceinline__ Lds carve(char* smem) {
    Lds l;
    l.actp = smem;
    l.auxp = smem + 2 * PLANE_A;
    char* p = l.auxp + 2 * 64 * SX;
    l.rs_main = reinterpret_cast<float*>(p);
    l.rs_aux = l.rs_main + 64;
    l.rmax = l.rs_aux + 64;                          // [64 rows][8 waves]
    l.wsc = l.rmax + 64 * 8;                         // [2 * NERO_MAX_LAYERS] block scales of the chain's packed images (mlp_f16_util.h)
    l.scr = reinterpret_cast<char*>(l.wsc + 32);
    return l;
}
inline int f16_lds_bytes(int wide) { return 2 * PLANE_A + 2 * 64 * (wide ? SX_W : SX_N) + LDS_SMALL_BYTES + 8 * SCR_BYTES; }
inline int tan_lds_bytes() { return 2 * PLANE_A + 2 * 64 * SX_N + LDS_SMALL_BYTES + 8 * 8192; }      // 150144

// ---------------------------------------------------------------------------------------------------------------------
// forward chain
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ void fwd_values(const f32x16 (&aH)[2], const f32x16 (&aL)[2], const float4 (&bq)[4], const float (&U)[2],
                                           float4 (&val)[2][4], float (&m)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        m[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = act_fwd<ACT>(fmaf(ACCV(aH, aL, r, 4 * g), U[r], bq[g].x));
            v.y = act_fwd<ACT>(fmaf(ACCV(aH, aL, r, 4 * g + 1), U[r], bq[g].y));
            v.z = act_fwd<ACT>(fmaf(ACCV(aH, aL, r, 4 * g + 2), U[r], bq[g].z));
            v.w = act_fwd<ACT>(fmaf(ACCV(aH, aL, r, 4 * g + 3), U[r], bq[g].w));
            val[r][g] = v;
            m[r] = fmaxf(m[r], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
}

// PERSISTENT (round 4): a workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (one workgroup per CU).  While the
// last GEMM of a tile runs its epilogue, the NEXT tile's input rows are already on their way from HBM into registers (tile_request;
// converted into planes when the tile is done), and after every GEMM the first three weight fragments of the next layer's first GEMM
// are requested (prefetch_w), so that neither the workgroup dispatch + a cold 64 KB HBM read per tile (~7 k cycles) nor an L2 round trip
// behind every layer's closing barrier (~900 cycles) sits in front of an idle matrix pipe.  Both requests are issued AFTER the last
// weight load of the running GEMM: vmcnt retires in order, anything issued earlier would hold the weight stream back.
#ifdef F16_NO_W_PRE
#define F16_W_PRE false
#else
#define F16_W_PRE true
#endif
// NVI = float4 registers per thread that hold the NEXT tile's input rows while the current tile finishes (tile_request): 4 covers
// k_init <= 128 (the SDF, NeRF++ trunk and light chains), 0 = the input is read at the top of the tile (256-wide inputs: eight more
// float4 registers on top of the weight fragments do not fit without spilling).
template <bool WIDE, int NVI>
__global__ __launch_bounds__(512, 1) void fwd_f16_kernel(nero_fwd_chain ch, int n_rows, int n_tiles_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SX = WIDE ? SX_W : SX_N;
    constexpr int PLANE_X = 64 * SX;
    constexpr int NVX = WIDE ? 3 : 2;                    // float4s per thread of an aux tile (<= 96 / 48 columns)
    constexpr int NVR = NVI > 0 ? NVI : 1, NXR = NVI > 0 ? NVX : 1;
    const Lds S = carve<SX>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    // the layers that have a GEMM (bit l), the first and the last of them
    unsigned gmask = 0u;
    for (int l = 0; l < ch.n_layers; ++l) gmask |= (ch.layer[l].n_tiles > 0 ? 1u : 0u) << l;
    const int first_gemm = gmask ? __builtin_ctz(gmask) : 0, last_gemm = gmask ? 31 - __builtin_clz(gmask) : -1;
    // first weight fragments of layer `Ln`'s first GEMM (the aux part when it has one) -> pw
    WF pw0, pw1, pw2;
    auto prefetch_layer = [&](const nero_fwd_layer& Ln) {      // (Ln: a by-value copy in SGPRs, load_layer)
#ifdef F16_NO_W_PRE
        return;
#endif
        if (wave >= Ln.n_tiles) return;
        const int sx = Ln.k_aux >> 4, sm = Ln.k_main >> 4;
        const int n = sx > 0 ? sx : sm;
        if (n <= 0) return;
        const float* img = sx > 0 ? Ln.w_aux : Ln.w_main;
        const uint4* wp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(img) + HDR_BYTES) + (size_t)wave * n * 128 + lane;
        const int last = n - 1;
        load_w(pw0, wp, 0);
        if (F16_PW_N > 1) load_w(pw1, wp, 1 < last ? 1 : last);
        if (F16_PW_N > 2) load_w(pw2, wp, 2 < last ? 2 : last);
    };
    float4 tin[NVR], tax[NXR];
    auto request_tile = [&](int t) {
        if (ch.init) tile_request<NVR>(tin, ch.init, ch.ld_init, ch.k_init, t * 64, n_rows, tid);
        if (ch.aux) tile_request<NXR>(tax, ch.aux, ch.ld_aux, ch.k_aux, t * 64, n_rows, tid);
    };
    int tile = blockIdx.x;
    if (tile >= n_tiles_total) return;
    if (NVI > 0) request_tile(tile);
    nero_fwd_layer Ln = load_layer(ch, first_gemm);        // descriptor of the next layer with a GEMM, one layer ahead
    int ln_idx = first_gemm;
    if (last_gemm >= 0) prefetch_layer(Ln);
    {
        WscRegs wr;
        wsc_request(wr, ch, [](const nero_fwd_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.k_main > 0 && Lx.n_tiles > 0 ? Lx.w_main : nullptr; pa = Lx.k_aux > 0 && Lx.n_tiles > 0 ? Lx.w_aux : nullptr; });
        wsc_commit(S.wsc, wr, tid);                        // (published by the first tile's barrier below)
    }
    for (; tile < n_tiles_total; tile += gridDim.x) {
    const int row0 = tile * 64;
    PH_DECL;
    if (NVI > 0) {
        if (ch.init) tile_commit<NVR>(S.actp, SA, PLANE_A, S.rs_main, tin, ch.k_init, tid);
        if (ch.aux) tile_commit<NXR>(S.auxp, SX, PLANE_X, S.rs_aux, tax, ch.k_aux, tid);
    } else {
        if (ch.init) load_planes_scaled(S.actp, SA, PLANE_A, S.rs_main, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
        if (ch.aux) load_planes_scaled(S.auxp, SX, PLANE_X, S.rs_aux, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    }
    __syncthreads();
    PH(0);
    for (int l = 0; l < ch.n_layers; ++l) {
        nero_fwd_layer L;
        if (l == ln_idx) L = Ln; else L = load_layer(ch, l);
        if (L.n_head > 0) eval_head_f16(S.actp, S.rs_main, L.head_w, L.head_b, L.head_out, L.n_head, L.head_k, row0, tid);
        if (L.n_tiles == 0) continue;
        const bool live_wave = wave < L.n_tiles;
        f32x16 aH[2], aL[2];
        zero2(aH);
        zero2(aL);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bq[g] = (live_wave && L.bias) ? *reinterpret_cast<const float4*>(L.bias + 32 * wave + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        float U[2] = {1.f, 1.f};                           // result unit of the accumulators, per 32-row half (this lane's rows i, 32+i)
        PH(1);
#if GEMM_SETPRIO == 2
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
        if (live_wave) {
            const int sm = L.k_main >> 4, sx = L.k_aux >> 4;
            if (sx > 0) {
                const float wsc = S.wsc[2 * l + 1];
                gemm_f16x3_loop(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_aux) + HDR_BYTES) + (size_t)wave * sx * 128 + lane,
                                S.auxp + i * SX + 16 * h, 32 * SX, PLANE_X, sx, F16_W_PRE, pw0, pw1, pw2);
                U[0] = wsc * S.rs_aux[i];
                U[1] = wsc * S.rs_aux[32 + i];
            }
            if (sm > 0) {
                const float wsc = S.wsc[2 * l];
                const float u0 = wsc * S.rs_main[i], u1 = wsc * S.rs_main[32 + i];
                if (sx > 0) {
                    // bring the aux partial sums into the main part's unit (exact: powers of two)
                    const float r0 = U[0] / u0, r1 = U[1] / u1;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        aH[0][v] *= r0; aH[1][v] *= r1;
#ifdef F16_TWO_ACC
                        aL[0][v] *= r0; aL[1][v] *= r1;
#endif
                    }
                }
                U[0] = u0;
                U[1] = u1;
#ifdef FWD_FIXED16
                if (sm == 16) gemm_f16x3_fixed<16>(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_main) + HDR_BYTES) + (size_t)wave * sm * 128 + lane,
                                                   S.actp + i * SA + 16 * h, 32 * SA, PLANE_A);
                else
#endif
                gemm_f16x3_loop(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_main) + HDR_BYTES) + (size_t)wave * sm * 128 + lane,
                                S.actp + i * SA + 16 * h, 32 * SA, PLANE_A, sm, F16_W_PRE && sx == 0, pw0, pw1, pw2);
            }
        }
        // requests for what comes next, behind this layer's last weight load (vmcnt retires in order): the next layer's first weight
        // fragments, or -- after the tile's last GEMM -- the next tile's input rows and the first layer's fragments again.  The
        // epilogue, two barriers and the plane conversion that follow (~6 k cycles) cover the L2 / HBM latency.
#if GEMM_SETPRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        NERO_FENCE();
        // (two separate call sites on purpose: merged into one, hipcc keeps the tile-ahead registers and the fragment registers of both
        //  branches alive together -- 256 VGPRs + 13-17 spilled instead of 229-233)
        if (l != last_gemm) {
            ln_idx = l + 1 + __builtin_ctz(gmask >> (l + 1));
            Ln = load_layer(ch, ln_idx);                   // (one batch of scalar loads while the matrix pipe drains)
            prefetch_layer(Ln);
        } else {
            if (NVI > 0 && tile + (int)gridDim.x < n_tiles_total) request_tile(tile + gridDim.x);
            ln_idx = first_gemm;
            Ln = load_layer(ch, ln_idx);
            prefetch_layer(Ln);
        }
        NERO_FENCE();
        // values, optional saves, row maxima
        float4 val[2][4];
        float m[2] = {0.f, 0.f};
        PH(2);
        if (live_wave) {
#ifdef F16_NO_EPI
            for (int r = 0; r < 2; ++r) for (int g = 0; g < 4; ++g) val[r][g] = make_float4(aH[r][4 * g], aL[r][4 * g], 0.f, 0.f);
            m[0] = m[1] = 1.f;
            if (false)
#endif
            if (L.act == NERO_ACT_RELU) fwd_values<NERO_ACT_RELU>(aH, aL, bq, U, val, m);
            else if (L.act == NERO_ACT_SOFTPLUS100) fwd_values<NERO_ACT_SOFTPLUS100>(aH, aL, bq, U, val, m);
            else fwd_values<NERO_ACT_NONE>(aH, aL, bq, U, val, m);
            PH(3);
            if (L.save) {
                float* scr = reinterpret_cast<float*>(S.scr + wave * SCR_BYTES);
                float* sblock = L.save + (size_t)row0 * NERO_HID + 32 * wave;
                acc_to_global(scr, val[0], sblock, lane);
                acc_to_global(scr, val[1], sblock + (size_t)32 * NERO_HID, lane);
            }
            if (L.relu_mask) {                              // sign bits of this lane's 2 x 16 outputs -> one word per (row, tile)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    unsigned bits = 0u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bits |= (val[r][g].x > 0.f ? 1u : 0u) << (4 * g);
                        bits |= (val[r][g].y > 0.f ? 1u : 0u) << (4 * g + 1);
                        bits |= (val[r][g].z > 0.f ? 1u : 0u) << (4 * g + 2);
                        bits |= (val[r][g].w > 0.f ? 1u : 0u) << (4 * g + 3);
                    }
                    const unsigned other = other_half(bits, h);
                    if (h == 0) L.relu_mask[(size_t)(row0 + 32 * r + i) * 8 + wave] = bits | (other << 16);
                }
            }
        }
        PH(4);
        publish_rowmax(S.rmax, m[0], m[1], wave, i, h);
        __syncthreads();                                   // row maxima visible; every wave is done reading the input planes
        PH(5);
        {
            const int e0 = scale_exp(row_max8(S.rmax, i)), e1 = scale_exp(row_max8(S.rmax, 32 + i));
            if (live_wave) {
                const float inv0 = pow2i(-e0), inv1 = pow2i(-e1);
                char* dst = S.actp + i * SA + (32 * wave + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = val[0][g], b = val[1][g];
                    store_planes4h(dst + 16 * g, PLANE_A, make_float4(a.x * inv0, a.y * inv0, a.z * inv0, a.w * inv0));
                    store_planes4h(dst + 32 * SA + 16 * g, PLANE_A, make_float4(b.x * inv1, b.y * inv1, b.z * inv1, b.w * inv1));
                }
            }
            // (nobody reads rs_main between the barrier above and the one below: the units U were taken before the GEMM)
            if (wave == 0 && h == 0) { S.rs_main[i] = pow2i(e0); S.rs_main[32 + i] = pow2i(e1); }
        }
        PH(6);
        __syncthreads();
        PH(7);
    }
    PH_END;
    __syncthreads();                                       // (a trailing head-only layer still reads the planes the next tile overwrites)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tangent chain (softplus networks):  adot_l = s_l * (W_l adot_{l-1}),  inj_l = gbar_l * beta (1-s_l) * zdot_l
// ---------------------------------------------------------------------------------------------------------------------
// shared tail of the three chain kernels: publish row maxima, barrier, rescale + store this wave's 64x32 block as plane pairs,
// new row scales, barrier
__device__ __forceinline__ void commit_planes(const Lds& S, const float4 (&val)[2][4], float m0, float m1, bool live_wave, int wave,
                                              int i, int h) {
    publish_rowmax(S.rmax, m0, m1, wave, i, h);
    __syncthreads();                                   // row maxima visible; every wave is done reading the input planes
    const int e0 = scale_exp(row_max8(S.rmax, i)), e1 = scale_exp(row_max8(S.rmax, 32 + i));
    if (live_wave) {
        const float inv0 = pow2i(-e0), inv1 = pow2i(-e1);
        char* dst = S.actp + i * SA + (32 * wave + 4 * h) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store_planes4h(dst + 16 * g, PLANE_A, scale4(val[0][g], inv0));
            store_planes4h(dst + 32 * SA + 16 * g, PLANE_A, scale4(val[1][g], inv1));
        }
    }
    if (wave == 0 && h == 0) { S.rs_main[i] = pow2i(e0); S.rs_main[32 + i] = pow2i(e1); }
    __syncthreads();
}

// aux part first (its own unit), converted into the main part's unit, then the main part: returns the unit of the result
__device__ __forceinline__ void gemm_two_sources(f32x16 (&aH)[2], f32x16 (&aL)[2], float (&U)[2], const Lds& S, const float* w_main,
                                                 const float* w_aux, float wsc_main, float wsc_aux, int sm, int sx, int SXb, int PLANE_Xb,
                                                 int wave, int lane, int i, int h) {
    if (sx > 0) {
        const float wsc = wsc_aux;
        gemm_f16x3(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w_aux) + HDR_BYTES) + (size_t)wave * sx * 128 + lane,
                   S.auxp + i * SXb + 16 * h, 32 * SXb, PLANE_Xb, sx);
        U[0] = wsc * S.rs_aux[i];
        U[1] = wsc * S.rs_aux[32 + i];
    }
    if (sm > 0) {
        const float wsc = wsc_main;
        const float u0 = wsc * S.rs_main[i], u1 = wsc * S.rs_main[32 + i];
        if (sx > 0) {
            const float r0 = U[0] / u0, r1 = U[1] / u1;    // exact: powers of two
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                        aH[0][v] *= r0; aH[1][v] *= r1;
#ifdef F16_TWO_ACC
                        aL[0][v] *= r0; aL[1][v] *= r1;
#endif
                    }
        }
        U[0] = u0;
        U[1] = u1;
        gemm_f16x3(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w_main) + HDR_BYTES) + (size_t)wave * sm * 128 + lane,
                   S.actp + i * SA + 16 * h, 32 * SA, PLANE_A, sm);
    }
}

__global__ __launch_bounds__(512, 1) void tan_f16_kernel(nero_tan_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SX = SX_N;
    constexpr int PLANE_X = 64 * SX;
    const Lds S = carve<SX>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 64;
    // the saved activations of the NEXT layer arrive by LDS-DMA in this lane's fragment order (as in the reverse kernel: 8 KB per
    // wave in place of the store scratch, which they double as once read out); the first-order signal gbar stays a register load
    char* pa_lds = S.scr + wave * 8192;
    const unsigned pa_addr = __builtin_amdgcn_readfirstlane(lds_offset_of(pa_lds));
    const size_t goff = (size_t)(row0 + i) * NERO_HID + 32 * wave + 4 * h;     // + r*32*HID + 8g
    auto prefetch_act = [&](const nero_tan_layer& Ln) {
        if (wave >= Ln.n_tiles) return;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) lds_dma16(Ln.a_saved + goff + (size_t)r * 32 * NERO_HID + 8 * g, pa_addr + (r * 4 + g) * 1024);
    };
    prefetch_act(ch.layer[0]);
    WscRegs wr;
    wsc_request(wr, ch, [](const nero_tan_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.k_main > 0 ? Lx.w_main : nullptr; pa = Lx.k_aux > 0 ? Lx.w_aux : nullptr; });
    if (ch.init) load_planes_scaled(S.actp, SA, PLANE_A, S.rs_main, ch.init, ch.ld_init, ch.k_init, row0, n_rows, tid);
    if (ch.aux) load_planes_scaled(S.auxp, SX, PLANE_X, S.rs_aux, ch.aux, ch.ld_aux, ch.k_aux, row0, n_rows, tid);
    wsc_commit(S.wsc, wr, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_tan_layer L = load_layer(ch, l);
        const bool live_wave = wave < L.n_tiles;
        const size_t boff = (size_t)row0 * NERO_HID + 32 * wave;
        float4 pg[2][4];
        const bool want_inj = L.inj != nullptr;            // (round 4 default: NULL -- the reverse kernel forms the injection, nero_bwd_layer.inj_adot)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) pg[r][g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live_wave && want_inj) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) pg[r][g] = *reinterpret_cast<const float4*>(L.gbar + goff + (size_t)r * 32 * NERO_HID + 8 * g);
        }
        f32x16 aH[2], aL[2];
        zero2(aH);
        zero2(aL);
        float U[2] = {1.f, 1.f};
        if (live_wave) gemm_two_sources(aH, aL, U, S, L.w_main, L.w_aux, S.wsc[2 * l], S.wsc[2 * l + 1], L.k_main >> 4, L.k_aux >> 4, SX, PLANE_X, wave, lane, i, h);
        float4 val[2][4];
        float m[2] = {0.f, 0.f};
        if (live_wave) {
            float4 pa[2][4];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's LDS-DMA of the tile has landed
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) pa[r][g] = *reinterpret_cast<const float4*>(pa_lds + (r * 4 + g) * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // read out: the buffer is the store scratch from here on
            float* scr = reinterpret_cast<float*>(pa_lds);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool live = (row0 + 32 * r + i) < n_rows;
                float4 ijq[4], adq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = pa[r][g], gb = pg[r][g];
                    float4 ad, ij;
                    tan_elem(a.x, ACCV(aH, aL, r, 4 * g) * U[r], gb.x, live, ad.x, ij.x);
                    tan_elem(a.y, ACCV(aH, aL, r, 4 * g + 1) * U[r], gb.y, live, ad.y, ij.y);
                    tan_elem(a.z, ACCV(aH, aL, r, 4 * g + 2) * U[r], gb.z, live, ad.z, ij.z);
                    tan_elem(a.w, ACCV(aH, aL, r, 4 * g + 3) * U[r], gb.w, live, ad.w, ij.w);
                    val[r][g] = ad;
                    m[r] = fmaxf(m[r], amax4(ad));
                    adq[g] = live ? ad : make_float4(0.f, 0.f, 0.f, 0.f);
                    ijq[g] = ij;
                }
                acc_to_global(scr, adq, L.adot + boff + (size_t)r * 32 * NERO_HID, lane);
                if (want_inj) acc_to_global(scr, ijq, L.inj + boff + (size_t)r * 32 * NERO_HID, lane);
            }
        }
        if (l + 1 < ch.n_layers) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the scratch reads are done before the DMA may land
            prefetch_act(ch.layer[l + 1]);
        }
        commit_planes(S, val, m[0], m[1], live_wave, wave, i, h);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse chain:  delta_{l-1} = (delta_l W_l [+ dy_head W_head]) * act'(a_{l-1}) [+ inj_{l-1}]
// ---------------------------------------------------------------------------------------------------------------------
// (bwd_values / bwd_values_h: mlp_f16_util.h, shared with mlp_f16p.hip)
// LDS of the reverse kernel: planes | row scales, row maxima | PA: the saved activations of the layer the walk reaches NEXT,
// 8 KB per wave in this lane's fragment order [r][g][lane] x 16 B, brought in by LDS-DMA (global_load_lds_dwordx4) while the
// current layer computes; between its read-out and the next request the same 8 KB serve as the wave's store-transposition scratch.  Without it the 64 KB per tile and layer were requested in front of
// the k-loop, every CU at once, and the weight stream queued behind them (vmcnt retires in order): 14.9k instead of 8.3k cycles
// per softplus layer (profiles/r02_phase_timing.txt); it also frees 32 VGPRs through the GEMM.
constexpr int BWD_PA_BYTES = 8 * 8192;
inline int bwd_lds_bytes() { return 2 * PLANE_A + LDS_SMALL_BYTES + BWD_PA_BYTES; }      // 135808

// FIXED: every reverse GEMM of the chain contracts over exactly 256 outputs (16 k-steps) -- the k-loop is then fully unrolled
// (no ring rotation, clamps or branches: 16 % faster on 256-wide chains); chains with other widths take the generic loop.  The host
// pads wide layers to 256 (nero_amd/chain.py::_rev_k: the SDF's 217-wide layer gets two all-zero k-steps), so both SDF reverse passes
// qualify: 1.47 -> 1.37 ms and 1.84 -> 1.74 ms per step.
// (Round 3, measured and dropped: requesting the INJECTIONS of the next layer one layer ahead into 32 registers -- they fit, 219 VGPRs,
// no spill -- instead of reading them in the epilogue: the second-order reverse pass went from 1.74 to 2.69 ms.  vmcnt retires in
// order, so the weight stream of the next GEMM queued behind eight 1 KB HBM loads per wave, exactly what the LDS-DMA of the saved
// activations had been introduced to avoid; there is no LDS left for a second DMA target: 135.7 of 160 KB.)
template <bool FIXED, bool WALK>
__global__ __launch_bounds__(512, 1) void bwd_f16_kernel(nero_bwd_chain ch, int n_rows, int n_tiles_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds S;
    S.actp = smem;
    S.auxp = nullptr;                                  // (the aux planes are unused by the reverse walk)
    S.rs_main = reinterpret_cast<float*>(smem + 2 * PLANE_A);
    S.rs_aux = S.rs_main + 64;
    S.rmax = S.rs_aux + 64;
    S.wsc = S.rmax + 64 * 8;
    S.scr = nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    char* pa_lds = reinterpret_cast<char*>(S.wsc + 32) + wave * 8192;
    {
        WscRegs wr;
        wsc_request(wr, ch, [](const nero_bwd_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.n_out > 0 ? Lx.w_main_t : nullptr; pa = Lx.n_out > 0 ? Lx.w_aux_t : nullptr; });
        wsc_commit(S.wsc, wr, tid);                        // (published by the first tile's barrier)
    }
    const unsigned pa_addr = __builtin_amdgcn_readfirstlane(lds_offset_of(pa_lds));
    // PERSISTENT walk (round 5; NERO_F16_PERSIST bit 1): with the second accumulator set gone (mlp_f16_util.h) the tile loop fits the
    // register budget -- round 4 measured it 2-5 % SLOWER because it spilled 12-20 VGPRs.  n_tiles_total = 0: one tile per workgroup.
    for (int tile = blockIdx.x; tile < (WALK ? n_tiles_total : (int)gridDim.x); tile += gridDim.x) {
    const int row0 = tile * 64;
    const size_t goff = (size_t)(row0 + i) * NERO_HID + 32 * wave + 4 * h;
    unsigned mbits[2] = {0u, 0u};                      // ReLU sign masks of the layer being processed
    // request what the epilogue of layer `Ln` needs of its input activation: the sign words (registers) or the fp32 tile (LDS-DMA)
    auto prefetch_act = [&](const nero_bwd_layer& Ln) {
        if (Ln.a_prev == nullptr || wave >= Ln.k_main_tiles) return;
        if (Ln.mask_prev && Ln.act_prev == NERO_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 2; ++r) mbits[r] = Ln.mask_prev[(size_t)(row0 + 32 * r + i) * 8 + wave];
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    lds_dma16(Ln.a_prev + goff + (size_t)r * 32 * NERO_HID + 8 * g, pa_addr + (r * 4 + g) * 1024);
        }
    };
    prefetch_act(ch.layer[ch.n_layers - 1]);
    if (ch.dy) load_planes_scaled(S.actp, SA, PLANE_A, S.rs_main, ch.dy, ch.ld_dy, ch.k_dy, row0, n_rows, tid);
    else {
        for (int idx = tid; idx < 2 * PLANE_A / 16; idx += 512) reinterpret_cast<uint4*>(S.actp)[idx] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 64) S.rs_main[tid] = 1.f;
    }
    __syncthreads();
    PH_DECL;
    for (int l = ch.n_layers - 1; l >= 0; --l) {
        const nero_bwd_layer L = load_layer(ch, l);
        const bool first = (L.a_prev == nullptr);
        if (first && ch.d_init == nullptr && !(ch.d_aux && L.w_aux_t)) break;
        const int nt = L.k_main_tiles;
        const bool live_wave = wave < nt;
        const int fbase = 32 * wave + 4 * h;
        const size_t boff = (size_t)row0 * NERO_HID + 32 * wave;
        const int steps = L.n_out >> 4;
        const bool has_inj = !first && L.inj != nullptr;
        float4 gq[2][4];                                   // incoming gradient of this lane's outputs, true units
        float4 ijp[2][4];                                  // injections of this lane's outputs, requested inside the GEMM (FIXED)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) ijp[r][g] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float rs0 = S.rs_main[i], rs1 = S.rs_main[32 + i];
        PH(1);
        if (L.n_out > 0) {
            f32x16 aH[2], aL[2];
            if (ch.d_aux && L.w_aux_t) {
                zero2(aH);
                zero2(aL);
                if (wave < L.k_aux_tiles) {
                    const float wsc = S.wsc[2 * l + 1];
                    const uint4* wpx = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_aux_t) + HDR_BYTES) + (size_t)wave * steps * 128 + lane;
                    if constexpr (FIXED) gemm_f16x3_fixed<16>(aH, aL, wpx, S.actp + i * SA + 16 * h, 32 * SA, PLANE_A);
                    else gemm_f16x3(aH, aL, wpx, S.actp + i * SA + 16 * h, 32 * SA, PLANE_A, steps);
                    const float u[2] = {wsc * rs0, wsc * rs1};
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int f = fbase + 8 * g;
                            if (f < ch.ld_daux)
                                *reinterpret_cast<float4*>(ch.d_aux + (size_t)(row0 + 32 * r + i) * ch.ld_daux + f) =
                                    make_float4(ACCV(aH, aL, r, 4 * g) * u[r], ACCV(aH, aL, r, 4 * g + 1) * u[r],
                                                ACCV(aH, aL, r, 4 * g + 2) * u[r], ACCV(aH, aL, r, 4 * g + 3) * u[r]);
                        }
                }
            }
            zero2(aH);
            zero2(aL);
            float u[2] = {1.f, 1.f};
            if (live_wave) {
                const float wsc = S.wsc[2 * l];
                const uint4* wpm = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_main_t) + HDR_BYTES) + (size_t)wave * steps * 128 + lane;
                // (Round 4, measured and dropped: the injections requested INSIDE the GEMM, behind its last weight request, so that the
                //  epilogue does not wait for them -- the second-order pass stayed at 1.55 ms: it is bound by its HBM traffic, 3 KB per row
                //  and layer at 4.6 TB/s, not by that round trip.)
                if constexpr (FIXED) gemm_f16x3_fixed<16>(aH, aL, wpm, S.actp + i * SA + 16 * h, 32 * SA, PLANE_A);
                else gemm_f16x3(aH, aL, wpm, S.actp + i * SA + 16 * h, 32 * SA, PLANE_A, steps);
                u[0] = wsc * rs0;
                u[1] = wsc * rs1;
            }
            PH(2);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    gq[r][g] = make_float4(ACCV(aH, aL, r, 4 * g) * u[r], ACCV(aH, aL, r, 4 * g + 1) * u[r],
                                           ACCV(aH, aL, r, 4 * g + 2) * u[r], ACCV(aH, aL, r, 4 * g + 3) * u[r]);
            if (first) {
                if (ch.d_init && live_wave) {
                    const int ldi = ch.ld_dinit;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int f = fbase + 8 * g;
                            if (f < ldi) {
                                float4 v = gq[r][g];
                                float4* dstp = reinterpret_cast<float4*>(ch.d_init + (size_t)(row0 + 32 * r + i) * ldi + f);
                                if (ch.accumulate_dinit) { const float4 o = *dstp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                                *dstp = v;
                            }
                        }
                }
                break;
            }
        } else {
            // head-only pseudo layer: the incoming gradient is the current content of the planes
            if (first) break;
            if (live_wave) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gq[0][g] = scale4(load_planes4h(S.actp + i * SA + (fbase + 8 * g) * 2, PLANE_A), rs0);
                    gq[1][g] = scale4(load_planes4h(S.actp + (32 + i) * SA + (fbase + 8 * g) * 2, PLANE_A), rs1);
                }
            }
        }
        float4 val[2][4];
        float m[2] = {0.f, 0.f};
        PH(3);
        if (live_wave) {
            float4 pa[2][4];                               // saved activations of this lane's outputs (ReLU: 1 / 0 from the sign mask)
            if (L.mask_prev && L.act_prev == NERO_ACT_RELU) {   // 4 bytes per (row, tile) instead of 128: only the sign is needed
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned bits = mbits[r] >> (16 * h);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pa[r][g] = make_float4((bits >> (4 * g)) & 1u ? 1.f : 0.f, (bits >> (4 * g + 1)) & 1u ? 1.f : 0.f,
                                               (bits >> (4 * g + 2)) & 1u ? 1.f : 0.f, (bits >> (4 * g + 3)) & 1u ? 1.f : 0.f);
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA of the tile has landed
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) pa[r][g] = *reinterpret_cast<const float4*>(pa_lds + (r * 4 + g) * 1024 + lane * 16);
            }
            constexpr bool PRE = false;
            if (L.act_prev == NERO_ACT_RELU) bwd_values_h<NERO_ACT_RELU, PRE>(gq, pa, goff, has_inj, L, row0, i, fbase, n_rows, val, m, ijp);
            else if (L.act_prev == NERO_ACT_SOFTPLUS100) bwd_values_h<NERO_ACT_SOFTPLUS100, PRE>(gq, pa, goff, has_inj, L, row0, i, fbase, n_rows, val, m, ijp);
            else bwd_values_h<NERO_ACT_NONE, PRE>(gq, pa, goff, has_inj, L, row0, i, fbase, n_rows, val, m, ijp);
            PH(4);
        }
        if (live_wave && L.delta_prev) {                   // (the PA buffer has been read out: it is the transposition scratch now)
            float* scr = reinterpret_cast<float*>(pa_lds);
            acc_to_global(scr, val[0], L.delta_prev + boff, lane);
            acc_to_global(scr, val[1], L.delta_prev + boff + (size_t)32 * NERO_HID, lane);
        }
        if (l > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the scratch reads are done before the DMA may land
            prefetch_act(ch.layer[l - 1]);
        }
        PH(5);
        commit_planes(S, val, m[0], m[1], live_wave, wave, i, h);
        PH(6);
    }
    PH_END;
    if (!WALK) break;
    __syncthreads();                                       // (the next tile's dy overwrites the planes every wave has just read)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// operand packing: header + two fp16 planes in A-fragment order
//   image = [256-byte header: float 2^ew, uint bits(max |A|)] [ (((t*nsteps + c)*2 + p)*64 + lane) * 16 bytes ]
//   plane_p of A[32t + (lane&31)][16c + 8(lane>>5) + j] * 2^-ew;   transpose as in nero_pack_weight_split
// Two passes per batch: pack_max (atomicMax of |A| into the zero-initialised header) and the pack itself.
// ---------------------------------------------------------------------------------------------------------------------
struct PackBatchH { nero_pack_job job[NERO_MAX_PACK_JOBS]; };

__device__ __forceinline__ float pack_elem(const nero_pack_job& J, int m, int k) {
    float x = 0.f;
    if (!J.transpose) { if (m < J.nrows && k < J.ncols) x = J.W[(size_t)m * J.ld + J.col0 + k]; }
    else              { if (m < J.ncols && k < J.nrows) x = J.W[(size_t)k * J.ld + J.col0 + m]; }
    return nero_mul_rn(x, J.scale);                    // (the fp32 value of the scaled weight: not contracted into the split's remainder)
}

__global__ __launch_bounds__(256) void pack_max_kernel(PackBatchH B) {
    const nero_pack_job& J = B.job[blockIdx.y];
    if (J.kind != 3) return;
    const int nsteps = J.kpad >> 4, total = J.nt_count * nsteps * 64;
    float m = 0.f;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int lane = idx & 63, tc = idx >> 6;
        const int c = tc % nsteps, t = tc / nsteps;
        const int mm = 32 * t + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(pack_elem(J, mm, k0 + j)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(J.out) + 1, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void pack_f16_kernel(PackBatchH B) {
    const nero_pack_job& J = B.job[blockIdx.y];
    if (J.kind != 3) return;
    const int nsteps = J.kpad >> 4, total = J.nt_count * nsteps * 64;
    const int e = scale_exp(__uint_as_float(reinterpret_cast<const unsigned*>(J.out)[1]));
    const float inv = pow2i(-e);
    uint4* out = reinterpret_cast<uint4*>(reinterpret_cast<char*>(J.out) + HDR_BYTES);
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(J.out)[0] = pow2i(e);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int lane = idx & 63, tc = idx >> 6;
        const int c = tc % nsteps, t = tc / nsteps;
        const int mm = 32 * t + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
        unsigned hp[4], lp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2h(pack_elem(J, mm, k0 + 2 * j) * inv, pack_elem(J, mm, k0 + 2 * j + 1) * inv, hp[j], lp[j]);
        out[((size_t)tc * 2 + 0) * 64 + lane] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        out[((size_t)tc * 2 + 1) * 64 + lane] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    }
}

}  // namespace

// ---- host side (dispatched from mlp_engine.hip / mlp_split.hip) --------------------------------------------------------------
#ifdef F16_PHASE_TIMING
extern "C" int nero_debug_phases(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
    return 0;
}
#endif
int nero_f16_pack_batch(const nero_pack_job* jobs, int n_jobs, hipStream_t stream) {
    PackBatchH B;
    int max_work = 1, any = 0;
    for (int i = 0; i < n_jobs; ++i) {
        B.job[i] = jobs[i];
        if (jobs[i].kind != 3) continue;
        any = 1;
        if (jobs[i].kpad & 15) return nero_fail(NERO_ERR_ARG, "nero_pack_batch: f16x3 kpad must be a multiple of 16");
        const int work = jobs[i].nt_count * (jobs[i].kpad >> 4) * 64;
        max_work = work > max_work ? work : max_work;
    }
    if (!any) return NERO_OK;
    int bx = (max_work + 255) / 256;
    bx = bx > 32 ? 32 : bx;
    hipLaunchKernelGGL(pack_max_kernel, dim3(bx, n_jobs), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(pack_f16_kernel, dim3(bx, n_jobs), dim3(256), 0, stream, B);
    return nero_check_launch("nero_pack_batch(f16x3)");
}

#ifndef F16_TILE_AHEAD
#define F16_TILE_AHEAD 1
#endif
// persistent chain kernels: one workgroup per CU, each walks its share of the 64-row tiles
static int nero_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}
// workgroups of a chain launch over n_tiles tiles.  NERO_F16_PERSIST (bit 0 = the forward kernel, default 1; the reverse and tangent kernels measured 2-5 % SLOWER as persistent walks
// -- their register budget has no room for the loop state: 12-20 spilled VGPRs, 40-50 spilled SGPRs -- and stayed as they were) is an experiment
// switch: a cleared bit launches one workgroup per tile, i.e. the same kernel without the persistent walk.
static int nero_chain_grid(int n_tiles, int kind_bit) {
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("NERO_F16_PERSIST"); mask = e ? atoi(e) : 1; }
    const int cus = nero_cu_count();
    return ((mask >> kind_bit) & 1) && n_tiles > cus ? cus : n_tiles;
}

// NERO_F16_PAIRED: bit 0 / 1 / 2 = the forward / tangent / reverse pass runs on the two-workgroups-per-CU kernels of mlp_f16p.hip when the
// launch is large enough to keep two of them on every CU for several rounds (more than 4 tiles per CU: below that the 4-wave workgroup's
// longer walk through a tile is the critical path -- 512 rays: forward 1.95 -> 2.12 ms); bit 3 = whatever the size (tests).  Narrow-aux
// chains only: the wide aux operand would be converted from global memory inside its k-steps, which costs more than it hides
// (measured: forward classes 9.32 -> 9.54 ms with the wide-aux SDF chain on the paired kernel, gpurun_out/r05/paired_wide_ab.txt).
// Measured (profiles/r05_paired_ab.txt, 4096 rays, same box): forward 9.19 -> 8.80 ms, tangent 1.61 -> 1.45 ms, reverse 8.88 -> 9.46 ms
// (-> 8.8 with the batched epilogue loads, still no gain: its epilogue wants the saved activations the 512-thread kernel brings in by
// LDS-DMA under the GEMM, for which two workgroups leave no LDS) => default 3.
#ifndef NERO_F16_PAIRED_DEFAULT
#define NERO_F16_PAIRED_DEFAULT 3
#endif
static int g_paired_mask = -1;
static int nero_paired_mask() {
    if (g_paired_mask < 0) { const char* e = getenv("NERO_F16_PAIRED"); g_paired_mask = (e ? atoi(e) : NERO_F16_PAIRED_DEFAULT) & 15; }
    return g_paired_mask;
}
static bool nero_paired(int kind_bit, int n_rows) {
    const int m = nero_paired_mask();
    return ((m >> kind_bit) & 1) && ((m & 8) || (n_rows + 63) / 64 > 4 * nero_cu_count());
}
extern "C" int nero_f16_paired(int mask) {           // mask >= 0: select; returns the previous selection (include/nero_hip.h)
    const int prev = nero_paired_mask();
    if (mask >= 0) g_paired_mask = mask & 15;
    return prev;
}

template <bool WIDE, int NVI>
static void launch_fwd(const nero_fwd_chain* ch, int n_rows, int n_tiles, dim3 grid, hipStream_t stream) {
    NERO_ONCE(hipFuncSetAttribute((const void*)fwd_f16_kernel<WIDE, NVI>, hipFuncAttributeMaxDynamicSharedMemorySize, f16_lds_bytes(WIDE ? 1 : 0)));
    hipLaunchKernelGGL((fwd_f16_kernel<WIDE, NVI>), grid, dim3(512), f16_lds_bytes(WIDE ? 1 : 0), stream, *ch, n_rows, n_tiles);
}

// NERO_F16_ROWOWNER: forward chains WITHOUT saves / masks on the row-owner kernel of mlp_f16r.hip (a wave owns 32 rows and all features, planes
// in registers, weights through an LDS-DMA ring): 1 = launches of at least 128 rows per CU, 3 = every launch (tests), 0 = off.
#ifndef NERO_F16_ROWOWNER_DEFAULT
#define NERO_F16_ROWOWNER_DEFAULT 0
#endif
static int g_rowowner = -1;
static int nero_rowowner_mask() {
    if (g_rowowner < 0) { const char* e = getenv("NERO_F16_ROWOWNER"); g_rowowner = (e ? atoi(e) : NERO_F16_ROWOWNER_DEFAULT) & 3; }
    return g_rowowner;
}
extern "C" int nero_f16_rowowner(int mask) {           // mask >= 0: select; returns the previous selection (include/nero_hip.h)
    const int prev = nero_rowowner_mask();
    if (mask >= 0) g_rowowner = mask & 3;
    return prev;
}

int nero_f16_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream) {
    {
        const int m = nero_rowowner_mask();
        if ((m & 1) && ((m & 2) || n_rows >= 128 * nero_cu_count()) && nero_f16r_covers(ch)) {
            for (int l = 0; l < ch->n_layers; ++l)
                if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
                    return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3): k_main / k_aux must be multiples of 16");
            return nero_f16r_forward(ch, n_rows, stream);
        }
    }
    if (nero_paired(0, n_rows) && !ch->aux_wide) return nero_f16p_forward(ch, n_rows, stream);
    const int n_tiles = (n_rows + 63) / 64, cus = nero_cu_count();
    const dim3 grid(nero_chain_grid(n_tiles, 0));
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3): k_main / k_aux must be multiples of 16");
    // the next tile's input travels in registers while the current tile finishes when it is narrow enough (k_init <= 128) and the
    // launch has more tiles than workgroups
    const bool ahead = F16_TILE_AHEAD && (!ch->init || ch->k_init <= 128) && (int)grid.x < n_tiles;
    (void)cus;
    if (ch->aux_wide) { if (ahead) launch_fwd<true, 4>(ch, n_rows, n_tiles, grid, stream); else launch_fwd<true, 0>(ch, n_rows, n_tiles, grid, stream); }
    else              { if (ahead) launch_fwd<false, 4>(ch, n_rows, n_tiles, grid, stream); else launch_fwd<false, 0>(ch, n_rows, n_tiles, grid, stream); }
    return NERO_OK;
}

int nero_f16_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream) {
    if (nero_paired(1, n_rows) && !ch->aux_wide) return nero_f16p_tangent(ch, n_rows, stream);
    const dim3 grid((n_rows + 63) / 64), block(512);
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_tangent(f16x3): k_main / k_aux must be multiples of 16");
    if (ch->aux_wide) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_mlp_tangent(f16x3): aux_wide chains are not supported");
    NERO_ONCE(hipFuncSetAttribute((const void*)tan_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tan_lds_bytes()));
    hipLaunchKernelGGL(tan_f16_kernel, grid, block, tan_lds_bytes(), stream, *ch, n_rows);
    return NERO_OK;
}

// NERO_F16_PAIRED_MASKS (round 6; default 1): reverse chains that need sign words only (ReLU / identity: the predictors, the NeRF++ networks)
// on the paired kernel's MASKS instantiation when the launch is large; the softplus chains keep the 512-thread kernel and its LDS-DMA.
static bool nero_paired_masks(int n_rows) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("NERO_F16_PAIRED_MASKS"); on = e ? atoi(e) : 1; }
    return on && ((on & 2) || (n_rows + 63) / 64 > 4 * nero_cu_count());
}
int nero_f16_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream) {
    if (nero_paired(2, n_rows)) return nero_f16p_backward(ch, n_rows, stream);
    if (nero_paired_masks(n_rows) && nero_f16p_masks_only(ch)) return nero_f16p_backward(ch, n_rows, stream);
    const dim3 grid((n_rows + 63) / 64), block(512);
    for (int l = 0; l < ch->n_layers; ++l)
        if (ch->layer[l].n_out & 15) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3): n_out must be a multiple of 16");
    if (ch->d_aux && (ch->ld_daux & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3): ld_daux must be a multiple of 4");
    if (ch->d_init && (ch->ld_dinit & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3): ld_dinit must be a multiple of 4");
    bool fixed = true;
    for (int l = 0; l < ch->n_layers; ++l) fixed = fixed && (ch->layer[l].n_out == 0 || ch->layer[l].n_out == 256);
    NERO_ONCE(hipFuncSetAttribute((const void*)bwd_f16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes()));
    NERO_ONCE(hipFuncSetAttribute((const void*)bwd_f16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes()));
    NERO_ONCE(hipFuncSetAttribute((const void*)bwd_f16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes()));
    NERO_ONCE(hipFuncSetAttribute((const void*)bwd_f16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes()));
    const int n_tiles = (n_rows + 63) / 64;
    const dim3 pgrid(nero_chain_grid(n_tiles, 1));
    if ((int)pgrid.x < n_tiles) {
        if (fixed) hipLaunchKernelGGL((bwd_f16_kernel<true, true>), pgrid, block, bwd_lds_bytes(), stream, *ch, n_rows, n_tiles);
        else hipLaunchKernelGGL((bwd_f16_kernel<false, true>), pgrid, block, bwd_lds_bytes(), stream, *ch, n_rows, n_tiles);
    } else {
        if (fixed) hipLaunchKernelGGL((bwd_f16_kernel<true, false>), grid, block, bwd_lds_bytes(), stream, *ch, n_rows, n_tiles);
        else hipLaunchKernelGGL((bwd_f16_kernel<false, false>), grid, block, bwd_lds_bytes(), stream, *ch, n_rows, n_tiles);
    }
    return NERO_OK;
}
