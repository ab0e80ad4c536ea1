// mlp_f16p.hip -- the fused MLP-chain passes of the fp16 two-plane arithmetic (mlp_f16x3.hip: three MFMAs per fp32 multiply-add,
// per-row block scaling) organised for TWO resident workgroups per CU.  Not a gemm_mode: an execution detail of NERO_GEMM_F16X3, chosen
// per pass and launch size by nero_f16_forward / _tangent / _backward (mlp_f16x3.hip: NERO_F16_PAIRED, nero_f16_paired).
//
// HISTORY.  Built in round 2 (forward default of rounds 2-3: -3 % on the training step), REMOVED in round 4 because one launch in three
// returned a wrong partial sum in a quarter of the lanes of one accumulator pair (docs/experiments.md 3i) and the mechanism was not
// found.  Round 5 found it: packed fp32 VALU instructions -- the epilogue's v_pk_fma_f32 -- next to the OTHER
// workgroup's MFMAs on the same SIMD.  The library is built without packed fp32 now (common.h), so the engine is back, ported to the
// one-accumulator plane format and the fused-injection protocol of the second-order pass, behind the bit-reproducibility tests that caught
// it (tests/test_determinism.py) and a bit-for-bit comparison with the 512-thread kernels (tests/test_paired_engine.py).  Measured
// (profiles/r05_paired_ab.txt): forward -4 %, tangent -10 %, reverse +-0 => forward + tangent of large launches by default.
// Measured and dropped here: the next tile's / next layer's first weight fragments requested behind the running GEMM (prefetch_layer of
// the 512-thread forward kernel): 242 instead of 224 VGPRs, forward class 6.82 -> 6.90 ms in 4 of 4 interleaved runs -- the sibling
// workgroup already covers that L2 round trip.
//
// Why: the 512-thread kernels of mlp_f16x3.hip hold one workgroup per CU, whose 8 waves walk the phases of a layer in lock step
// (GEMM -> activation / saves -> row-maximum exchange -> plane conversion, two barriers), so the matrix pipe idles through every
// epilogue, barrier and HBM wait: the per-phase shader-clock profile (scripts/phase_timing.py, profiles/r02_phase_timing.txt)
// shows 57 % of a forward layer inside the GEMM loop and 30-45 % of a reverse layer, the rest serial.  Here a workgroup is
// 256 threads = 4 waves that still own 64 rows; wave w computes the feature tiles w and w + 4 one after the other (same packed
// operand images, same per-row scaling, same results bit for bit).  LDS per workgroup drops to 79.4 KB -- the aux operand
// (skip / direction inputs, <= 96 columns) is converted from global memory inside its k-steps instead of living in LDS, the
// store-transposition scratch covers 16 rows at a time -- so two workgroups share a CU and one's epilogue, barriers and HBM
// latency are covered by the other's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

#include "mlp_f16_util.h"

// NW = waves per workgroup.  4 (round 2 / 5): wave w computes the feature tiles w and w + 4 one after the other, 256 registers per wave,
// two waves per SIMD.  8 (round 6): wave w computes tile w only, at most 128 registers per wave, FOUR waves per SIMD from two workgroups --
// the occupancy at which scripts/probe/rowowner_probe.hip measures today's layer structure 13 % faster than with one workgroup per CU.
// reverse kernel: the saved activations of a wave's first / second feature tile are requested in front of (true) or behind (false) its GEMM
#ifndef P_PA_EARLY_A
#define P_PA_EARLY_A true
#endif
#ifndef P_PA_EARLY_B
#define P_PA_EARLY_B true
#endif

struct LdsP { char* actp; float* rs_main; float* rs_aux; float* rmax; float* wsc; char* scr; };
__device__ __forceinline__ LdsP carve_p(char* smem) {
    LdsP l;
    l.actp = smem;
    l.rs_main = reinterpret_cast<float*>(smem + 2 * PLANE_A);
    l.rs_aux = l.rs_main + 64;
    l.rmax = l.rs_aux + 64;                            // [64 rows][8 tiles]
    l.wsc = l.rmax + 64 * 8;                           // block scales of the chain's packed images (mlp_f16_util.h: wsc_request)
    l.scr = reinterpret_cast<char*>(l.wsc + 32);
    return l;
}
#ifndef NERO_F16_PW_DEFAULT
#define NERO_F16_PW_DEFAULT 4
#endif
#ifndef P_LDS_EXTRA
#define P_LDS_EXTRA 0                                   // (timing experiment: > 2560 forces ONE workgroup per CU)
#endif
inline int p_lds_bytes() { return 2 * PLANE_A + LDS_SMALL_BYTES + 4 * SCRP_BYTES + P_LDS_EXTRA; }      // 79488 (NW = 8: 8 x 8-row scratch, the same)

struct Ctx { LdsP S; int wave, lane, i, h, row0, n_rows; float* scr; };

// rows [row0, row0+64) x first k columns of a row-major fp32 matrix -> scaled plane pairs + per-row scale (NW threads per row)
template <int NW>
__device__ __forceinline__ void load_planes_scaled_p(char* planes, float* rs, const float* __restrict__ src, int ld, int k, int row0,
                                                     int n_rows, int tid) {
    constexpr int NV = 64 / NW;
    const int r = tid / NW, q = tid % NW;
    const int k16 = (k + 15) & ~15, q4 = k16 >> 2;
    int gr = row0 + r;
    gr = gr < n_rows ? gr : n_rows - 1;
    const float* rowp = src + (size_t)gr * ld;
    float4 v[NV];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c4 = 4 * (q + NW * j);
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < k) v[j] = *reinterpret_cast<const float4*>(rowp + c4);
        m = fmaxf(m, amax4(v[j]));
    }
    m = NW == 8 ? max_8lanes(m) : max_4lanes(m);
    const int e = scale_exp(m);
    const float inv = pow2i(-e);
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (q + NW * j < q4) store_planes4h(planes + r * SA + (q + NW * j) * 8, PLANE_A, scale4(v[j], inv));
    if (q == 0) rs[r] = pow2i(e);
}

// per-row scale of the aux operand (its planes are built on the fly, gemm_aux_global)
template <int NW>
__device__ __forceinline__ void aux_row_scales_p(float* rs, const float* __restrict__ src, int ld, int k, int row0, int n_rows, int tid) {
    const int r = tid / NW, q = tid % NW;
    int gr = row0 + r;
    gr = gr < n_rows ? gr : n_rows - 1;
    const float* rowp = src + (size_t)gr * ld;
    float m = 0.f;
    for (int c4 = 4 * q; c4 < k; c4 += 4 * NW) m = fmaxf(m, amax4(*reinterpret_cast<const float4*>(rowp + c4)));
    m = NW == 8 ? max_8lanes(m) : max_4lanes(m);
    if (q == 0) rs[r] = pow2i(scale_exp(m));
}

// ---- aux GEMM with the activation fragments converted straight from global memory -----------------------------------------
struct RawX { float4 a0, a1, b0, b1; };               // 8 consecutive columns of row i (a) and row 32 + i (b)
__device__ __forceinline__ void load_raw(RawX& o, const float* p0, const float* p1, int c, int h, int k) {
    const int c0 = 16 * c + 8 * h;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    o.a0 = c0 < k ? *reinterpret_cast<const float4*>(p0 + c0) : z;
    o.a1 = c0 + 4 < k ? *reinterpret_cast<const float4*>(p0 + c0 + 4) : z;
    o.b0 = c0 < k ? *reinterpret_cast<const float4*>(p1 + c0) : z;
    o.b1 = c0 + 4 < k ? *reinterpret_cast<const float4*>(p1 + c0 + 4) : z;
}
__device__ __forceinline__ void split8(uint4& hp, uint4& lp, float4 a, float4 b, float inv) {
    split2h(a.x * inv, a.y * inv, hp.x, lp.x);
    split2h(a.z * inv, a.w * inv, hp.y, lp.y);
    split2h(b.x * inv, b.y * inv, hp.z, lp.z);
    split2h(b.z * inv, b.w * inv, hp.w, lp.w);
}
__device__ __forceinline__ void raw_to_x(XF& x, const RawX& r, float inv0, float inv1) {
    split8(x.xh0, x.xl0, r.a0, r.a1, inv0);
    split8(x.xh1, x.xl1, r.b0, r.b1, inv1);
}
__device__ __forceinline__ void gemm_aux_global(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const float* __restrict__ aux, int ld,
                                                int k, int n, const Ctx& c, float inv0, float inv1) {
    if (n <= 0) return;
    int g0 = c.row0 + c.i, g1 = c.row0 + 32 + c.i;
    g0 = g0 < c.n_rows ? g0 : c.n_rows - 1;
    g1 = g1 < c.n_rows ? g1 : c.n_rows - 1;
    const float* p0 = aux + (size_t)g0 * ld;
    const float* p1 = aux + (size_t)g1 * ld;
    WF wa, wb;
    RawX ra, rb;
    XF x;
    load_w(wa, wp, 0);
    load_raw(ra, p0, p1, 0, c.h, k);
    for (int s = 0; s < n; s += 2) {
        if (s + 1 < n) { load_w(wb, wp, s + 1); load_raw(rb, p0, p1, s + 1, c.h, k); }
        NERO_FENCE();
        raw_to_x(x, ra, inv0, inv1);
        ops_compute(aH, aL, wa, x);
        NERO_FENCE();
        if (s + 1 < n) {
            if (s + 2 < n) { load_w(wa, wp, s + 2); load_raw(ra, p0, p1, s + 2, c.h, k); }
            NERO_FENCE();
            raw_to_x(x, rb, inv0, inv1);
            ops_compute(aH, aL, wb, x);
            NERO_FENCE();
        }
    }
}

// aux part (its own unit) then main part of one feature tile; U = unit of the result per 32-row half
template <int NW>
__device__ __forceinline__ void gemm_tile(f32x16 (&aH)[2], f32x16 (&aL)[2], float (&U)[2], const Ctx& c, int t, const float* w_main,
                                          const float* w_aux, float wsc_main, float wsc_aux, int sm, int sx, const float* aux, int ld_aux,
                                          int k_aux_cols) {
    if (sx > 0) {
        const float wsc = wsc_aux;
        const float ra0 = c.S.rs_aux[c.i], ra1 = c.S.rs_aux[32 + c.i];
        gemm_aux_global(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w_aux) + HDR_BYTES) + (size_t)t * sx * 128 + c.lane,
                        aux, ld_aux, k_aux_cols, sx, c, 1.f / ra0, 1.f / ra1);
        U[0] = wsc * ra0;
        U[1] = wsc * ra1;
    }
    if (sm > 0) {
        const float wsc = wsc_main;
        const float u0 = wsc * c.S.rs_main[c.i], u1 = wsc * c.S.rs_main[32 + c.i];
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
        const uint4* wp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w_main) + HDR_BYTES) + (size_t)t * sm * 128 + c.lane;
        if (NW == 8) gemm_f16x3_lean(aH, aL, wp, c.S.actp + c.i * SA + 16 * c.h, 32 * SA, PLANE_A, sm);
        else gemm_f16x3(aH, aL, wp, c.S.actp + c.i * SA + 16 * c.h, 32 * SA, PLANE_A, sm);
    }
}

// VALU head on the current activation planes (8 threads per row; NW = 4: two passes of 32 rows)
template <int NW>
__device__ __forceinline__ void eval_head_p(const char* planes, const float* rs, const float* __restrict__ w, const float* __restrict__ b,
                                            float* __restrict__ out, int n_head, int hk, int row0, int tid) {
    const int q = tid & 7;
#pragma unroll
    for (int pass = 0; pass < 8 / NW; ++pass) {
        const int r = (tid >> 3) + 32 * pass;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c4 = 4 * q; c4 < hk; c4 += 32) {
            const float4 x = load_planes4h(planes + r * SA + c4 * 2, PLANE_A);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < n_head) {
                    const float4 ww = *reinterpret_cast<const float4*>(w + j * NERO_HID + c4);
                    s[j] = fmaf(x.x, ww.x, fmaf(x.y, ww.y, fmaf(x.z, ww.z, fmaf(x.w, ww.w, s[j]))));
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = sum_8lanes(s[j]);
        if (q == 0) {
            const float sc = rs[r];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < n_head) out[(size_t)(row0 + r) * 4 + j] = s[j] * sc + (b ? b[j] : 0.f);
        }
    }
}

// shared tail: barrier (row maxima of all 8 tiles visible, every wave done reading the input planes), rescale + store the two
// tiles of this wave as plane pairs, new row scales, barrier
template <int NW>
__device__ __forceinline__ void commit_planes_p(const Ctx& c, const float4 (&v0)[2][4], const float4 (&v1)[2][4], bool live0, bool live1) {
    __syncthreads();
    const int e0 = scale_exp(row_max8(c.S.rmax, c.i)), e1 = scale_exp(row_max8(c.S.rmax, 32 + c.i));
    const float inv0 = pow2i(-e0), inv1 = pow2i(-e1);
    if (live0) {
        char* dst = c.S.actp + c.i * SA + (32 * c.wave + 4 * c.h) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store_planes4h(dst + 16 * g, PLANE_A, scale4(v0[0][g], inv0));
            store_planes4h(dst + 32 * SA + 16 * g, PLANE_A, scale4(v0[1][g], inv1));
        }
    }
    if (NW == 4 && live1) {
        char* dst = c.S.actp + c.i * SA + (32 * (c.wave + NW) + 4 * c.h) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store_planes4h(dst + 16 * g, PLANE_A, scale4(v1[0][g], inv0));
            store_planes4h(dst + 32 * SA + 16 * g, PLANE_A, scale4(v1[1][g], inv1));
        }
    }
    if (c.wave == 0 && c.h == 0) { c.S.rs_main[c.i] = pow2i(e0); c.S.rs_main[32 + c.i] = pow2i(e1); }
    __syncthreads();
}

template <int NW>
__device__ __forceinline__ Ctx make_ctx(char* smem, int n_rows) {
    Ctx c;
    c.S = carve_p(smem);
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.i = c.lane & 31;
    c.h = c.lane >> 5;
    c.row0 = blockIdx.x * 64;
    c.n_rows = n_rows;
    c.scr = reinterpret_cast<float*>(c.S.scr + c.wave * (NW == 8 ? SCRP_BYTES / 2 : SCRP_BYTES));     // 8 / 16 rows at a time
    return c;
}

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
            m[r] = fmaxf(m[r], amax4(v));
        }
    }
}

template <int NW>
__device__ __forceinline__ void fwd_tile(const nero_fwd_chain& ch, const nero_fwd_layer& L, int l, const Ctx& c, int t, float4 (&val)[2][4],
                                         float (&m)[2] PH_PARAM) {
    m[0] = m[1] = 0.f;
    if (t < L.n_tiles) {
        f32x16 aH[2], aL[2];
        zero2(aH);
        zero2(aL);
        float4 bq[4];
        auto load_bias = [&]() {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bq[g] = L.bias ? *reinterpret_cast<const float4*>(L.bias + 32 * t + 8 * g + 4 * c.h) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if (NW == 4) load_bias();                       // (NW = 8: behind the GEMM -- 16 registers its 128 have no room for)
        float U[2] = {1.f, 1.f};
        PH(1);
        gemm_tile<NW>(aH, aL, U, c, t, L.w_main, L.w_aux, c.S.wsc[2 * l], c.S.wsc[2 * l + 1], L.k_main >> 4, L.k_aux >> 4, ch.aux, ch.ld_aux, ch.k_aux);
        if (NW == 8) { NERO_FENCE(); load_bias(); }
        PH(2);
        if (L.act == NERO_ACT_RELU) fwd_values<NERO_ACT_RELU>(aH, aL, bq, U, val, m);
        else if (L.act == NERO_ACT_SOFTPLUS100) fwd_values<NERO_ACT_SOFTPLUS100>(aH, aL, bq, U, val, m);
        else fwd_values<NERO_ACT_NONE>(aH, aL, bq, U, val, m);
        PH(3);
        if (L.save) {
            float* sblock = L.save + (size_t)c.row0 * NERO_HID + 32 * t;
            acc_to_global_rows<64 / NW>(c.scr, val[0], sblock, c.lane);
            acc_to_global_rows<64 / NW>(c.scr, val[1], sblock + (size_t)32 * NERO_HID, c.lane);
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
                const unsigned other = other_half(bits, c.h);
                if (c.h == 0) L.relu_mask[(size_t)(c.row0 + 32 * r + c.i) * 8 + t] = bits | (other << 16);
            }
        }
    }
    publish_rowmax(c.S.rmax, m[0], m[1], t, c.i, c.h);
    PH(4);
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void fwd_p_kernel(nero_fwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Ctx c = make_ctx<NW>(smem, n_rows);
    const int tid = threadIdx.x;
    PH_DECL;
#ifdef F16_PHASE_TIMING
    const long long ph_start = ph_t;
#endif
    WscRegs wr;
    wsc_request(wr, ch, [](const nero_fwd_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.k_main > 0 && Lx.n_tiles > 0 ? Lx.w_main : nullptr; pa = Lx.k_aux > 0 && Lx.n_tiles > 0 ? Lx.w_aux : nullptr; });
    if (ch.init) load_planes_scaled_p<NW>(c.S.actp, c.S.rs_main, ch.init, ch.ld_init, ch.k_init, c.row0, n_rows, tid);
    if (ch.aux) aux_row_scales_p<NW>(c.S.rs_aux, ch.aux, ch.ld_aux, ch.k_aux, c.row0, n_rows, tid);
    wsc_commit(c.S.wsc, wr, tid);
    __syncthreads();
    PH(0);
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_fwd_layer L = load_layer(ch, l);
        if (L.n_head > 0) eval_head_p<NW>(c.S.actp, c.S.rs_main, L.head_w, L.head_b, L.head_out, L.n_head, L.head_k, c.row0, tid);
        if (L.n_tiles == 0) continue;
        float4 v0[2][4], v1[2][4];
        float m0[2], m1[2];
        fwd_tile<NW>(ch, L, l, c, c.wave, v0, m0 PH_ARG);
        if (NW == 4) fwd_tile<NW>(ch, L, l, c, c.wave + NW, v1, m1 PH_ARG);
        commit_planes_p<NW>(c, v0, v1, c.wave < L.n_tiles, c.wave + NW < L.n_tiles);
        PH(5);
    }
#ifdef F16_PHASE_TIMING
    ph_acc[7] = clock64() - ph_start;                  // whole-workgroup residence (-> average workgroups in flight per CU)
    ph_acc[6] = 0;
#endif
    PH_END;
}

// ---------------------------------------------------------------------------------------------------------------------
// tangent chain (softplus networks):  adot_l = s_l * (W_l adot_{l-1}),  inj_l = gbar_l * beta (1-s_l) * zdot_l
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void tan_tile(const nero_tan_chain& ch, const nero_tan_layer& L, int l, const Ctx& c, int t, float4 (&val)[2][4],
                                         float (&m)[2]) {
    m[0] = m[1] = 0.f;
    if (t < L.n_tiles) {
        const size_t goff = (size_t)(c.row0 + c.i) * NERO_HID + 32 * t + 4 * c.h;     // + r*32*HID + 8g
        const size_t boff = (size_t)c.row0 * NERO_HID + 32 * t;
        const bool want_inj = L.inj != nullptr;        // (default: NULL -- the reverse kernel forms the injection, nero_bwd_layer.inj_adot)
        f32x16 aH[2], aL[2];
        zero2(aH);
        zero2(aL);
        float U[2] = {1.f, 1.f};
        gemm_tile<NW>(aH, aL, U, c, t, L.w_main, L.w_aux, c.S.wsc[2 * l], c.S.wsc[2 * l + 1], L.k_main >> 4, L.k_aux >> 4, ch.aux, ch.ld_aux, ch.k_aux);
        // the saved activations are requested BEHIND the GEMM (32 registers it has no room for at two workgroups per CU: 23 spilled);
        // the round trip is covered by the other workgroup's MFMAs, which is what this engine is for
        NERO_FENCE();
        float4 pa[2][4], pg[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pa[r][g] = *reinterpret_cast<const float4*>(L.a_saved + goff + (size_t)r * 32 * NERO_HID + 8 * g);
                pg[r][g] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        if (want_inj) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) pg[r][g] = *reinterpret_cast<const float4*>(L.gbar + goff + (size_t)r * 32 * NERO_HID + 8 * g);
        }
        float* scr = c.scr;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool live = (c.row0 + 32 * r + c.i) < c.n_rows;
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
            acc_to_global_rows<64 / NW>(scr, adq, L.adot + boff + (size_t)r * 32 * NERO_HID, c.lane);
            if (want_inj) acc_to_global_rows<64 / NW>(scr, ijq, L.inj + boff + (size_t)r * 32 * NERO_HID, c.lane);
        }
    }
    publish_rowmax(c.S.rmax, m[0], m[1], t, c.i, c.h);
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void tan_p_kernel(nero_tan_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Ctx c = make_ctx<NW>(smem, n_rows);
    const int tid = threadIdx.x;
    WscRegs wr;
    wsc_request(wr, ch, [](const nero_tan_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.k_main > 0 ? Lx.w_main : nullptr; pa = Lx.k_aux > 0 ? Lx.w_aux : nullptr; });
    if (ch.init) load_planes_scaled_p<NW>(c.S.actp, c.S.rs_main, ch.init, ch.ld_init, ch.k_init, c.row0, n_rows, tid);
    if (ch.aux) aux_row_scales_p<NW>(c.S.rs_aux, ch.aux, ch.ld_aux, ch.k_aux, c.row0, n_rows, tid);
    wsc_commit(c.S.wsc, wr, tid);
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_tan_layer L = load_layer(ch, l);
        float4 v0[2][4], v1[2][4];
        float m0[2], m1[2];
        tan_tile<NW>(ch, L, l, c, c.wave, v0, m0);
        if (NW == 4) tan_tile<NW>(ch, L, l, c, c.wave + NW, v1, m1);
        commit_planes_p<NW>(c, v0, v1, c.wave < L.n_tiles, c.wave + NW < L.n_tiles);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse chain:  delta_{l-1} = (delta_l W_l [+ dy_head W_head]) * act'(a_{l-1}) [+ inj_{l-1}]
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void combine_acc(float4 (&gq)[2][4], const f32x16 (&aH)[2], const f32x16 (&aL)[2], const float (&u)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            gq[r][g] = make_float4(ACCV(aH, aL, r, 4 * g) * u[r], ACCV(aH, aL, r, 4 * g + 1) * u[r],
                                   ACCV(aH, aL, r, 4 * g + 2) * u[r], ACCV(aH, aL, r, 4 * g + 3) * u[r]);
}

// one feature tile of one reverse layer; `first` = the chain's first dense layer (its input gradient goes to d_init / d_aux)
// MASKS (round 6): a chain whose reverse pass needs NO saved activation and NO injection -- every act_prev is ReLU with sign words
// (nero_bwd_layer.mask_prev) or the identity: the material / light predictors and the NeRF++ networks.  Compiled without the softplus path the
// tile keeps 2 sign words instead of 32 activation registers, and the kernel fits its 256 registers without the 14-16 spills of the generic
// bwd_p_kernel (which lost to the 512-thread kernel for exactly that reason).
template <bool PA_EARLY, int NW, bool MASKS>
__device__ __forceinline__ void bwd_tile(const nero_bwd_chain& ch, const nero_bwd_layer& L, int l, const Ctx& c, int t, bool first, float rs0, float rs1,
                                         float4 (&val)[2][4], float (&m)[2]) {
    m[0] = m[1] = 0.f;
    const bool live_t = t < L.k_main_tiles;
    const int fbase = 32 * t + 4 * c.h;
    const size_t goff = (size_t)(c.row0 + c.i) * NERO_HID + fbase;
    const size_t boff = (size_t)c.row0 * NERO_HID + 32 * t;
    const int steps = L.n_out >> 4;
    const bool has_inj = !MASKS && !first && L.inj != nullptr;
    const char* xp = c.S.actp + c.i * SA + 16 * c.h;
    // saved activations of this lane's outputs (ReLU: 1 / 0 from the sign words).  PA_EARLY: requested in FRONT of the main GEMM (32
    // registers through the k-loop, the HBM round trip under it); default: behind it (the round trip under the sibling workgroup's MFMAs)
    float4 pa[2][4];
    auto load_pa = [&]() {
        if (first || !live_t) return;
        if (MASKS && L.act_prev != NERO_ACT_RELU) return;   // (identity: the activation is not looked at)
        if (MASKS || (L.mask_prev && L.act_prev == NERO_ACT_RELU)) {   // 4 bytes per (row, tile) instead of 128: only the sign is needed
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned bits = L.mask_prev[(size_t)(c.row0 + 32 * r + c.i) * 8 + t] >> (16 * c.h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pa[r][g] = make_float4((bits >> (4 * g)) & 1u ? 1.f : 0.f, (bits >> (4 * g + 1)) & 1u ? 1.f : 0.f,
                                           (bits >> (4 * g + 2)) & 1u ? 1.f : 0.f, (bits >> (4 * g + 3)) & 1u ? 1.f : 0.f);
            }
        } else if (!MASKS) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) pa[r][g] = *reinterpret_cast<const float4*>(L.a_prev + goff + (size_t)r * 32 * NERO_HID + 8 * g);
        }
    };
    float4 gq[2][4];                                   // incoming gradient of this lane's outputs, true units
    if (L.n_out > 0) {
        f32x16 aH[2], aL[2];
        if (ch.d_aux && L.w_aux_t && t < L.k_aux_tiles) {
            zero2(aH);
            zero2(aL);
            const float wsc = c.S.wsc[2 * l + 1];
            if (NW == 8) gemm_f16x3_lean(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_aux_t) + HDR_BYTES) + (size_t)t * steps * 128 + c.lane, xp, 32 * SA, PLANE_A, steps);
            else gemm_f16x3(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_aux_t) + HDR_BYTES) + (size_t)t * steps * 128 + c.lane,
                       xp, 32 * SA, PLANE_A, steps);
            const float u[2] = {wsc * rs0, wsc * rs1};
            combine_acc(gq, aH, aL, u);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = fbase + 8 * g;
                    if (f < ch.ld_daux) *reinterpret_cast<float4*>(ch.d_aux + (size_t)(c.row0 + 32 * r + c.i) * ch.ld_daux + f) = gq[r][g];
                }
        }
        if (!live_t || (first && !ch.d_init)) { publish_rowmax(c.S.rmax, 0.f, 0.f, t, c.i, c.h); return; }
        if (PA_EARLY) {
            load_pa();
            NERO_FENCE();
        }
        zero2(aH);
        zero2(aL);
        const float wsc = c.S.wsc[2 * l];
        if (NW == 8) gemm_f16x3_lean(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_main_t) + HDR_BYTES) + (size_t)t * steps * 128 + c.lane, xp, 32 * SA, PLANE_A, steps);
        else gemm_f16x3(aH, aL, reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(L.w_main_t) + HDR_BYTES) + (size_t)t * steps * 128 + c.lane,
                   xp, 32 * SA, PLANE_A, steps);
        const float u[2] = {wsc * rs0, wsc * rs1};
        combine_acc(gq, aH, aL, u);
        if (first) {
            if (ch.d_init) {
                const int ldi = ch.ld_dinit;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = fbase + 8 * g;
                        if (f < ldi) {
                            float4 v = gq[r][g];
                            float4* dstp = reinterpret_cast<float4*>(ch.d_init + (size_t)(c.row0 + 32 * r + c.i) * ldi + f);
                            if (ch.accumulate_dinit) { const float4 o = *dstp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            *dstp = v;
                        }
                    }
            }
            return;                                    // (the caller leaves the layer loop)
        }
    } else {
        // head-only pseudo layer: the incoming gradient is the current content of the planes
        if (!live_t) { publish_rowmax(c.S.rmax, 0.f, 0.f, t, c.i, c.h); return; }
        if (PA_EARLY) load_pa();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gq[0][g] = scale4(load_planes4h(c.S.actp + c.i * SA + (fbase + 8 * g) * 2, PLANE_A), rs0);
            gq[1][g] = scale4(load_planes4h(c.S.actp + (32 + c.i) * SA + (fbase + 8 * g) * 2, PLANE_A), rs1);
        }
    }
    if (!PA_EARLY) {
        NERO_FENCE();
        load_pa();
    }
    float4 ijp[2][4];                                  // (unused: PRE = false)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) ijp[r][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.act_prev == NERO_ACT_RELU) bwd_values_h<NERO_ACT_RELU, false>(gq, pa, goff, has_inj, L, c.row0, c.i, fbase, c.n_rows, val, m, ijp);
    else if (!MASKS && L.act_prev == NERO_ACT_SOFTPLUS100) bwd_values_h<NERO_ACT_SOFTPLUS100, false>(gq, pa, goff, has_inj, L, c.row0, c.i, fbase, c.n_rows, val, m, ijp);
    else bwd_values_h<NERO_ACT_NONE, false>(gq, pa, goff, has_inj, L, c.row0, c.i, fbase, c.n_rows, val, m, ijp);
    if (L.delta_prev) {
        acc_to_global_rows<64 / NW>(c.scr, val[0], L.delta_prev + boff, c.lane);
        acc_to_global_rows<64 / NW>(c.scr, val[1], L.delta_prev + boff + (size_t)32 * NERO_HID, c.lane);
    }
    publish_rowmax(c.S.rmax, m[0], m[1], t, c.i, c.h);
}

template <int NW, bool MASKS>
__global__ __launch_bounds__(NW * 64, NW / 2) void bwd_p_kernel(nero_bwd_chain ch, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Ctx c = make_ctx<NW>(smem, n_rows);
    const int tid = threadIdx.x;
    WscRegs wr;
    wsc_request(wr, ch, [](const nero_bwd_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.n_out > 0 ? Lx.w_main_t : nullptr; pa = Lx.n_out > 0 ? Lx.w_aux_t : nullptr; });
    if (ch.dy) load_planes_scaled_p<NW>(c.S.actp, c.S.rs_main, ch.dy, ch.ld_dy, ch.k_dy, c.row0, n_rows, tid);
    else {
        for (int idx = tid; idx < 2 * PLANE_A / 16; idx += NW * 64) reinterpret_cast<uint4*>(c.S.actp)[idx] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 64) c.S.rs_main[tid] = 1.f;
    }
    wsc_commit(c.S.wsc, wr, tid);
    __syncthreads();
    for (int l = ch.n_layers - 1; l >= 0; --l) {
        const nero_bwd_layer L = load_layer(ch, l);
        const bool first = (L.a_prev == nullptr);
        if (first && ch.d_init == nullptr && !(ch.d_aux && L.w_aux_t)) break;
        if (first && L.n_out == 0) break;
        const float rs0 = c.S.rs_main[c.i], rs1 = c.S.rs_main[32 + c.i];
        float4 v0[2][4], v1[2][4];
        float m0[2], m1[2];
        bwd_tile<P_PA_EARLY_A, NW, MASKS>(ch, L, l, c, c.wave, first, rs0, rs1, v0, m0);
        if (NW == 4) bwd_tile<P_PA_EARLY_B, NW, MASKS>(ch, L, l, c, c.wave + NW, first, rs0, rs1, v1, m1);
        if (first) break;
        commit_planes_p<NW>(c, v0, v1, c.wave < L.k_main_tiles, c.wave + NW < L.k_main_tiles);
    }
}

}  // namespace

// ---- host side (dispatched from mlp_engine.hip) ------------------------------------------------------------------------------
#include <stdio.h>
#include <stdlib.h>
#ifdef F16_PHASE_TIMING
extern "C" int nero_debug_phases_p(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
    return 0;
}
#endif
static void report_occupancy(const void* f, const char* name) {      // NERO_DEBUG_OCC=1: resident workgroups per CU as the runtime sees it
    if (!getenv("NERO_DEBUG_OCC")) return;
    for (int lds = 32768; lds <= 81920; lds += 2048 * 3) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 256, lds);
        fprintf(stderr, "[nero] %s: dynamic LDS %d -> %d workgroups per CU (err %d)\n", name, lds, nb, (int)e);
    }
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 256, p_lds_bytes());
    fprintf(stderr, "[nero] %s: dynamic LDS %d -> %d workgroups per CU\n", name, p_lds_bytes(), nb);
}
// waves per workgroup of the paired kernels: NERO_F16_PW = 4 | 8, per pass as a decimal digit string "fwd tan bwd" (e.g. 848)
static int nero_pw(int kind) {
    static int sel[3] = {-1, -1, -1};
    if (sel[0] < 0) {
        const char* e = getenv("NERO_F16_PW");
        int v = e ? atoi(e) : NERO_F16_PW_DEFAULT;
        if (v < 10) v = v * 111;
        sel[0] = (v / 100) % 10 == 8 ? 8 : 4; sel[1] = (v / 10) % 10 == 8 ? 8 : 4; sel[2] = v % 10 == 8 ? 8 : 4;
    }
    return sel[kind];
}
template <class K, class CH> static void launch_p(K k4, K k8, int nw, const CH* ch, int n_rows, hipStream_t stream) {
    NERO_ONCE(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, p_lds_bytes()));
    NERO_ONCE(hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, p_lds_bytes()));
    nero_prof_mark_paired();
    if (nw == 8) hipLaunchKernelGGL(k8, dim3((n_rows + 63) / 64), dim3(512), p_lds_bytes(), stream, *ch, n_rows);
    else hipLaunchKernelGGL(k4, dim3((n_rows + 63) / 64), dim3(256), p_lds_bytes(), stream, *ch, n_rows);
}
int nero_f16p_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream) {
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3p): k_main / k_aux must be multiples of 16");
    NERO_ONCE(report_occupancy((const void*)fwd_p_kernel<4>, "fwd_p_kernel<4>"));
    launch_p(fwd_p_kernel<4>, fwd_p_kernel<8>, nero_pw(0), ch, n_rows, stream);
    return NERO_OK;
}

int nero_f16p_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream) {
    for (int l = 0; l < ch->n_layers; ++l)
        if ((ch->layer[l].k_main | ch->layer[l].k_aux) & 15)
            return nero_fail(NERO_ERR_ARG, "nero_mlp_tangent(f16x3p): k_main / k_aux must be multiples of 16");
    launch_p(tan_p_kernel<4>, tan_p_kernel<8>, nero_pw(1), ch, n_rows, stream);
    return NERO_OK;
}

// a chain whose reverse walk looks at no saved activation and adds no injection (bwd_tile<.., MASKS>)
bool nero_f16p_masks_only(const nero_bwd_chain* ch) {
    for (int l = 0; l < ch->n_layers; ++l) {
        const nero_bwd_layer& L = ch->layer[l];
        if (L.inj || L.inj_adot) return false;
        if (L.a_prev == nullptr) continue;                 // the first layer: no activation in front of it
        if (L.act_prev == NERO_ACT_NONE) continue;
        if (!(L.act_prev == NERO_ACT_RELU && L.mask_prev)) return false;
    }
    return true;
}
int nero_f16p_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream) {
    for (int l = 0; l < ch->n_layers; ++l)
        if (ch->layer[l].n_out & 15) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3p): n_out must be a multiple of 16");
    if (ch->d_aux && (ch->ld_daux & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3p): ld_daux must be a multiple of 4");
    if (ch->d_init && (ch->ld_dinit & 3)) return nero_fail(NERO_ERR_ARG, "nero_mlp_backward(f16x3p): ld_dinit must be a multiple of 4");
    if (nero_f16p_masks_only(ch)) launch_p(bwd_p_kernel<4, true>, bwd_p_kernel<8, true>, nero_pw(2), ch, n_rows, stream);
    else launch_p(bwd_p_kernel<4, false>, bwd_p_kernel<8, false>, nero_pw(2), ch, n_rows, stream);
    return NERO_OK;
}
