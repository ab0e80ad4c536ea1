// mlp_f16r.hip -- forward chain pass of the fp16 two-plane arithmetic (mlp_f16x3.hip) with the chain walk re-cut so that a WAVE OWNS ROWS
// (round 6).  Not a gemm_mode: an execution detail of NERO_GEMM_F16X3 for forward chains that save nothing (the sampler's and the occlusion
// march's SDF evaluations, forward-only inference), chosen inside nero_f16_forward (NERO_F16_ROWOWNER).  Same packed operand images, same
// per-row block scaling, same products in the same order: results BIT FOR BIT those of the 512-thread kernel (tests/test_rowowner_engine.py).
//
// Why (scripts/probe/rowowner_probe.hip, profiles/r06_rowowner_probe.txt).  In mlp_f16x3.hip eight waves share a 64-row tile and split the
// FEATURES of a layer: every layer ends in a row-maximum exchange through LDS and two workgroup barriers, the epilogue (bias, activation,
// maximum, plane conversion: ~450 VALU per wave) runs with an idle matrix pipe, every wave streams its own weight fragments L2 -> registers,
// and the activation planes travel through LDS (4 fragment reads per 6 MFMAs).  Here a 256-thread workgroup is four waves, ONE per SIMD with
// 512 registers each, and a wave owns 32 rows and ALL features of them:
//   * the activation planes (MFMA B operand) of its rows live in REGISTERS: 16 k-steps x (h, l) x 4 VGPRs.  The epilogue builds the next
//     layer's planes from the accumulator layout with one v_permlane32_swap per two registers.  No LDS round trip, no barrier between layers,
//     the row maximum is lane-local plus one exchange of the two lane halves;
//   * the packed weight image (A operand) streams L2 -> LDS by LDS-DMA into a ring of four 32 KB slots, one UNIT (one feature tile of one
//     layer part: <= 16 k-steps x 2 planes x 1 KB) per slot, each wave requesting a quarter of every unit: one copy per CU serves 128 rows
//     (512-thread kernel: one per 64 rows).  One s_barrier per unit keeps the four waves inside the ring window;
//   * the GEMM runs TILE-major (one accumulator, 48 MFMAs per 256-wide feature tile), and the epilogue arithmetic of tile t - 1 (one element
//     per k-step) is issued between the MFMAs of tile t.
// The block scale stays the EXACT row maximum (bit compatibility): the conversion of a layer's 128 values per lane into planes waits for the
// last tile and is exposed (~15 % of a layer in the probe; a scale known before the GEMM would hide it as well).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

#include "mlp_f16_util.h"

constexpr int R_RING = 4, R_SLOT = 32768, R_DMA = 8;           // slots, bytes per slot, LDS-DMA requests per wave and unit (fixed: vmcnt bookkeeping)
constexpr int R_MAX_UNITS = 2 * 8 * NERO_MAX_LAYERS;           // (aux + main) x feature tiles x layers
struct RUnit { unsigned long long src; int n_k, pad; };        // device address of the unit's first 1 KB chunk, its k-steps

struct LdsR {
    char* ring; RUnit* units; float* bias; float* wsc; int* n_units;
};
__device__ __forceinline__ LdsR carve_r(char* smem) {
    LdsR l;
    l.ring = smem;
    l.units = reinterpret_cast<RUnit*>(smem + R_RING * R_SLOT);
    l.bias = reinterpret_cast<float*>(l.units + R_MAX_UNITS);   // [NERO_MAX_LAYERS][256]
    l.wsc = l.bias + NERO_MAX_LAYERS * 256;
    l.n_units = reinterpret_cast<int*>(l.wsc + 32);
    return l;
}
inline int r_lds_bytes() { return R_RING * R_SLOT + R_MAX_UNITS * 16 + NERO_MAX_LAYERS * 256 * 4 + 32 * 4 + 16; }     // 144 016

struct RCtx {
    LdsR S; int lane, wave, i, hh; unsigned ring_addr;
    int q;                                              // running unit counter of this workgroup (ring position = q mod R_RING)
    int u_iss;                                          // index (in the unit list) of the unit the NEXT requests are for: (q + R_RING - 1) mod n_units
    int n_units, rows_pad;
};
struct RSrc { unsigned long long src; int nk; };        // a unit's entry, wave-uniform
__device__ __forceinline__ RSrc r_unit_entry(const RCtx& c, int u) {
    const RUnit e = c.S.units[u];
    RSrc r;
    r.src = __builtin_amdgcn_readfirstlane((unsigned)(e.src & 0xffffffffull)) |
            ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(e.src >> 32)) << 32);
    r.nk = __builtin_amdgcn_readfirstlane(e.n_k);
    return r;
}
// request chunk (8 wave + jj) of the unit `u` that will be consumed as unit number q (clamped to its last chunk: every wave issues
// exactly R_DMA requests per unit, so that "everything but the 2 R_DMA youngest requests has landed" is a constant s_waitcnt)
__device__ __forceinline__ void r_issue(const RCtx& c, const RSrc& u, int q, int jj) {
    int j = c.wave * R_DMA + jj;
    j = j < 2 * u.nk ? j : 2 * u.nk - 1;
    lds_dma16(reinterpret_cast<const char*>(u.src) + (size_t)j * 1024 + c.lane * 16, c.ring_addr + (q & (R_RING - 1)) * R_SLOT + j * 1024);
}

// rows [row0, row0 + 32) of this wave x the first k columns of a row-major fp32 matrix -> block-scaled plane fragments IN REGISTERS
// (lane (i, hh) holds columns 16 c + 8 hh + 0..7 of row i for every k-step c: the MFMA B layout) + the row's scale 2^e
template <int NK>
__device__ __forceinline__ float load_planes_regs(uint4 (&xh)[NK], uint4 (&xl)[NK], const float* __restrict__ src, int ld, int k, int row,
                                                  int n_rows, int hh) {
    int gr = row < n_rows ? row : n_rows - 1;
    const float* rowp = src + (size_t)gr * ld;
    float4 v[NK][2];
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NK; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c4 = 16 * c + 8 * hh + 4 * u;
            v[c][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < k) v[c][u] = *reinterpret_cast<const float4*>(rowp + c4);
            m = fmaxf(m, amax4(v[c][u]));
        }
    m = max_xor32(m);
    const int e = scale_exp(m);
    const float inv = pow2i(-e);
#pragma unroll
    for (int c = 0; c < NK; ++c) {
        split2h(v[c][0].x * inv, v[c][0].y * inv, xh[c].x, xl[c].x);
        split2h(v[c][0].z * inv, v[c][0].w * inv, xh[c].y, xl[c].y);
        split2h(v[c][1].x * inv, v[c][1].y * inv, xh[c].z, xl[c].z);
        split2h(v[c][1].z * inv, v[c][1].w * inv, xh[c].w, xl[c].w);
    }
    return pow2i(e);
}

// the 16 values of a finished feature tile (accumulator layout: v = 4 g + j <-> feature 32 t + 8 g + 4 hh + j of row i) -> the plane
// fragments of the next layer's k-steps 2 t, 2 t + 1.  Lane half hh' of k-step 2 t + gp needs features 16 gp + 8 hh' + 0..7 of the tile,
// i.e. quad g = 2 gp + hh' of BOTH lane halves: one v_permlane32_swap per register pair hands each half the other's quad.
__device__ __forceinline__ void tile_to_planes(const float (&v)[16], float inv, uint4& xh0, uint4& xl0, uint4& xh1, uint4& xl1) {
    unsigned hp[8], lp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split2h(v[2 * k] * inv, v[2 * k + 1] * inv, hp[k], lp[k]);
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(hp[4 * gp], hp[4 * gp + 2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hp[4 * gp + 1], hp[4 * gp + 3], false, false);
        const auto s2 = __builtin_amdgcn_permlane32_swap(lp[4 * gp], lp[4 * gp + 2], false, false);
        const auto s3 = __builtin_amdgcn_permlane32_swap(lp[4 * gp + 1], lp[4 * gp + 3], false, false);
        const uint4 H = make_uint4(s0[0], s1[0], s0[1], s1[1]), Lo = make_uint4(s2[0], s3[0], s2[1], s3[1]);
        if (gp == 0) { xh0 = H; xl0 = Lo; } else { xh1 = H; xl1 = Lo; }
    }
}

__device__ __forceinline__ float plane_elem(const uint4& h, const uint4& l, int j) {       // block-scaled value of element j (0..7) of a fragment
    const unsigned hw = j < 2 ? h.x : j < 4 ? h.y : j < 6 ? h.z : h.w, lw = j < 2 ? l.x : j < 4 ? l.y : j < 6 ? l.z : l.w;
    const f16x2 a = __builtin_bit_cast(f16x2, hw), b = __builtin_bit_cast(f16x2, lw);
    return (float)a[j & 1] + (float)b[j & 1];
}

// VALU head on the register planes: out[row][j] = b[j] + sum_k x[row][k] W[j][k], k < hk -- in the summation order of eval_head_f16
// (mlp_f16x3.hip: 8 threads per row, thread q sums the float4 groups c4 = 4 q + 32 m by nested fmaf, then a butterfly over q), so that the
// result is the same bit for bit.  Lane half hh holds the groups of q = 2 hh + u (even k-steps) and q = 4 + 2 hh + u (odd k-steps), u = 0, 1.
__device__ __forceinline__ void eval_head_regs(const uint4 (&xh)[16], const uint4 (&xl)[16], float rs, const float* __restrict__ w,
                                               const float* __restrict__ b, float* __restrict__ out, int n_head, int hk, int row, int hh,
                                               bool in_buffer) {
    float P[4][2][2];                                   // [head j][k-step parity][u]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) P[j][p][0] = P[j][p][1] = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c4 = 16 * c + 8 * hh + 4 * u;
            if (c4 < hk) {
                const float4 x = make_float4(plane_elem(xh[c], xl[c], 4 * u), plane_elem(xh[c], xl[c], 4 * u + 1), plane_elem(xh[c], xl[c], 4 * u + 2),
                                             plane_elem(xh[c], xl[c], 4 * u + 3));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 ww = *reinterpret_cast<const float4*>(w + (j < n_head ? j : 0) * NERO_HID + c4);
                    P[j][c & 1][u] = fmaf(x.x, ww.x, fmaf(x.y, ww.y, fmaf(x.z, ww.z, fmaf(x.w, ww.w, P[j][c & 1][u]))));
                }
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // butterfly of eval_head_f16 as seen by thread q = 0: ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))
        const float pa = P[j][0][0] + P[j][0][1], pb = P[j][1][0] + P[j][1][1];     // half 0: s0 + s1, s4 + s5;  half 1: s2 + s3, s6 + s7
        const auto ra = __builtin_amdgcn_permlane32_swap(__float_as_uint(pa), __float_as_uint(pa), false, false);
        const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(pb), __float_as_uint(pb), false, false);
        const float s = (__uint_as_float(ra[0]) + __uint_as_float(ra[1])) + (__uint_as_float(rb[0]) + __uint_as_float(rb[1]));
        if (hh == 0 && j < n_head && in_buffer) out[(size_t)row * 4 + j] = s * rs + (b ? b[j] : 0.f);      // (the caller's buffers are padded to 64 rows, a group is 128)
    }
}

// ---- one GEMM unit: n (<= NK) k-steps of one feature tile.  `epi(c)` runs once per iteration c = 0..15 (the caller's epilogue of the
// previous tile, one element per k-step), whether or not the unit has a k-step c.
// FULL: the unit has exactly NK k-steps (every 256-wide layer: NK = 16) -- no run-time guard inside the stream: with `k < n` tests every k-step
// is a basic block of its own, and hipcc drains lgkmcnt at every join, which serialises the fragment reads two steps ahead.
template <int NK, bool FULL, class EPI>
__device__ __forceinline__ void r_unit(RCtx& c, f32x16& acc, const uint4 (&bh)[NK], const uint4 (&bl)[NK], int n_rt, EPI&& epi) {
    const int n = FULL ? NK : n_rt;
    const int q = c.q;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // this wave's quarter of unit q has landed (2 R_DMA younger requests may be in flight)
    __builtin_amdgcn_s_barrier();                          // ... everybody's has, and everybody is done reading the slot the next requests overwrite
    const char* slot = c.S.ring + (q & (R_RING - 1)) * R_SLOT + c.lane * 16;
    const RSrc nxt = r_unit_entry(c, c.u_iss);
    uint4 w[3][2];
    w[0][0] = *reinterpret_cast<const uint4*>(slot);
    w[0][1] = *reinterpret_cast<const uint4*>(slot + 1024);
    if (FULL ? NK > 1 : n > 1) { w[1][0] = *reinterpret_cast<const uint4*>(slot + 2048); w[1][1] = *reinterpret_cast<const uint4*>(slot + 3072); }
    NERO_FENCE();
    static_for<0, 16>([&](auto cc) {
        constexpr int k = cc.value;
        if (k < NK && (FULL || k < n)) {
            if (FULL ? k + 2 < NK : k + 2 < n) {
                w[(k + 2) % 3][0] = *reinterpret_cast<const uint4*>(slot + (2 * k + 4) * 1024);
                w[(k + 2) % 3][1] = *reinterpret_cast<const uint4*>(slot + (2 * k + 5) * 1024);
            }
        }
        if ((k & 1) == 0) r_issue(c, nxt, q + R_RING - 1, k >> 1);     // the requests of unit q + 3, one per two iterations
        if (k < NK && (FULL || k < n)) {
            constexpr int kk = k < NK ? k : 0;
            NERO_MFH(acc, w[k % 3][1], bh[kk]);                            // wl xh, wh xl, wh xh: the order of ops_compute (mlp_f16_util.h)
            NERO_MFH(acc, w[k % 3][0], bl[kk]);
            NERO_MFH(acc, w[k % 3][0], bh[kk]);
        }
        epi(cc);
        NERO_FENCE();
    });
    c.q = q + 1;
    c.u_iss = c.u_iss + 1 < c.n_units ? c.u_iss + 1 : 0;
}

template <int ACT, int NAUX>
__device__ __forceinline__ void r_layer(RCtx& c, const nero_fwd_layer& L, int l, uint4 (&xh)[16], uint4 (&xl)[16], float& rs_main,
                                        const uint4 (&ah)[NAUX], const uint4 (&al)[NAUX], float rs_aux) {
    const int sm = L.k_main >> 4, sx = L.k_aux >> 4, nt = L.n_tiles;
    const float u_aux = c.S.wsc[2 * l + 1] * rs_aux, u_main = c.S.wsc[2 * l] * rs_main;
    const float U = sm > 0 ? u_main : u_aux;
    const float conv = sm > 0 ? u_aux / u_main : 1.f;      // aux partial sums -> the main part's unit (exact: powers of two)
    const float* bias = c.S.bias + l * 256 + 4 * c.hh;
    float vals[8][16];
    float m = 0.f;
    float4 bq_prev[4], bq_cur[4];
    static_for<0, 8>([&](auto tt) {
        constexpr int t = tt.value;
        if (t < nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bq_cur[g] = L.bias ? *reinterpret_cast<const float4*>(bias + 32 * t + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            auto epi = [&](auto ee) {
                constexpr int e = ee.value;
                if (t > 0) {
                    const float4 bb = bq_prev[e >> 2];
                    const float bv = (e & 3) == 0 ? bb.x : (e & 3) == 1 ? bb.y : (e & 3) == 2 ? bb.z : bb.w;
                    const float y = act_fwd<ACT>(fmaf(vals[t > 0 ? t - 1 : 0][e], U, bv));
                    vals[t > 0 ? t - 1 : 0][e] = y;
                    m = fmaxf(m, fabsf(y));
                }
            };
            if (sx > 0) {
                r_unit<NAUX, false>(c, acc, ah, al, sx, [](auto) {});
                if (sm > 0) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[v] *= conv;
                }
            }
            if (sm == 16) r_unit<16, true>(c, acc, xh, xl, 16, epi);
            else if (sm > 0) r_unit<16, false>(c, acc, xh, xl, sm, epi);
            else static_for<0, 16>(epi);
#pragma unroll
            for (int v = 0; v < 16; ++v) vals[t][v] = acc[v];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq_prev[g] = bq_cur[g];
        }
    });
    // the last tile's epilogue, the row maximum, the conversion of all tiles into the next layer's planes
    static_for<0, 8>([&](auto tt) {
        constexpr int t = tt.value;
        if (t == nt - 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float4 bb = bq_prev[e >> 2];
                const float bv = (e & 3) == 0 ? bb.x : (e & 3) == 1 ? bb.y : (e & 3) == 2 ? bb.z : bb.w;
                const float y = act_fwd<ACT>(fmaf(vals[t][e], U, bv));
                vals[t][e] = y;
                m = fmaxf(m, fabsf(y));
            }
        }
    });
    m = max_xor32(m);
    const int e = scale_exp(m);
    const float inv = pow2i(-e);
    static_for<0, 8>([&](auto tt) {
        constexpr int t = tt.value;
        if (t < nt) tile_to_planes(vals[t], inv, xh[2 * t], xl[2 * t], xh[2 * t + 1], xl[2 * t + 1]);
    });
    rs_main = pow2i(e);
}

template <int NAUX>
__global__ __launch_bounds__(256, 1) void fwd_r_kernel(nero_fwd_chain ch, int n_rows, int n_groups, int rows_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RCtx c;
    c.S = carve_r(smem);
    const int tid = threadIdx.x;
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.i = c.lane & 31;
    c.hh = c.lane >> 5;
    c.ring_addr = __builtin_amdgcn_readfirstlane(lds_offset_of(c.S.ring));
    c.q = 0;
    c.rows_pad = rows_pad;
    // ---- once per workgroup: the unit list (consumption order), every layer's bias vector, the packed images' scales -> LDS
    if (tid == 0) {
        int nu = 0;
        for (int l = 0; l < ch.n_layers; ++l) {
            const nero_fwd_layer& L = ch.layer[l];
            const int sm = L.k_main >> 4, sx = L.k_aux >> 4;
            for (int t = 0; t < L.n_tiles; ++t) {
                if (sx > 0) { c.S.units[nu].src = (unsigned long long)(reinterpret_cast<const char*>(L.w_aux) + HDR_BYTES + (size_t)t * sx * 2048); c.S.units[nu].n_k = sx; ++nu; }
                if (sm > 0) { c.S.units[nu].src = (unsigned long long)(reinterpret_cast<const char*>(L.w_main) + HDR_BYTES + (size_t)t * sm * 2048); c.S.units[nu].n_k = sm; ++nu; }
            }
        }
        *c.S.n_units = nu;
    }
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_fwd_layer& L = ch.layer[l];
        if (L.bias && tid < 32 * L.n_tiles) c.S.bias[l * 256 + tid] = L.bias[tid];
    }
    {
        WscRegs wr;
        wsc_request(wr, ch, [](const nero_fwd_layer& Lx, const float*& pm, const float*& pa) { pm = Lx.k_main > 0 && Lx.n_tiles > 0 ? Lx.w_main : nullptr; pa = Lx.k_aux > 0 && Lx.n_tiles > 0 ? Lx.w_aux : nullptr; });
        wsc_commit(c.S.wsc, wr, tid);
    }
    __syncthreads();
    c.n_units = __builtin_amdgcn_readfirstlane(*c.S.n_units);
    c.u_iss = 0;
    for (int q = 0; q < R_RING - 1; ++q) {                 // the first R_RING - 1 units of the (cyclic) list
        const RSrc u0 = r_unit_entry(c, c.u_iss);
#pragma unroll
        for (int jj = 0; jj < R_DMA; ++jj) r_issue(c, u0, q, jj);
        c.u_iss = c.u_iss + 1 < c.n_units ? c.u_iss + 1 : 0;
    }
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int row = grp * 128 + 32 * c.wave + c.i;
        uint4 xh[16], xl[16], ah[NAUX], al[NAUX];
        float rs_main = 1.f, rs_aux = 1.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) xh[k] = xl[k] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < NAUX; ++k) ah[k] = al[k] = make_uint4(0u, 0u, 0u, 0u);
        if (ch.init) rs_main = load_planes_regs<16>(xh, xl, ch.init, ch.ld_init, ch.k_init, row, n_rows, c.hh);
        if (ch.aux) rs_aux = load_planes_regs<NAUX>(ah, al, ch.aux, ch.ld_aux, ch.k_aux, row, n_rows, c.hh);
        for (int l = 0; l < ch.n_layers; ++l) {
            const nero_fwd_layer L = load_layer(ch, l);
            if (L.n_head > 0) eval_head_regs(xh, xl, rs_main, L.head_w, L.head_b, L.head_out, L.n_head, L.head_k, row, c.hh, row < rows_pad);
            if (L.n_tiles == 0) continue;
            if (L.act == NERO_ACT_RELU) r_layer<NERO_ACT_RELU, NAUX>(c, L, l, xh, xl, rs_main, ah, al, rs_aux);
            else if (L.act == NERO_ACT_SOFTPLUS100) r_layer<NERO_ACT_SOFTPLUS100, NAUX>(c, L, l, xh, xl, rs_main, ah, al, rs_aux);
            else r_layer<NERO_ACT_NONE, NAUX>(c, L, l, xh, xl, rs_main, ah, al, rs_aux);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the ring runs R_RING - 1 units past the last group: let the requests land)
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------------
static int r_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}
// can this chain run on the row-owner kernel?  Forward chains that save nothing (no saved activations, no ReLU masks), with at least one
// GEMM layer, operand widths in whole k-steps, heads that read whole float4 groups, head rows inside the caller's padded buffers.
bool nero_f16r_covers(const nero_fwd_chain* ch) {
    int gemm = 0;
    for (int l = 0; l < ch->n_layers; ++l) {
        const nero_fwd_layer& L = ch->layer[l];
        if (L.save || L.relu_mask) return false;
        if ((L.k_main | L.k_aux) & 15) return false;
        if (L.n_tiles > 8 || (L.k_main >> 4) > 16 || (L.k_aux >> 4) > 6) return false;
        if (L.n_head > 0 && (L.head_k & 3)) return false;
        if (L.n_tiles > 0) { if (L.k_main == 0 && L.k_aux == 0) return false; ++gemm; }
    }
    return gemm > 0 && (!ch->init || ch->k_init <= 256) && (!ch->aux || ch->k_aux <= 96);
}
int nero_f16r_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream) {
    const int n_groups = (n_rows + 127) / 128;
    const int grid = n_groups < r_cu_count() ? n_groups : r_cu_count();
    if (ch->aux && ch->k_aux > 48) {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_r_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, r_lds_bytes()));
        hipLaunchKernelGGL(fwd_r_kernel<6>, dim3(grid), dim3(256), r_lds_bytes(), stream, *ch, n_rows, n_groups, (n_rows + 63) / 64 * 64);
    } else {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_r_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, r_lds_bytes()));
        hipLaunchKernelGGL(fwd_r_kernel<3>, dim3(grid), dim3(256), r_lds_bytes(), stream, *ch, n_rows, n_groups, (n_rows + 63) / 64 * 64);
    }
    return NERO_OK;
}
