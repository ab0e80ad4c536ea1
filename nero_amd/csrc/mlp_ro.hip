// mlp_ro.hip -- the fused MLP-chain passes, "row-owner" organisation (gemm_mode NERO_GEMM_F16X3R).
//
// Same arithmetic as mlp_f16x3.hip (two block-scaled fp16 planes per fp32 operand, three v_mfma_f32_32x32x16_f16 plane
// products per k-step in two accumulator sets, per-row activation scales, per-matrix weight scales: fp32-grade), same packed
// weight images (nero_pack_batch kind 3) and the same row-major fp32 saved tensors -- but a different mapping onto the CU:
//
//   * one WAVE owns 32 batch rows for the WHOLE chain and keeps the MFMA B operand of the current layer -- the fp16 plane
//     fragments of its 32 rows x <= 256 input features, 128 registers per lane -- in its register file for all feature tiles of
//     the layer.  The layer's fp32 outputs go to a wave-private, XOR-swizzled LDS image (32 rows x 256 x 4 B) and come back as
//     the next layer's fragments; no other wave ever touches them: no activation barrier, no 8-fold re-read of activation planes.
//   * a workgroup = 4 such waves (one per SIMD) = 128 rows.  What the waves share is the WEIGHTS: the packed image of the
//     current layer is streamed ONCE per workgroup from L2 into a two-slot LDS ring by LDS-DMA (global_load_lds_dwordx4, 8
//     k-steps = 16 KiB per slot) and read by all four waves with conflict-free ds_read_b128: half the L2 -> CU weight traffic
//     per row of the 64-row-tile engines.  One s_barrier per 8 k-steps (768 MFMA cycles) hands the next chunk over.
//   * the VALU epilogue of feature tile t (bias, activation, row maximum) is software-pipelined into the k-loop of tile t+1:
//     one accumulator element per k-step, in the issue shadow of that step's three MFMAs; two accumulator sets alternate.
//
// LDS: 4 x 32 KiB outputs + 2 x 16 KiB weight ring = 160 KiB = the whole CU (one workgroup per CU by construction: a wave uses
// ~300 registers).  Layer walk (tile index T within a layer: dense tiles 0..nt-1, then the optional head as tile nt -- heads run
// on the matrix pipe too, from a packed 32-row image):
//     body(T):  save(T-2)  { k-loop of tile T  ||  epilogue elements of tile T-1 }
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

#include "mlp_f16_util.h"

constexpr int RO_ROWS = 128;                       // rows per workgroup: 4 waves x 32
constexpr int STEP_BYTES = 2048;                   // one k-step (16 k) of one 32-feature tile: two planes x 64 lanes x 16 B
constexpr int CHUNK_STEPS = 8;
constexpr int SLOT_BYTES = CHUNK_STEPS * STEP_BYTES;            // 16 KiB
constexpr int Y_WAVE_BYTES = 32 * 1024;                         // 32 rows x 256 fp32
constexpr int RING_OFF = 4 * Y_WAVE_BYTES;                      // 128 KiB
constexpr int RO_LDS_BYTES = RING_OFF + 2 * SLOT_BYTES;         // 160 KiB

// The chain descriptors are read straight from the kernarg segment through constant-address-space references (scalar loads into
// SGPRs).  Left to the optimizer, a by-value struct argument of this size that is indexed dynamically from several inlined
// helpers gets copied to scratch, and every descriptor field then costs a scratch load and every pointer a flat access.
#define NERO_CONST __attribute__((address_space(4)))
#define NERO_GLOBAL __attribute__((address_space(1)))
typedef const NERO_CONST nero_fwd_chain CFwdChain;
typedef const NERO_CONST nero_fwd_layer CFwdLayer;
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p) {                 // explicit global_load_dwordx4 (never a flat access)
    const v4f v = *(const NERO_GLOBAL v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4(float* p, const float4& x) {
    const v4f v = {x.x, x.y, x.z, x.w};
    *(NERO_GLOBAL v4f*)p = v;
}
__device__ __forceinline__ float g1(const float* p) { return *(const NERO_GLOBAL float*)p; }

struct Frag { uint4 h, l; };

#define NERO_MFR(ACC, A, B) \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)

__device__ __forceinline__ void lds_frag(Frag& f, const char* lane_slot, int step) {
    f.h = *reinterpret_cast<const uint4*>(lane_slot + step * STEP_BYTES);
    f.l = *reinterpret_cast<const uint4*>(lane_slot + step * STEP_BYTES + 1024);
}
// H += wh xh ;  L += wl xh + wh xl      (the dropped wl xl term is <= 2^-24 of the product)
__device__ __forceinline__ void step3(f32x16& H, f32x16& L, const Frag& w, const uint4& xh, const uint4& xl) {
#ifdef RO_NO_MFMA                                    // (timing experiments: scripts/ro_variants.sh)
    H[0] += __uint_as_float(w.h.x ^ xh.x); L[0] += __uint_as_float(w.l.x ^ xl.x);
    return;
#endif
    NERO_MFR(L, w.l, xh);
    NERO_MFR(H, w.h, xh);
    NERO_MFR(L, w.h, xl);
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int v = 0; v < 16; ++v) a[v] = 0.f;
}
__device__ __forceinline__ const char* img_body(const float* img) { return reinterpret_cast<const char*>(img) + HDR_BYTES; }
__device__ __forceinline__ float partner_max(float m) { return fmaxf(m, __shfl_xor(m, 32)); }

// ---- weight stream ------------------------------------------------------------------------------------------------------------
// A chunk = up to 8 consecutive k-steps of one tile's slab inside a packed image (contiguous there).  Every wave copies its
// quarter: 1-KiB pieces q = wave, wave + 4, ... (LDS destination = wave-uniform base + lane*16).
struct Chunk { const char* src; int steps; };
__device__ __forceinline__ void issue_chunk(const Chunk& ck, char* slot, int wave, int lane) {
    const int n = ck.steps * 2;
    for (int q = wave; q < n; q += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ck.src + q * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
}

// ---- wave-private fp32 output image in LDS -----------------------------------------------------------------------------------
// row i at i*1024 B; 16-byte chunk q (columns 4q..4q+3) is stored at chunk q ^ (i & 15): conflict-free for the accumulator-layout
// writes (8 consecutive lanes = 8 rows, one chunk column) and for the fragment-order reads (16 rows, one chunk column).
__device__ __forceinline__ char* y_chunk(char* ybase, int i, int q) { return ybase + i * 1024 + ((q ^ (i & 15)) << 4); }

// the k-step c fragment of this lane: columns 16c + 8h + (0..7) of row i, scaled by inv, as fp16 plane pairs
__device__ __forceinline__ void y_to_frag(char* ybase, int i, int h, int c, float inv, uint4& xh, uint4& xl) {
    const float4 a = *reinterpret_cast<const float4*>(y_chunk(ybase, i, 4 * c + 2 * h));
    const float4 b = *reinterpret_cast<const float4*>(y_chunk(ybase, i, 4 * c + 2 * h + 1));
    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
    split2h(a.x * inv, a.y * inv, h0, l0);
    split2h(a.z * inv, a.w * inv, h1, l1);
    split2h(b.x * inv, b.y * inv, h2, l2);
    split2h(b.z * inv, b.w * inv, h3, l3);
    xh = make_uint4(h0, h1, h2, h3);
    xl = make_uint4(l0, l1, l2, l3);
}
// columns [32T, 32T+32) of row `grow` of a row-major fp32 matrix -> LDS image (accumulator-layout pieces: 8g + 4h); returns max |.|
__device__ __forceinline__ float load_tile_to_y(char* ybase, const float* __restrict__ src, int ld, int k, int grow, int T, int i, int h) {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 32 * T + 8 * g + 4 * h;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < k) v = ldg4(src + (size_t)grow * ld + c);
        *reinterpret_cast<float4*>(y_chunk(ybase, i, c >> 2)) = v;
        m = fmaxf(m, amax4(v));
    }
    return m;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward chain
// ---------------------------------------------------------------------------------------------------------------------------
// Chunk sequence of ONE layer: for every tile j (dense tiles 0..nt-1, then the head as tile nt): [aux chunk] [main k-steps 0..7]
// [main k-steps 8..15].  The prefetcher runs exactly one chunk ahead of the consumer, so it only ever needs the current layer's
// schedule (a handful of SGPRs) plus the FIRST chunk of the next layer, fetched from the kernarg segment once per layer.
struct LayerSched {
    const char *aux, *main, *head;     // image bodies
    int nt, ntt, sa, sm, sh;           // dense tiles, tiles incl. head, k-steps of the aux / main / head slabs
};
template <class LT>
__device__ __forceinline__ void fwd_tile_steps(const LT& L, int j, int& sa, int& sm) {
    if (j < L.n_tiles) { sa = L.k_aux >> 4; sm = L.k_main >> 4; }
    else { sa = 0; sm = L.head_k >> 4; }
}
template <class LT>
__device__ __forceinline__ LayerSched make_sched(const LT& L) {
    LayerSched s;
    s.nt = L.n_tiles;
    s.ntt = L.n_tiles + (L.n_head > 0 ? 1 : 0);
    s.sa = L.k_aux >> 4; s.sm = L.k_main >> 4; s.sh = L.n_head > 0 ? (L.head_k >> 4) : 0;
    s.aux = s.sa ? img_body(L.w_aux) : nullptr;
    s.main = s.sm ? img_body(L.w_main) : nullptr;
    s.head = s.sh ? img_body(L.head_w) : nullptr;
    return s;
}
__device__ __forceinline__ bool sched_exists(const LayerSched& s, int j, int part) {
    const int sa = j < s.nt ? s.sa : 0, sm = j < s.nt ? s.sm : s.sh;
    return part == 0 ? sa > 0 : (part == 1 ? sm > 0 : sm > CHUNK_STEPS);
}
// first existing chunk at or after (j, part); false when the layer is exhausted
__device__ __forceinline__ bool sched_settle(const LayerSched& s, int& j, int& part) {
    while (j < s.ntt) {
        if (part > 2) { part = 0; ++j; continue; }
        if (sched_exists(s, j, part)) return true;
        ++part;
    }
    return false;
}
__device__ __forceinline__ Chunk sched_chunk(const LayerSched& s, int j, int part) {
    Chunk ck;
    if (part == 0) { ck.src = s.aux + (size_t)j * s.sa * STEP_BYTES; ck.steps = s.sa; return ck; }
    const int sm = j < s.nt ? s.sm : s.sh;
    const char* base = j < s.nt ? s.main + (size_t)j * s.sm * STEP_BYTES : s.head;
    if (part == 1) { ck.src = base; ck.steps = sm < CHUNK_STEPS ? sm : CHUNK_STEPS; }
    else { ck.src = base + CHUNK_STEPS * STEP_BYTES; ck.steps = sm - CHUNK_STEPS; }
    return ck;
}
__device__ __forceinline__ nero_fwd_layer fwd_layer_copy(CFwdLayer& s) {
    nero_fwd_layer L;
    L.w_main = s.w_main; L.w_aux = s.w_aux; L.bias = s.bias; L.save = s.save; L.head_w = s.head_w; L.head_b = s.head_b;
    L.head_out = s.head_out; L.k_main = s.k_main; L.k_aux = s.k_aux; L.n_tiles = s.n_tiles; L.n_head = s.n_head; L.act = s.act;
    L.head_k = s.head_k; L.relu_mask = nullptr;
    return L;
}

struct FwdShared {                 // wave-uniform stream state
    LayerSched sched;              // schedule of the layer being consumed
    int pj, ppart;                 // position of the chunk in `pre`
    Chunk pre;                     // next chunk to issue (steps == 0: nothing left)
    Chunk next_first;              // first chunk of the next non-empty layer (steps == 0: none)
    bool crossed;                  // `pre` already belongs to the next layer
    int n_chunk;                   // chunks consumed so far (slot parity)
};
// first chunk of the first layer with tiles at or after l0 (steps == 0 if none)
__device__ __forceinline__ Chunk first_chunk_from(CFwdChain& ch, int l0) {
    Chunk ck;
    ck.src = nullptr; ck.steps = 0;
    for (int l = l0; l < ch.n_layers; ++l) {
        CFwdLayer& L = ch.layer[l];
        if (L.n_tiles + (L.n_head > 0 ? 1 : 0) == 0) continue;
        const LayerSched s = make_sched(L);
        int j = 0, part = 0;
        if (sched_settle(s, j, part)) ck = sched_chunk(s, j, part);
        break;
    }
    return ck;
}
// the consumer enters layer l (its first chunk has been issued already): point the prefetcher at the layer's second chunk
__device__ __forceinline__ void stream_enter_layer(CFwdChain& ch, FwdShared& S, const LayerSched& sched, int l) {
    S.sched = sched;
    S.next_first = first_chunk_from(ch, l + 1);
    S.crossed = false;
    S.pj = 0; S.ppart = 0;
    sched_settle(sched, S.pj, S.ppart);          // = the first chunk (already in flight)
    ++S.ppart;
    if (sched_settle(sched, S.pj, S.ppart)) S.pre = sched_chunk(sched, S.pj, S.ppart);
    else { S.pre = S.next_first; S.crossed = true; }
}
// chunk hand-over: my pieces of the chunk about to be consumed have landed -> barrier (all pieces present, previous slot free)
// -> start the DMA of the chunk after it into the slot just released.  Returns this lane's read pointer into the current slot.
__device__ __forceinline__ const char* chunk_begin(FwdShared& S, char* smem, int wave, int lane) {
#ifndef RO_NO_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    const int cur = S.n_chunk & 1;
    if (S.pre.steps > 0) {
#ifndef RO_NO_DMA
        issue_chunk(S.pre, smem + RING_OFF + (cur ^ 1) * SLOT_BYTES, wave, lane);
#endif
        if (S.crossed) S.pre.steps = 0;           // one chunk into the next layer is as far as the stream runs ahead
        else {
            ++S.ppart;
            if (sched_settle(S.sched, S.pj, S.ppart)) S.pre = sched_chunk(S.sched, S.pj, S.ppart);
            else { S.pre = S.next_first; S.crossed = true; }
        }
    }
    ++S.n_chunk;
    return smem + RING_OFF + cur * SLOT_BYTES + lane * 16;
}

// ---- epilogue element ------------------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act_hw(float z) {
    if (ACT == NERO_ACT_SOFTPLUS100) return softplus100(z);
    if (ACT == NERO_ACT_RELU) return fmaxf(z, 0.f);
    return z;
}
__device__ __forceinline__ float f4_get(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
__device__ __forceinline__ void f4_set(float4& v, int k, float x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }

struct FwdEpi {                    // per-tile epilogue state of one lane
    float4 bq[4];                  // bias of the 16 accumulator elements
    char* yaddr[4];                // LDS addresses of the four 16-byte pieces of this lane in the output image
    float4 yq;
};
// one accumulator element of the previous tile: bias, activation (wave-uniform switch), row maximum, 16-byte LDS store per 4
__device__ __forceinline__ void fwd_epi_elem(int c, int act, const f32x16& Hp, const f32x16& Lp, float U, FwdEpi& E, float& m_run) {
#ifdef RO_NO_EPI
    if (c == 0) m_run = fmaxf(m_run, Hp[0] + Lp[0]);
    return;
#endif
    const float z = fmaf(fmaf(Lp[c], LO_INV, Hp[c]), U, f4_get(E.bq[c >> 2], c & 3));
    float y;
    if (act == NERO_ACT_SOFTPLUS100) y = softplus100(z);
    else y = act == NERO_ACT_RELU ? fmaxf(z, 0.f) : z;
    m_run = fmaxf(m_run, fabsf(y));
    f4_set(E.yq, c & 3, y);
    if ((c & 3) == 3) *reinterpret_cast<float4*>(E.yaddr[c >> 2]) = E.yq;
}

// One tile body, ONE static instance per accumulator parity (all switches are wave-uniform branches: several inlined variants of
// this body made the register allocator duplicate the accumulator tuples at their merge points and spill).
//   live : run the k-loop of tile T into accumulator set P (aux block + whole 8-step chunks: the operand fragments beyond the
//          layer's true K are zero)
//   epi  : weave the epilogue of tile T-1 (accumulator set 1-P) into that loop, one element per k-step
template <int P, int NA>
__device__ __forceinline__ void fwd_tile(const nero_fwd_layer& L, FwdShared& S, char* smem, char* ybase, int T, int nt, int ntt,
                                         f32x16 (&acc)[2][2], const uint4 (&XH)[16], const uint4 (&XL)[16], const uint4 (&AH)[NA], const uint4 (&AL)[NA],
                                         float U, float Uh, float ratio, float& m_run,
                                         int wave, int lane, int i, int h, int row, bool alive) {
    const bool live = T < ntt;                       // wave-uniform
    const bool epi = T >= 1 && (T - 1) < nt;
    const bool epi_head = T >= 1 && (T - 1) == nt && ntt > nt;
    f32x16& H = acc[P][0];
    f32x16& Lo = acc[P][1];
    const f32x16& Hp = acc[1 - P][0];
    const f32x16& Lp = acc[1 - P][1];
    // save of tile T-2 (its epilogue completed in the previous body), from the LDS image: a whole chunk ahead of the next vmcnt(0)
    if (T >= 2 && (T - 2) < nt && L.save && alive) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * (T - 2) + 8 * g + 4 * h;
            stg4(L.save + (size_t)row * NERO_HID + col, *reinterpret_cast<const float4*>(y_chunk(ybase, i, col >> 2)));
        }
    }
    int sa = 0, sm = 0;
    if (live) fwd_tile_steps(L, T, sa, sm);
    const int act = L.act;
    FwdEpi E;
    if (epi) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            E.bq[g] = ldg4(L.bias + 32 * (T - 1) + 8 * g + 4 * h);
            E.yaddr[g] = y_chunk(ybase, i, (32 * (T - 1) + 8 * g + 4 * h) >> 2);
        }
        E.yq = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (live) {
        zero16(H);
        zero16(Lo);
        if (sa > 0) {                                // aux part (skip-connection columns), its own unit
            const char* ls = chunk_begin(S, smem, wave, lane);
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                Frag w;
                lds_frag(w, ls, c);
                step3(H, Lo, w, AH[c], AL[c]);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) { H[v] *= ratio; Lo[v] *= ratio; }
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int s0 = half * CHUNK_STEPS;
        if (live && sm > s0) {                       // wave-uniform: this tile has k-steps in [s0, s0 + 8)
            const char* ls = chunk_begin(S, smem, wave, lane);
            Frag w[2];
            lds_frag(w[0], ls, 0);
#pragma unroll
            for (int cc = 0; cc < CHUNK_STEPS; ++cc) {
                const int c = s0 + cc;
                if (cc + 1 < CHUNK_STEPS) lds_frag(w[(cc + 1) & 1], ls, cc + 1);
                step3(H, Lo, w[cc & 1], XH[c], XL[c]);
                if (epi) fwd_epi_elem(c, act, Hp, Lp, U, E, m_run);
                __builtin_amdgcn_sched_barrier(0);   // no motion across k-steps: the scheduler would hoist every LDS read of the chunk
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < CHUNK_STEPS; ++cc) {
                const int c = s0 + cc;
                if (epi) fwd_epi_elem(c, act, Hp, Lp, U, E, m_run);
            }
        }
    }
    if (epi_head && h == 0 && alive) {               // head outputs: features 0..3 of the head tile = registers 0..3 of lanes h == 0
        float4 o;
        o.x = fmaf(fmaf(Lp[0], LO_INV, Hp[0]), Uh, L.head_b ? g1(L.head_b) : 0.f);
        o.y = L.n_head > 1 ? fmaf(fmaf(Lp[1], LO_INV, Hp[1]), Uh, L.head_b ? g1(L.head_b + 1) : 0.f) : 0.f;
        o.z = L.n_head > 2 ? fmaf(fmaf(Lp[2], LO_INV, Hp[2]), Uh, L.head_b ? g1(L.head_b + 2) : 0.f) : 0.f;
        o.w = L.n_head > 3 ? fmaf(fmaf(Lp[3], LO_INV, Hp[3]), Uh, L.head_b ? g1(L.head_b + 3) : 0.f) : 0.f;
        stg4(L.head_out + (size_t)row * 4, o);
    }
}

template <int NA>
__global__ __launch_bounds__(256, 1) void fwd_ro_kernel(nero_fwd_chain ch_arg, int n_rows, int rows_alloc) {
    CFwdChain& ch = *(CFwdChain*)__builtin_amdgcn_kernarg_segment_ptr();     // == ch_arg (first argument)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * RO_ROWS + 32 * wave;
    const int row = row0 + i;
    const bool alive = row0 < rows_alloc;            // wave-uniform: this wave's rows exist in the caller's buffers
    const int grow = row < n_rows ? row : n_rows - 1;
    char* ybase = smem + wave * Y_WAVE_BYTES;

    // the ring starts zeroed: chunks are executed as whole 8-step blocks and a short slab leaves the tail of its slot untouched
    for (int q = tid; q < 2 * SLOT_BYTES / 16; q += 256) *reinterpret_cast<uint4*>(smem + RING_OFF + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // ---- weight stream: first chunk ----------------------------------------------------------------------------------------
    FwdShared S;
    S.n_chunk = 0;
    {
        const Chunk first = first_chunk_from(ch, 0);
        if (first.steps > 0) issue_chunk(first, smem + RING_OFF, wave, lane);
    }

    uint4 XH[16], XL[16];                            // B-operand plane fragments of the current layer's input, per k-step
    uint4 AH[NA], AL[NA];                            // ... of the aux input (skip connections), loaded once
    float rs_main = 1.f, rs_aux = 1.f;               // row scales 2^e of this lane's row
#pragma unroll
    for (int c = 0; c < 16; ++c) { XH[c] = make_uint4(0u, 0u, 0u, 0u); XL[c] = make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int c = 0; c < NA; ++c) { AH[c] = make_uint4(0u, 0u, 0u, 0u); AL[c] = make_uint4(0u, 0u, 0u, 0u); }

    // ---- aux planes (through the wave's LDS image) ---------------------------------------------------------------------------
    if (ch.aux && alive) {
        float m = 0.f;
#pragma unroll
        for (int T = 0; T < (NA + 1) / 2; ++T) m = fmaxf(m, load_tile_to_y(ybase, ch.aux, ch.ld_aux, ch.k_aux, grow, T, i, h));
        const int e = scale_exp(partner_max(m));
        const float inv = pow2i(-e);
        rs_aux = pow2i(e);
        const int a_steps = (ch.k_aux + 15) >> 4;
#pragma unroll
        for (int c = 0; c < NA; ++c)
            if (c < a_steps) y_to_frag(ybase, i, h, c, inv, AH[c], AL[c]);
    }
    // ---- init values: "outputs of layer -1" ---------------------------------------------------------------------------------
    int y_steps = 0;                                 // k-steps of the LDS image that hold the next layer's input
    float m_run = 0.f;                               // running row maximum of the image
    if (ch.init) {
        y_steps = (ch.k_init + 15) >> 4;
        if (alive) {
            const int tiles = (ch.k_init + 31) >> 5;
            for (int T = 0; T < tiles; ++T) m_run = fmaxf(m_run, load_tile_to_y(ybase, ch.init, ch.ld_init, ch.k_init, grow, T, i, h));
        }
    }

    f32x16 acc[2][2];                                // [tile parity][H, L]
    bool fresh = true;                               // the LDS image holds values not yet turned into fragments
    for (int l = 0; l < ch.n_layers; ++l) {
        const nero_fwd_layer L = fwd_layer_copy(ch.layer[l]);   // by value: the fields live in SGPRs for the whole layer
        const int nt = L.n_tiles, has_head = L.n_head > 0 ? 1 : 0;
        const int ntt = nt + has_head;
        if (ntt == 0) continue;
        const int sm = L.k_main >> 4, sa = L.k_aux >> 4;
        stream_enter_layer(ch, S, make_sched(L), l);
        // ---- LDS image (previous layer's outputs) -> this layer's operand fragments (zero beyond the image: whole chunks run) ----
        if (fresh) {
            const int e = scale_exp(partner_max(m_run));
            const float inv_in = pow2i(-e);
            rs_main = pow2i(e);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < y_steps) y_to_frag(ybase, i, h, c, inv_in, XH[c], XL[c]);
                else { XH[c] = make_uint4(0u, 0u, 0u, 0u); XL[c] = make_uint4(0u, 0u, 0u, 0u); }
            }
        }
        // result units of the accumulators (per lane = per row)
        const float wm = sm ? g1(L.w_main) : 1.f, wa = sa ? g1(L.w_aux) : 1.f, wh = has_head ? g1(L.head_w) : 1.f;
        const float u_main = wm * rs_main, u_aux = wa * rs_aux;
        const float U = sm ? u_main : u_aux;
        const float ratio = (sa && sm) ? u_aux / u_main : 1.f;         // exact: powers of two
        const float Uh = wh * rs_main;
        if (nt > 0) { fresh = true; } else fresh = false;       // a dense layer writes a new image; a head-only layer keeps it
        float m_new = 0.f;                                     // row maximum of the image this layer writes
        for (int T = 0; T < ntt + 2; T += 2) {
            fwd_tile<0, NA>(L, S, smem, ybase, T, nt, ntt, acc, XH, XL, AH, AL, U, Uh, ratio, m_new, wave, lane, i, h, row, alive);
            fwd_tile<1, NA>(L, S, smem, ybase, T + 1, nt, ntt, acc, XH, XL, AH, AL, U, Uh, ratio, m_new, wave, lane, i, h, row, alive);
        }
        if (nt > 0) m_run = m_new;
        if (nt > 0) y_steps = (nt * 32) >> 4;
    }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------------
int nero_ro_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream) {
    for (int l = 0; l < ch->n_layers; ++l) {
        const nero_fwd_layer& L = ch->layer[l];
        if ((L.k_main | L.k_aux) & 15) return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3r): k_main / k_aux must be multiples of 16");
        if (L.k_main > 256 || L.k_aux > 96 || L.n_tiles > 8) return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3r): layer too wide");
        if (L.n_head > 0 && ((L.head_k & 15) || L.head_k > 256)) return nero_fail(NERO_ERR_ARG, "nero_mlp_forward(f16x3r): head_k must be a multiple of 16");
    }
    const int rows_alloc = (n_rows + 63) / 64 * 64;
    const dim3 grid((n_rows + RO_ROWS - 1) / RO_ROWS), block(256);
    if (ch->k_aux > 48) {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_ro_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, RO_LDS_BYTES));
        hipLaunchKernelGGL(fwd_ro_kernel<6>, grid, block, RO_LDS_BYTES, stream, *ch, n_rows, rows_alloc);
    } else {
        NERO_ONCE(hipFuncSetAttribute((const void*)fwd_ro_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, RO_LDS_BYTES));
        hipLaunchKernelGGL(fwd_ro_kernel<3>, grid, block, RO_LDS_BYTES, stream, *ch, n_rows, rows_alloc);
    }
    return NERO_OK;
}
