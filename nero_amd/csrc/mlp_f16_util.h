// mlp_f16_util.h -- device helpers of the fp16 two-plane chain kernels (mlp_f16x3.hip: one 512-thread workgroup per CU): block scaling, fp32 <-> fp16 plane pairs, the softplus (beta = 100) family,
// the three-product GEMM core.  Include inside an anonymous namespace.
#pragma once

// ---- phase timing (scripts/f16_variants.sh, -DF16_PHASE_TIMING): shader-clock cycles of wave 0 of every workgroup per phase ----
#ifdef F16_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#define PH_DECL long long ph_t = clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PH(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); ph_acc[k] += t_ - ph_t; ph_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PH_PARAM , long long& ph_t, long long (&ph_acc)[8]
#define PH_ARG , ph_t, ph_acc
#define PH_END do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_phase[k_], (unsigned long long)ph_acc[k_]); } while (0)
#else
#define PH_PARAM
#define PH_ARG
#define PH_DECL
#define PH(k)
#define PH_END
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float BETA = 100.0f;
constexpr int SA = 528;                 // bytes per row of a main plane: 256 fp16 + 16 B pad
constexpr int PLANE_A = 64 * SA;
constexpr int SX_N = 112, SX_W = 208;   // aux plane row strides (48 / 96 columns + 16 B)
constexpr float LO = 2048.f, LO_INV = 1.f / 2048.f;
// Round 5 -- ONE accumulator set (default; -DF16_TWO_ACC restores the H / L pair of rounds 1-4).  The block scale puts an operand's
// largest magnitude into [2^14, 2^15), the TOP of fp16's range, instead of [0.5, 1): the remainder l = fp16(xs - h) is then an ordinary
// fp16 number at its TRUE scale (|l| <= 2^3 for the largest elements, normal down to 2^-14, i.e. for every element within 2^-17 of the
// row maximum; below that its absolute error 2^-25 is 2^-40 of the maximum) -- no 2^11 pre-scale -- and the three plane products
// hw hx + hw lx + lw hx accumulate into the SAME fp32 accumulator inside the matrix pipe (what the weight-gradient GEMM has done since
// round 2, mlp_f16dw.hip).  Same three MFMAs per k-step; what goes away is the second accumulator set (32 VGPRs per wave), the
// H + 2^-11 L combine (32 VALU per lane and layer), the 2^11 multiply of every split (32 more) and half of the skip-layer rescale.
// Products reach 2^30, a 256-term sum 2^38: far inside fp32.  Error against fp64: tests/test_mlp_engine.py (unchanged tolerances).
#ifdef F16_TWO_ACC
constexpr int F16_TOP = 0;
#define ACCV(aH, aL, r, v) fmaf((aL)[r][v], LO_INV, (aH)[r][v])
#else
constexpr int F16_TOP = 15;
#define ACCV(aH, aL, r, v) ((aH)[r][v])
#endif
constexpr int HDR_BYTES = 256;          // packed-image header: float[0] = 2^ew (the factor results are multiplied by), uint[1] = max bits
constexpr int SCR_LD = 36, SCR_BYTES = 32 * SCR_LD * 4;

// row-major workspace stores (saved activations, deltas, tangents: 1 KB per row and layer, written once, read by a LATER kernel)
// carry the non-temporal policy so that they do not displace the weight images -- re-read by every tile -- from the XCD's L2:
// -3.3 % on the whole training step (forward -2 %, reverse -4 %, tangent -4 %; -DNERO_PLAIN_STORES restores the default policy).
// The mirror image on the read side (nt loads / nt LDS-DMA of those workspaces in the consumer kernels) LOSES 0.5-0.9 ms per step:
// much of what a pass wrote is still in the MALL when the next pass asks for it.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_ws4(float* dst, float4 v) {
#ifdef NERO_PLAIN_STORES
    *reinterpret_cast<float4*>(dst) = v;
#else
    f32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(dst));
#endif
}

constexpr int SCRP_ROWS = 16;
constexpr int SCRP_BYTES = SCRP_ROWS * SCR_LD * 4;     // 2304 B per wave: store-transposition scratch, 16 rows at a time

// accumulator-layout 32x32 block -> row-major global store, ROWS (16 or 8) rows at a time through the wave-private scratch
template <int ROWS>
__device__ __forceinline__ void acc_to_global_rows(float* scr, const float4 (&q)[4], float* __restrict__ gblock, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int p = 0; p < 32 / ROWS; ++p) {
        if (i / ROWS == p) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(scr + (i % ROWS) * SCR_LD + 8 * g + 4 * h) = q[g];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < ROWS / 8; ++k) {
            const int idx = lane + 64 * k, row = idx >> 3, c4 = 4 * (idx & 7);
            store_ws4(gblock + (size_t)(ROWS * p + row) * NERO_HID + c4, *reinterpret_cast<const float4*>(scr + row * SCR_LD + c4));
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__device__ __forceinline__ void acc_to_global16(float* scr, const float4 (&q)[4], float* __restrict__ gblock, int lane) {
    acc_to_global_rows<16>(scr, q, gblock, lane);
}

// ---- LDS-DMA ----------------------------------------------------------------------------------------------------------------
// global_load_lds_dwordx4: lane l's 16 bytes at `gptr` land at LDS byte address `lds_addr` + 16 l, no VGPRs in between.  Issued
// through inline asm on purpose: the compiler's wait-count pass otherwise puts a vmcnt(0) in front of EVERY later LDS access
// (it cannot prove the rest of the dynamic LDS array disjoint from the DMA target), which turns the prefetch into a stall.  Its
// vmcnt bookkeeping stays conservative (an untracked older load only makes counted waits longer); the reader must issue its own
// s_waitcnt vmcnt(0) before touching the target.
__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_addr) {
    // M0 = LDS base of the transfer.  It is handed over as a "{m0}"-constrained INPUT, so the compiler materialises the s_mov_b32 m0
    // itself and knows the register is live here (a clobber of m0 is ignored by hipcc: "reserved register").  The s_nop covers the
    // S_MOV-to-M0 -> LDS-DMA wait state, which hipcc's hazard recogniser cannot insert for an instruction inside an asm block.
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "{m0}"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(size_t)p; }      // generic -> LDS byte offset

// ---- scaling -----------------------------------------------------------------------------------------------------------
// exponent e with m * 2^-e in [0.5, 1) * 2^F16_TOP for normal m > 0 (-F16_TOP for m == 0 / denormal).  The exponent of m is taken from
// [2^-40, 2^100]: below, the block is scaled as if its maximum were 2^-40 (everything in it is then tiny but finite in the planes); the
// upper end keeps 2^e and 2^-e normal floats with room to spare (|e - F16_TOP| <= 115 < 126).  Rounds 1-5 clamped the exponent to +-40 BEFORE the shift by F16_TOP: with the
// one-accumulator format a block maximum beyond 2^40 would then have been scaled above 65504 and become fp16 infinity (the
// two-accumulator format stayed finite to 2^56).  No value of the training step comes near either end.
__device__ __forceinline__ int scale_exp(float m) {
    const int eb = (__float_as_uint(m) >> 23) & 0xff;
    int e = eb ? eb - 126 : 0;
    e = e < -40 ? -40 : (e > 100 ? 100 : e);
    return e - F16_TOP;                                  // (one accumulator: m * 2^-e in [2^14, 2^15), the top of fp16's range)
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

// ---- fp32 <-> fp16 plane pairs -----------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_f16(float a, float b) {       // v_cvt_pk_f16_f32 (RN): a -> low half
    f32x2 v = {a, b};
    f16x2 c = __builtin_convertvector(v, f16x2);
    return __builtin_bit_cast(unsigned, c);
}
__device__ __forceinline__ void split2h(float a, float b, unsigned& h, unsigned& l) {     // a, b already block-scaled
    h = pk_f16(a, b);
    const f16x2 hh = __builtin_bit_cast(f16x2, h);
#ifdef F16_TWO_ACC
    l = pk_f16((a - (float)hh[0]) * LO, (b - (float)hh[1]) * LO);
#else
    l = pk_f16(a - (float)hh[0], b - (float)hh[1]);
#endif
}
__device__ __forceinline__ void store_planes4h(char* dst, int plane_bytes, float4 v) {
    unsigned h0, l0, h1, l1;
    split2h(v.x, v.y, h0, l0);
    split2h(v.z, v.w, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + plane_bytes) = make_uint2(l0, l1);
}
__device__ __forceinline__ float4 load_planes4h(const char* src, int plane_bytes) {       // -> block-scaled values
    const uint2 a = *reinterpret_cast<const uint2*>(src);
    const uint2 b = *reinterpret_cast<const uint2*>(src + plane_bytes);
    const f16x2 a0 = __builtin_bit_cast(f16x2, a.x), a1 = __builtin_bit_cast(f16x2, a.y);
    const f16x2 b0 = __builtin_bit_cast(f16x2, b.x), b1 = __builtin_bit_cast(f16x2, b.y);
    float4 v;
#ifdef F16_TWO_ACC
    v.x = fmaf((float)b0[0], LO_INV, (float)a0[0]);
    v.y = fmaf((float)b0[1], LO_INV, (float)a0[1]);
    v.z = fmaf((float)b1[0], LO_INV, (float)a1[0]);
    v.w = fmaf((float)b1[1], LO_INV, (float)a1[1]);
#else
    v.x = (float)a0[0] + (float)b0[0];
    v.y = (float)a0[1] + (float)b0[1];
    v.z = (float)a1[0] + (float)b1[0];
    v.w = (float)a1[1] + (float)b1[1];
#endif
    return v;
}

// ---- activations (identical to mlp_split.hip) ---------------------------------------------------------------------------
// softplus(beta = 100) on the hardware exp2 / log2 units (v_exp_f32 / v_log_f32, ~1 ulp each, no range fix-ups needed here:
// the exp2 argument is <= 0 and the log2 argument lies in [1, 2]):   max(x, 0) + ln(1 + exp(-|100 x|)) / 100.
// Absolute error <= 1e-9 (the rounding of 1 + t), i.e. far below one fp32 ulp of the row maximum every value is block-scaled
// against.  Beyond 100 x > 20 the correction term is < 2e-11 and the sum rounds to x itself: torch's threshold rule
// (nn.Softplus(beta=100), network/field.py:124) without a branch.  8 VALU ops instead of ~30 (the epilogue is issue-bound).
__device__ __forceinline__ float softplus100(float x) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 144.26950408889634f);
    const float l = __builtin_amdgcn_logf(1.0f + t);
    return fmaf(l, 0.0069314718055994531f, fmaxf(x, 0.f));
}
template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
    if (ACT == NERO_ACT_RELU) return fmaxf(x, 0.f);
    if (ACT == NERO_ACT_SOFTPLUS100) return softplus100(x);
    return x;
}

// sigma'(z) of softplus(beta = 100) from the SAVED OUTPUT a = softplus(z):  1 - exp(-100 a)  (= sigmoid(100 z)); a short series
// below 100 a = 1/32 keeps the relative accuracy where the subtraction would cancel; 1 beyond torch's threshold.
__device__ __forceinline__ float softplus100_grad_from_out(float a) {
    const float ba = BETA * a;
    const float ps = ba * (1.f + ba * (-0.5f + ba * (0.16666667f + ba * (-0.041666668f))));
    const float pe = 1.f - __builtin_amdgcn_exp2f(ba * -1.4426950408889634f);
    const float r = ba < 0.03125f ? ps : pe;
    return ba > 20.f ? 1.f : r;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float a, float g) {
    if (ACT == NERO_ACT_RELU) return a > 0.f ? g : 0.f;
    if (ACT == NERO_ACT_SOFTPLUS100) return g * softplus100_grad_from_out(a);
    return g;
}
__device__ __forceinline__ void tan_elem(float a, float zd, float gb, bool live, float& ad, float& ij) {
    const float s = softplus100_grad_from_out(a);
    ad = s * zd;
    const float r2 = (BETA * a > 20.f) ? 0.f : BETA * (1.f - s);
    ij = live ? gb * r2 * zd : 0.f;
}
// the same injection from the tangent's OUTPUT: adot = s zdot  =>  inj = gbar * beta (1 - s) * adot / s   (s == 0: a == 0 exactly, then
// gbar = upstream * s is 0 too and the term vanishes; beyond torch's threshold sigma'' is 0)
__device__ __forceinline__ float inj_elem(float a, float gb, float ad) {
    const float s = softplus100_grad_from_out(a);
    const float r2 = (BETA * a > 20.f) ? 0.f : BETA * (1.f - s);
    return s > 0.f ? gb * r2 * (ad / s) : 0.f;
}
__device__ __forceinline__ float amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ float4 scale4(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }


// ---- chain descriptors and operand scales without dependent round trips (round 6) ------------------------------------------------------
// The layer descriptors live in the kernel-argument segment.  Read through a reference (`const nero_fwd_layer& L = ch.layer[l]`) every
// field became its own s_load + s_waitcnt lgkmcnt(0) at its point of use -- about ten scalar-cache round trips per layer, serial, on the
// critical path between two barriers -- and the block scale of a packed image (`*L.w_main`, a device-side value in the image header) a
// scalar load FOLLOWED BY a global load and a vmcnt(0) in front of the GEMM, which also drained the weight prefetch.  Now: the
// descriptor of a layer is copied by value in one batch of scalar loads (PIN_LAYER keeps hipcc from sinking the loads back to the uses),
// the NEXT layer's while the matrix pipe drains the current GEMM, and the scales of every image of the chain sit in a 20-entry LDS table
// filled once per workgroup (wsc[2 l] = main, wsc[2 l + 1] = aux image of layer l).
constexpr int WSC_N = 2 * NERO_MAX_LAYERS;
constexpr int LDS_SMALL_BYTES = (64 + 64 + 512 + 32) * 4;      // rs_main[64] rs_aux[64] rmax[64][8] wsc[<= 32]
#define PIN_S(x) asm volatile("" : "+s"(x))
// (pointers are pinned as INPUTS only: a pointer redefined by an asm statement loses its provenance and hipcc addresses through it with
//  flat_load / flat_store, which count on vmcnt AND lgkmcnt -- every LDS wait of the k-loop then waits for the weight stream as well)
#define PIN_P(x) asm volatile("" : : "s"(x))
__device__ __forceinline__ nero_fwd_layer load_layer(const nero_fwd_chain& ch, int l) {
    nero_fwd_layer L = ch.layer[l];
#ifndef NERO_NO_PIN
    PIN_P(L.w_main); PIN_P(L.w_aux); PIN_P(L.bias); PIN_P(L.save); PIN_P(L.head_w); PIN_P(L.head_b); PIN_P(L.head_out);
    PIN_S(L.k_main); PIN_S(L.k_aux); PIN_S(L.n_tiles); PIN_S(L.n_head); PIN_S(L.act); PIN_S(L.head_k); PIN_P(L.relu_mask);
#endif
    return L;
}
__device__ __forceinline__ nero_tan_layer load_layer(const nero_tan_chain& ch, int l) {
    nero_tan_layer L = ch.layer[l];
#ifndef NERO_NO_PIN
    PIN_P(L.w_main); PIN_P(L.w_aux); PIN_P(L.a_saved); PIN_P(L.gbar); PIN_P(L.adot); PIN_P(L.inj);
    PIN_S(L.k_main); PIN_S(L.k_aux); PIN_S(L.n_tiles);
#endif
    return L;
}
__device__ __forceinline__ nero_bwd_layer load_layer(const nero_bwd_chain& ch, int l) {
    nero_bwd_layer L = ch.layer[l];
#ifndef NERO_NO_PIN
    PIN_P(L.w_main_t); PIN_P(L.w_aux_t); PIN_P(L.a_prev); PIN_P(L.inj); PIN_P(L.delta_prev); PIN_P(L.head_w); PIN_P(L.head_dy);
    PIN_S(L.n_out); PIN_S(L.k_main_tiles); PIN_S(L.k_aux_tiles); PIN_S(L.n_head); PIN_S(L.act_prev); PIN_P(L.mask_prev); PIN_P(L.inj_adot);
#endif
    return L;
}
// request the scale of every packed image of the chain (uniform addresses; branch-free: a missing image reads the chain's first one) ...
struct WscRegs { float v[WSC_N]; };
template <class CH, class F> __device__ __forceinline__ void wsc_request(WscRegs& r, const CH& ch, F&& images) {
    const float* fb = nullptr;
#pragma unroll
    for (int l = NERO_MAX_LAYERS - 1; l >= 0; --l) {
        const float *pm, *pa;
        images(ch.layer[l], pm, pa);
        if (l < ch.n_layers) { fb = pa ? pa : fb; fb = pm ? pm : fb; }
    }
#pragma unroll
    for (int k = 0; k < WSC_N; ++k) r.v[k] = 1.f;
    if (fb == nullptr) return;
#pragma unroll
    for (int l = 0; l < NERO_MAX_LAYERS; ++l) {
        const float *pm, *pa;
        images(ch.layer[l], pm, pa);
        const bool in = l < ch.n_layers;
        pm = (in && pm) ? pm : fb;
        pa = (in && pa) ? pa : fb;
        r.v[2 * l] = *pm;
        r.v[2 * l + 1] = *pa;
    }
}
// ... and put them into the table (one lane; the caller's next barrier publishes it)
__device__ __forceinline__ void wsc_commit(float* wsc, const WscRegs& r, int tid) {
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < WSC_N; k += 4) *reinterpret_cast<float4*>(wsc + k) = make_float4(r.v[k], r.v[k + 1], r.v[k + 2], r.v[k + 3]);
    }
}

// ---- GEMM core ------------------------------------------------------------------------------------------------------------
// accH[r] += wh xh,  accL[r] += wh xl + wl xh  over `n` k-steps of 16 (r = 32-row half).  Weight planes three steps ahead in a
// ring of four register sets (L2 stream), activation planes one step ahead in a double buffer (LDS).
struct WF { uint4 wh, wl; };
struct XF { uint4 xh0, xl0, xh1, xl1; };

__device__ __forceinline__ void load_w(WF& o, const uint4* wp, int c) {
#ifdef F16_NO_WSTREAM                               // (timing experiments, scripts/f16_variants.sh: every k-step re-reads step 0)
    c = 0;
#endif
    const uint4* w = wp + (size_t)c * 128;
    o.wh = w[0];
    o.wl = w[64];
}
__device__ __forceinline__ void load_x(XF& o, const char* xp, int half_bytes, int plane_bytes, int c) {
    const char* x = xp + c * 32;
    o.xh0 = *reinterpret_cast<const uint4*>(x);
    o.xl0 = *reinterpret_cast<const uint4*>(x + plane_bytes);
    x += half_bytes;
    o.xh1 = *reinterpret_cast<const uint4*>(x);
    o.xl1 = *reinterpret_cast<const uint4*>(x + plane_bytes);
}
#define NERO_MFH(ACC, A, B) \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)
// GEMM_SETPRIO (experiment, round 6): 1 = a wave raises its issue priority for the six MFMAs of a k-step and drops it for the operand
// requests of the next one (the two waves of a SIMD would then alternate k-steps instead of one running ahead and the other finishing
// alone at the lone-wave rate); 2 = the second wave of every SIMD (waves 4-7) holds priority 1 through the whole GEMM.
#ifndef GEMM_SETPRIO
#define GEMM_SETPRIO 0
#endif
__device__ __forceinline__ void ops_compute(f32x16 (&aH)[2], f32x16 (&aL)[2], const WF& w, const XF& x) {
#if GEMM_SETPRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
#ifdef F16_NO_MFMA
    aH[0][0] += __uint_as_float(w.wh.x ^ x.xh0.x); aL[0][0] += __uint_as_float(w.wl.x ^ x.xl0.x);
    aH[1][0] += __uint_as_float(w.wh.y ^ x.xh1.x); aL[1][0] += __uint_as_float(w.wl.y ^ x.xl1.x);
    return;
#endif
#ifdef F16_TWO_ACC
    NERO_MFH(aL[0], w.wl, x.xh0); NERO_MFH(aL[1], w.wl, x.xh1);
    NERO_MFH(aH[0], w.wh, x.xh0); NERO_MFH(aH[1], w.wh, x.xh1);
    NERO_MFH(aL[0], w.wh, x.xl0); NERO_MFH(aL[1], w.wh, x.xl1);
#else
    // (the two small products first: they enter an accumulator that still holds little; dependent MFMAs on one accumulator issue back
    //  to back at full rate -- accumulation forwarding, measured for the weight-gradient kernel in round 2)
    NERO_MFH(aH[0], w.wl, x.xh0); NERO_MFH(aH[1], w.wl, x.xh1);
    NERO_MFH(aH[0], w.wh, x.xl0); NERO_MFH(aH[1], w.wh, x.xl1);
    NERO_MFH(aH[0], w.wh, x.xh0); NERO_MFH(aH[1], w.wh, x.xh1);
    (void)aL;
#endif
#if GEMM_SETPRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
}
#define NERO_FENCE() __builtin_amdgcn_sched_barrier(0)      // (without the fences hipcc sinks the prefetches: the kernels run 40 % slower)
// One k-step = 2 weight loads (VMEM) + 4 activation-fragment reads (DS) for a LATER step + 6 MFMAs of this step.  Issued as
// [6 loads][6 MFMAs] the wave spends the load-issue time with an idle matrix pipe and the MFMA time with idle issue slots (an in-order
// wave: a lone wave reaches 46 % of the pipe rate in the loop, two per SIMD 70 %, profiles/r03_phase_probes.txt).  -DGEMM_SGB interleaves
// them -- MFMA, DS, MFMA, DS, ..., MFMA, VMEM, MFMA, VMEM (checked in the ISA) -- so that each load would issue in the shadow of the
// preceding MFMA.  MEASURED SLOWER (round 3, same file): the GEMM phase of a layer-tile goes from 8.6 k to 9.4 k cycles (forward),
// 4.7 k to 6.0 k (reverse), 17.6 k to 19.4 k (two workgroups per CU) -- a load between two MFMAs costs the pipe more than its issue
// slot.  Kept as an experiment switch, off.
#ifdef GEMM_SGB
#define NERO_KSTEP_SCHED() do { \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } while (0)
#define NERO_MID_FENCE()
#else
#define NERO_KSTEP_SCHED()
#define NERO_MID_FENCE() NERO_FENCE()
#endif

// The first three weight fragments of a GEMM can be requested by the CALLER ahead of time (`pre`: wa, wb, wc already hold -- or are about
// to receive -- steps 0, 1, 2; see prefetch_w): a layer's first MFMA then waits for an LDS read instead of an L2 round trip behind
// the barrier that ends the previous layer (~900 cycles per layer-tile).
#ifndef F16_PW_N
#define F16_PW_N 3                                   // fragments requested ahead by the caller (the rest at the start of the GEMM)
#endif
__device__ __forceinline__ void prefetch_w(WF& wa, WF& wb, WF& wc, const uint4* wp, int n, int from = 0) {
    const int last = n - 1;
    if (from <= 0) load_w(wa, wp, 0);
    if (from <= 1) load_w(wb, wp, 1 < last ? 1 : last);
    if (from <= 2) load_w(wc, wp, 2 < last ? 2 : last);
}
__device__ __forceinline__ void gemm_f16x3_loop(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                           int plane_bytes, int n, bool pre, WF& wa, WF& wb, WF& wc) {
    if (n <= 0) return;
    WF wd;
    XF xa, xb;
    const int last = n - 1;
#define NERO_CL(c) ((c) < last ? (c) : last)
    prefetch_w(wa, wb, wc, wp, n, pre ? F16_PW_N : 0);
    load_x(xa, xp, half_bytes, plane_bytes, 0);
    NERO_FENCE();
    for (int c = 0; c < n; c += 4) {
        load_w(wd, wp, NERO_CL(c + 3)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 1)); NERO_MID_FENCE();
        ops_compute(aH, aL, wa, xa); NERO_KSTEP_SCHED(); NERO_FENCE();
        if (c + 1 < n) {
            load_w(wa, wp, NERO_CL(c + 4)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 2)); NERO_MID_FENCE();
            ops_compute(aH, aL, wb, xb); NERO_KSTEP_SCHED(); NERO_FENCE();
        }
        if (c + 2 < n) {
            load_w(wb, wp, NERO_CL(c + 5)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 3)); NERO_MID_FENCE();
            ops_compute(aH, aL, wc, xa); NERO_KSTEP_SCHED(); NERO_FENCE();
        }
        if (c + 3 < n) {
            load_w(wc, wp, NERO_CL(c + 6)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 4)); NERO_MID_FENCE();
            ops_compute(aH, aL, wd, xb); NERO_KSTEP_SCHED(); NERO_FENCE();
        }
    }
#undef NERO_CL
}
__device__ __forceinline__ void gemm_f16x3_loop(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                           int plane_bytes, int n) {
    WF wa, wb, wc;
    gemm_f16x3_loop(aH, aL, wp, xp, half_bytes, plane_bytes, n, false, wa, wb, wc);
}


// lean k-loop for the kernels that run FOUR waves per SIMD (<= 128 registers per wave; mlp_f16p.hip, NW = 8): weights two steps ahead in a
// ring of three register sets, fragments one step ahead -- 56 operand registers instead of 64 + 32; the other three waves of the SIMD cover
// what the shallower ring no longer does.  Same products in the same order as gemm_f16x3_loop.
__device__ __forceinline__ void gemm_f16x3_lean(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                                int plane_bytes, int n) {
    if (n <= 0) return;
    WF w0, w1, w2;
    XF xa, xb;
    const int last = n - 1;
#define NERO_CL(c) ((c) < last ? (c) : last)
    load_w(w0, wp, 0);
    load_w(w1, wp, NERO_CL(1));
    load_x(xa, xp, half_bytes, plane_bytes, 0);
    NERO_FENCE();
    for (int c = 0; c < n; c += 6) {
        load_w(w2, wp, NERO_CL(c + 2)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 1)); NERO_FENCE();
        ops_compute(aH, aL, w0, xa); NERO_FENCE();
        if (c + 1 < n) { load_w(w0, wp, NERO_CL(c + 3)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 2)); NERO_FENCE(); ops_compute(aH, aL, w1, xb); NERO_FENCE(); }
        if (c + 2 < n) { load_w(w1, wp, NERO_CL(c + 4)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 3)); NERO_FENCE(); ops_compute(aH, aL, w2, xa); NERO_FENCE(); }
        if (c + 3 < n) { load_w(w2, wp, NERO_CL(c + 5)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 4)); NERO_FENCE(); ops_compute(aH, aL, w0, xb); NERO_FENCE(); }
        if (c + 4 < n) { load_w(w0, wp, NERO_CL(c + 6)); load_x(xb, xp, half_bytes, plane_bytes, NERO_CL(c + 5)); NERO_FENCE(); ops_compute(aH, aL, w1, xa); NERO_FENCE(); }
        if (c + 5 < n) { load_w(w1, wp, NERO_CL(c + 7)); load_x(xa, xp, half_bytes, plane_bytes, NERO_CL(c + 6)); NERO_FENCE(); ops_compute(aH, aL, w2, xb); NERO_FENCE(); }
    }
#undef NERO_CL
}

// compile-time step count (the 256-wide layers: 16 steps), fully unrolled: no clamps, no branches, every ring slot a fixed register
// set.  XD = how many steps ahead the activation fragments are requested (ring of XD + 1 sets), weights three steps ahead.
template <int V> struct IC { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
#ifndef GEMM_XD
#define GEMM_XD 1
#endif
#ifndef GEMM_WD
#define GEMM_WD 3
#endif
// `hook(IC<c>)` runs in every step behind that step's operand requests: the caller issues its own global loads at the step after
// which the GEMM requests no more weights (c = N - WD: vmcnt retires in order, a load issued earlier would hold the weight stream back,
// one issued later has less of the GEMM left to hide behind).
template <int N, class F>
__device__ __forceinline__ void gemm_f16x3_fixed_hook(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                                      int plane_bytes, F&& hook) {
    constexpr int XD = GEMM_XD, XN = XD + 1, WD = GEMM_WD, WN = WD + 1;
    WF w[WN];
    XF x[XN];
    static_for<0, WD>([&](auto c) { if (c.value < N) load_w(w[c.value], wp, c.value); });
    static_for<0, XD>([&](auto c) { if (c.value < N) load_x(x[c.value], xp, half_bytes, plane_bytes, c.value); });
    NERO_FENCE();
    static_for<0, N>([&](auto cc) {
        constexpr int c = cc.value;
        if (c + WD < N) load_w(w[(c + WD) % WN], wp, c + WD);
        if (c + XD < N) load_x(x[(c + XD) % XN], xp, half_bytes, plane_bytes, c + XD);
        hook(cc);
        NERO_MID_FENCE();
        ops_compute(aH, aL, w[c % WN], x[c % XN]);
        if (c + WD < N && c + XD < N) NERO_KSTEP_SCHED();
        NERO_FENCE();
    });
}
template <int N>
__device__ __forceinline__ void gemm_f16x3_fixed(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                                 int plane_bytes) {
    gemm_f16x3_fixed_hook<N>(aH, aL, wp, xp, half_bytes, plane_bytes, [](auto) {});
}

__device__ __forceinline__ void gemm_f16x3(f32x16 (&aH)[2], f32x16 (&aL)[2], const uint4* wp, const char* xp, int half_bytes,
                                           int plane_bytes, int n) {
#ifdef GEMM_ASSUME16                                   // (timing experiments on all-256-wide chains only: wrong for any other K)
    gemm_f16x3_fixed<16>(aH, aL, wp, xp, half_bytes, plane_bytes); return;
#endif
    gemm_f16x3_loop(aH, aL, wp, xp, half_bytes, plane_bytes, n);
}

__device__ __forceinline__ void zero2(f32x16 (&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
}

// ---- lane exchanges without the LDS (round 6) ---------------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 + a bounds select + s_waitcnt lgkmcnt: an LDS round trip (~100+ cycles, and lgkmcnt is shared
// with the scalar loads) per exchange, several of them on the critical path of every layer epilogue.  gfx950 has the exchange of the two
// 32-lane halves as ONE VALU instruction (v_permlane32_swap) and the exchanges inside a row of 16 as DPP modifiers.
#ifdef NERO_SHFL_LDS                                   // (experiment switch: the round-5 code)
__device__ __forceinline__ float max_xor32(float m) { return fmaxf(m, __shfl_xor(m, 32)); }
__device__ __forceinline__ unsigned other_half(unsigned v, int h) { (void)h; return __shfl_xor(v, 32); }
__device__ __forceinline__ float max_8lanes(float m) { m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); return fmaxf(m, __shfl_xor(m, 4)); }
__device__ __forceinline__ float max_4lanes(float m) { m = fmaxf(m, __shfl_xor(m, 1)); return fmaxf(m, __shfl_xor(m, 2)); }
__device__ __forceinline__ float sum_8lanes(float s) { s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); return s + __shfl_xor(s, 4); }
#else
// v_permlane32_swap vdst, src0: vdst[32..63] <-> src0[0..31].  With both operands = v: r[0] = v's lower half in both halves, r[1] = its
// upper half in both halves.
__device__ __forceinline__ float max_xor32(float m) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ unsigned other_half(unsigned v, int h) {          // the value lane ^ 32 holds
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? r[0] : r[1];
}
// DPP: quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), quad_perm [2,3,0,1] = 0x4E (lane ^ 2), row_half_mirror = 0x141 (lane j <-> 7 - j of each
// group of 8: the OTHER quad, which is all a reduction over 8 lanes needs once every lane holds its quad's result)
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float max_4lanes(float m) { m = fmaxf(m, dpp_f32<0xB1>(m)); return fmaxf(m, dpp_f32<0x4E>(m)); }
__device__ __forceinline__ float max_8lanes(float m) { m = max_4lanes(m); return fmaxf(m, dpp_f32<0x141>(m)); }
// (same bits as the shuffle butterfly: every step adds the partner's partial sum to the lane's own, and fp32 addition commutes)
__device__ __forceinline__ float sum_8lanes(float s) { s += dpp_f32<0xB1>(s); s += dpp_f32<0x4E>(s); return s + dpp_f32<0x141>(s); }
#endif

// row maxima of this wave's 64x32 block (two rows per lane) -> rmax[row][wave]
__device__ __forceinline__ void publish_rowmax(float* rmax, float m0, float m1, int wave, int i, int h) {
    m0 = max_xor32(m0);
    m1 = max_xor32(m1);
    if (h == 0) { rmax[i * 8 + wave] = m0; rmax[(32 + i) * 8 + wave] = m1; }
}
__device__ __forceinline__ float row_max8(const float* rmax, int row) {
    const float4 a = *reinterpret_cast<const float4*>(rmax + row * 8);
    const float4 b = *reinterpret_cast<const float4*>(rmax + row * 8 + 4);
    return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
}

// ---- epilogue of a reverse layer (shared by mlp_f16x3.hip and mlp_f16p.hip) ---------------------------------------------------------
// PRE: the injections were requested inside the GEMM (ijp, bwd_f16_kernel<true>); otherwise they are read here
template <int ACT, bool HEAD, bool PRE>
__device__ __forceinline__ void bwd_values(const float4 (&gq)[2][4], const float4 (&pa)[2][4], size_t goff, bool has_inj,
                                           const nero_bwd_layer& L, int row0, int i, int fbase, int n_rows, float4 (&val)[2][4],
                                           float (&m)[2], const float4 (&ijp)[2][4]) {
    // the global operands of the epilogue are requested in branch-free batches ahead of their use: with the loads inside
    // `if (has_inj)` / `if (j < nh)` hipcc waited for each of them separately (147 of the reverse kernel's 240 loads were followed by
    // s_waitcnt vmcnt(0)) -- eight serial HBM round trips per layer-tile for the injections of the second-order pass, up to 32 serial
    // L2 round trips for the head weights of a chain's first step.  Now: four head-weight rows per request group, and the four
    // injection vectors of a row half together (both halves at once costs 32 more registers: 255 + spills).
    const int nh = L.n_head;
    float4 gs[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) gs[r][g] = gq[r][g];
    if (HEAD) {
        float dj[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float4 dyh = *reinterpret_cast<const float4*>(L.head_dy + (size_t)(row0 + 32 * r + i) * 4);
            dj[r][0] = dyh.x; dj[r][1] = dyh.y; dj[r][2] = dyh.z; dj[r][3] = dyh.w;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 hw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hw[j] = *reinterpret_cast<const float4*>(L.head_w + (j < nh ? j : 0) * NERO_HID + fbase + 8 * g);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nh) {
                        gs[r][g].x = fmaf(dj[r][j], hw[j].x, gs[r][g].x); gs[r][g].y = fmaf(dj[r][j], hw[j].y, gs[r][g].y);
                        gs[r][g].z = fmaf(dj[r][j], hw[j].z, gs[r][g].z); gs[r][g].w = fmaf(dj[r][j], hw[j].w, gs[r][g].w);
                    }
            __builtin_amdgcn_sched_barrier(0);         // (one request group at a time: all sixteen hoisted cost 40 registers and spills)
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool live = (row0 + 32 * r + i) < n_rows;
        float4 ij[4];
        if (PRE) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ij[g] = ijp[r][g];
        } else if (has_inj) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ij[g] = *reinterpret_cast<const float4*>(L.inj + goff + (size_t)r * 32 * NERO_HID + 8 * g);
            if (ACT == NERO_ACT_SOFTPLUS100 && L.inj_adot) {
                // `inj` is gbar and inj_adot the tangent: the sigma'' injection gbar beta (1 - s) zdot with zdot = adot / s is formed here
                // (s = sigma'(a_prev) is in hand anyway), so the tangent pass neither reads gbar nor writes a finished term
                float4 ad[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) ad[g] = *reinterpret_cast<const float4*>(L.inj_adot + goff + (size_t)r * 32 * NERO_HID + 8 * g);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = pa[r][g];
                    ij[g].x = inj_elem(a.x, ij[g].x, ad[g].x); ij[g].y = inj_elem(a.y, ij[g].y, ad[g].y);
                    ij[g].z = inj_elem(a.z, ij[g].z, ad[g].z); ij[g].w = inj_elem(a.w, ij[g].w, ad[g].w);
                }
            }
        }
        m[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = pa[r][g];
            float4 d;
            d.x = act_grad<ACT>(a.x, gs[r][g].x); d.y = act_grad<ACT>(a.y, gs[r][g].y);
            d.z = act_grad<ACT>(a.z, gs[r][g].z); d.w = act_grad<ACT>(a.w, gs[r][g].w);
            if (has_inj) { d.x += ij[g].x; d.y += ij[g].y; d.z += ij[g].z; d.w += ij[g].w; }
            if (!live) d = make_float4(0.f, 0.f, 0.f, 0.f);
            val[r][g] = d;
            m[r] = fmaxf(m[r], amax4(d));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int ACT, bool PRE>
__device__ __forceinline__ void bwd_values_h(const float4 (&gq)[2][4], const float4 (&pa)[2][4], size_t goff, bool has_inj,
                                             const nero_bwd_layer& L, int row0, int i, int fbase, int n_rows, float4 (&val)[2][4],
                                             float (&m)[2], const float4 (&ijp)[2][4]) {
    if (L.n_head > 0) bwd_values<ACT, true, PRE>(gq, pa, goff, has_inj, L, row0, i, fbase, n_rows, val, m, ijp);
    else bwd_values<ACT, false, PRE>(gq, pa, goff, has_inj, L, row0, i, fbase, n_rows, val, m, ijp);
}

