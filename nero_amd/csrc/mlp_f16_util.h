// mlp_f16_util.h -- device helpers shared by the two fp16 two-plane engines (mlp_f16x3.hip: activations in LDS, 64-row tiles;
// mlp_ro.hip: activations in registers, weights streamed through LDS): block scaling, fp32 <-> fp16 plane pairs, the softplus
// (beta = 100) family.  Include inside an anonymous namespace.
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float BETA = 100.0f;
constexpr int SA = 528;                 // bytes per row of a main plane: 256 fp16 + 16 B pad
constexpr int PLANE_A = 64 * SA;
constexpr int SX_N = 112, SX_W = 208;   // aux plane row strides (48 / 96 columns + 16 B)
constexpr float LO = 2048.f, LO_INV = 1.f / 2048.f;
constexpr int HDR_BYTES = 256;          // packed-image header: float[0] = 2^ew (the factor results are multiplied by), uint[1] = max bits
constexpr int SCR_LD = 36, SCR_BYTES = 32 * SCR_LD * 4;

// ---- scaling -----------------------------------------------------------------------------------------------------------
// exponent e with m * 2^-e in [0.5, 1) for normal m > 0 (0 for m == 0 / denormal), clamped to [-40, 40]
__device__ __forceinline__ int scale_exp(float m) {
    const int eb = (__float_as_uint(m) >> 23) & 0xff;
    int e = eb ? eb - 126 : 0;
    e = e < -40 ? -40 : (e > 40 ? 40 : e);
    return e;
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
    l = pk_f16((a - (float)hh[0]) * LO, (b - (float)hh[1]) * LO);
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
    v.x = fmaf((float)b0[0], LO_INV, (float)a0[0]);
    v.y = fmaf((float)b0[1], LO_INV, (float)a0[1]);
    v.z = fmaf((float)b1[0], LO_INV, (float)a1[0]);
    v.w = fmaf((float)b1[1], LO_INV, (float)a1[1]);
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
__device__ __forceinline__ float amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ float4 scale4(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

