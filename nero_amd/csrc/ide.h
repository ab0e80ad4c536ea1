// ide.h -- the integrated directional encoding (generate_ide_fn(5), utils/ref_utils.py:53-117) with everything the compiler can know
// at compile time known at compile time: the 36 (l, m) pairs, the 222 polynomial coefficients, every array index.  Included by
// shade.hip (Stage I, with the roughness attenuation) and mc_shade.hip (Stage II, kappa_inv = 0) inside their anonymous namespaces.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rows.h"                                    // rows_put / rows_flush: the encoders' coalesced row stores

constexpr int IDE_N = 36;

// The same coefficients as COMPILE-TIME constants for the unrolled kernels below (literal operands of the FMAs: 222 run-time scalars
// loaded from c_ide_mat do not fit the scalar register file -- the compiler parked them in VGPR lanes, 1800 v_readlane per thread).
// The host initialisers check them against their libm-computed tables bit for bit (ide_tab_equals).
constexpr double cx_fact(int n) { double r = 1.0; for (int i = 2; i <= n; ++i) r *= i; return r; }
constexpr double cx_binom(double a, int k) { double p = 1.0; for (int i = 0; i < k; ++i) p *= (a - i); return p / cx_fact(k); }
constexpr double cx_sqrt(double x) { double r = x > 1.0 ? x : 1.0; for (int i = 0; i < 256; ++i) r = 0.5 * (r + x / r); return r; }
constexpr double cx_coeff(int l, int m, int k) {
    const double al = ((m & 1) ? -1.0 : 1.0) * (double)(1 << l) * cx_fact(l) / cx_fact(k) / cx_fact(l - k - m) * cx_binom(0.5 * (l + k + m - 1.0), l);
    return cx_sqrt((2.0 * l + 1.0) * cx_fact(l - m) / (4.0 * 3.14159265358979323846 * cx_fact(l + m))) * al;
}
struct IdeTab { float c[17 * IDE_N]; };
constexpr IdeTab make_ide_tab() {
    IdeTab t{};
    int i = 0;
    for (int e = 0; e < 5; ++e) {
        const int l = 1 << e;
        for (int m = 0; m <= l; ++m, ++i)
            for (int k = 0; k <= l - m; ++k) t.c[k * IDE_N + i] = (float)cx_coeff(l, m, k);
    }
    return t;
}
__device__ constexpr IdeTab IDE_TAB = make_ide_tab();

inline bool ide_tab_equals(const float* mat /*[17 * IDE_N], k-major*/) {
    constexpr IdeTab tab = make_ide_tab();
    for (int q = 0; q < 17 * IDE_N; ++q)
        if (tab.c[q] != mat[q]) return false;
    return true;
}

// Compile-time walk over the 36 (l, m) pairs in table order (l = 1, 2, 4, 8, 16; m = 0..l): f(i, e, l, m) with all four as
// integral constants.  The loop form -- m = c_ide_m[i] read at run time -- indexes re[], im[], dre[] dynamically, which puts those
// arrays (and the caller's 72-entry gradient row) into scratch memory: shade_encode_bwd_kernel ran on 512 VGPRs with 186 spilled and
// 1.3 KB of private segment per thread.  Unrolled, every index is a constant and the arrays are registers.
template <int E, int M, int I, class F>
__device__ __forceinline__ void ide_pairs(F&& f) {
    constexpr int Lv = 1 << E;
    f(std::integral_constant<int, I>{}, std::integral_constant<int, E>{}, std::integral_constant<int, Lv>{}, std::integral_constant<int, M>{});
    if constexpr (M < Lv) ide_pairs<E, M + 1, I + 1>(f);
    else if constexpr (E < 4) ide_pairs<E + 1, 0, I + 1>(f);
}
__device__ __forceinline__ void ide_powers(float x, float y, float z, float (&zp)[17], float (&re)[17], float (&im)[17]) {
    zp[0] = 1.f; re[0] = 1.f; im[0] = 0.f;
#pragma unroll
    for (int k = 1; k <= 16; ++k) {
        zp[k] = zp[k - 1] * z;
        re[k] = re[k - 1] * x - im[k - 1] * y;
        im[k] = re[k - 1] * y + im[k - 1] * x;
    }
}

// out[0..36) = Re, out[36..72) = Im of (x+iy)^m * P_i(z) * exp(-l(l+1)/2 * kinv)
template <bool ATT>
__device__ __forceinline__ void ide_forward(float x, float y, float z, float kinv, float* __restrict__ out) {
    float zp[17], re[17], im[17], att[5];
    ide_powers(x, y, z, zp, re, im);
#pragma unroll
    for (int e = 0; e < 5; ++e) att[e] = ATT ? expf(-0.5f * (float)((1 << e) * ((1 << e) + 1)) * kinv) : 1.f;
    ide_pairs<0, 0, 0>([&](auto I, auto Ec, auto Lc, auto Mc) {
        constexpr int i = decltype(I)::value, e = decltype(Ec)::value, l = decltype(Lc)::value, m = decltype(Mc)::value;
        float poly = 0.f;
#pragma unroll
        for (int k = 0; k <= l - m; ++k) poly = fmaf(zp[k], IDE_TAB.c[k * IDE_N + i], poly);
        if (ATT) {
            out[i] = re[m] * poly * att[e];
            out[IDE_N + i] = im[m] * poly * att[e];
        } else {
            out[i] = re[m] * poly;
            out[IDE_N + i] = im[m] * poly;
        }
    });
}

// gradient of sum(g * ide(x,y,z,kinv)) w.r.t. (x,y,z,kinv); accumulates into dx,dy,dz,dk
// (g: callable, g(c) = the incoming gradient of channel c -- read where it is needed, not staged in a 72-entry array)
template <bool ATT, class G>
__device__ __forceinline__ void ide_backward(float x, float y, float z, float kinv, G&& g, float& dx, float& dy, float& dz, float& dk) {
    float zp[17], re[17], im[17], dre[17], dim_[17], att[5];
    ide_powers(x, y, z, zp, re, im);
#pragma unroll
    for (int k = 0; k <= 16; ++k) { dre[k] = 0.f; dim_[k] = 0.f; }
#pragma unroll
    for (int e = 0; e < 5; ++e) att[e] = ATT ? expf(-(0.5f * (float)((1 << e) * ((1 << e) + 1))) * kinv) : 1.f;
    float gz = 0.f, gk = 0.f;
    ide_pairs<0, 0, 0>([&](auto I, auto Ec, auto Lc, auto Mc) {
        constexpr int i = decltype(I)::value, e = decltype(Ec)::value, l = decltype(Lc)::value, m = decltype(Mc)::value;
        float poly = 0.f, dpoly = 0.f;
#pragma unroll
        for (int k = 0; k <= l - m; ++k) {
            const float c = IDE_TAB.c[k * IDE_N + i];
            poly = fmaf(zp[k], c, poly);
            if (k > 0) dpoly = fmaf((float)k * zp[k - 1], c, dpoly);
        }
        constexpr float sig = 0.5f * (float)(l * (l + 1));
        const float gr = g(i), gi = g(IDE_N + i);
        const float s = gr * re[m] + gi * im[m];
        if (ATT) {
            gz += s * att[e] * dpoly;
            gk += s * poly * (-sig * att[e]);
            dre[m] += gr * poly * att[e];
            dim_[m] += gi * poly * att[e];
        } else {                                           // kappa_inv = 0 (Stage II, network/field.py:817,838): no attenuation, no d/dk
            gz += s * dpoly;
            dre[m] += gr * poly;
            dim_[m] += gi * poly;
        }
    });
    // w^m = re + i im:  d re_m/dx = m re_{m-1}, d im_m/dx = m im_{m-1}, d re_m/dy = -m im_{m-1}, d im_m/dy = m re_{m-1}
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int m = 1; m <= 16; ++m) {
        gx += (float)m * (dre[m] * re[m - 1] + dim_[m] * im[m - 1]);
        gy += (float)m * (-dre[m] * im[m - 1] + dim_[m] * re[m - 1]);
    }
    dx += gx; dy += gy; dz += gz; dk += gk;
}


