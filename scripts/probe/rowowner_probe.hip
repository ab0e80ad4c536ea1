// rowowner_probe.hip -- round 6: can ONE wave per SIMD keep the matrix pipe fed when the chain walk is re-cut so that a wave owns ROWS?
//
// Today (mlp_f16x3.hip; MODE 0 below = phase_shift_probe's MODE 0): 8 waves share a 64-row tile and split the FEATURES of a layer, so every
// layer ends in a row-maximum exchange and two workgroup barriers, the epilogue runs with an idle matrix pipe, and the activation planes
// travel through LDS (4 fragment reads per 6 MFMAs).
//
// MODE R (the candidate): a 256-thread workgroup = 4 waves, ONE per SIMD, 512 registers each.  A wave owns 32 rows and ALL 256 features:
//   * the activation planes (B operand) of its rows live in REGISTERS (16 k-steps x (h, l) x 4 VGPRs = 128), built by the epilogue from
//     the accumulator layout with one v_permlane32_swap per two VGPRs -- no LDS round trip, no barrier, the row maximum is lane-local
//     (+ one cross-half exchange);
//   * the weight image (A operand) streams L2 -> LDS by LDS-DMA into a ring of four 32 KB slots (one feature tile = 16 k-steps x 2 planes
//     x 1 KB), each wave requesting a quarter of every slot: ONE copy per CU serves 128 rows (today: one per 64 rows);
//   * the GEMM runs TILE-major (48 dependent-accumulator MFMAs per feature tile), and the VALU epilogue of tile t - 1 (bias, activation,
//     maximum: one element per k-step) is issued between the MFMAs of tile t -- the "<= 5 fillers per MFMA gap" regime of
//     MI355X_MICROARCH.md; one s_barrier per feature tile keeps the four waves inside the ring window.
//   EXACT: the block scale is the exact row maximum (bit-compatible with today's engine): the conversion of the 128 values per lane into
//   the next layer's planes waits for the last tile and is exposed.  LAZY: the scale is known before the GEMM (a bound), the conversion
//   runs per tile under the next tile's MFMAs as well.
// Reported: microseconds per 64-row tile-layer per CU (the unit of phase_shift_probe / DESIGN.md 9.1) and fp32-equivalent TFLOP/s.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o rowowner_probe rowowner_probe.hip && ./rowowner_probe [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ unsigned long long g_cyc[2];          // sum of the shader-clock residence of every workgroup's wave 0, workgroups
#define CYC_BEGIN const long long cyc_t0 = clock64()
#define CYC_END do { if (threadIdx.x == 0) { atomicAdd(&g_cyc[0], (unsigned long long)(clock64() - cyc_t0)); atomicAdd(&g_cyc[1], 1ull); } } while (0)

// ---------------------------------------------------------------- MODE 0 / 3: today's structure (copied from phase_shift_probe.hip)
constexpr int SA = 528, PLANE = 64 * SA, TILE = 2 * PLANE;
struct WF { uint4 wh, wl; };
struct XF { uint4 xh0, xl0, xh1, xl1; };
__device__ __forceinline__ void load_w(WF& o, const uint4* wp, int c) { const uint4* w = wp + (size_t)c * 128; o.wh = w[0]; o.wl = w[64]; }
__device__ __forceinline__ void load_x(XF& o, const char* xp, int c) {
    const char* x = xp + c * 32;
    o.xh0 = *(const uint4*)x; o.xl0 = *(const uint4*)(x + PLANE);
    x += 32 * SA;
    o.xh1 = *(const uint4*)x; o.xl1 = *(const uint4*)(x + PLANE);
}
__device__ __forceinline__ void mm(f32x16 (&a)[2], const WF& w, const XF& x) {
    MF(a[0], w.wl, x.xh0); MF(a[1], w.wl, x.xh1); MF(a[0], w.wh, x.xl0); MF(a[1], w.wh, x.xl1); MF(a[0], w.wh, x.xh0); MF(a[1], w.wh, x.xh1);
}
__device__ __forceinline__ void gemm_old(f32x16 (&a)[2], const uint4* wp, const char* xp) {
    WF w[4];
    XF x[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) load_w(w[c], wp, c);
    load_x(x[0], xp, 0);
    FENCE();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c + 3 < 16) load_w(w[(c + 3) & 3], wp, c + 3);
        if (c + 1 < 16) load_x(x[(c + 1) & 1], xp, c + 1);
        FENCE();
        mm(a, w[c & 3], x[c & 1]);
        FENCE();
    }
}
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split2(float a, float b, unsigned& hp, unsigned& lp) {
    hp = pk_f16(a, b);
    const f16x2 hh = __builtin_bit_cast(f16x2, hp);
    lp = pk_f16(a - (float)hh[0], b - (float)hh[1]);
}
template <int ACT> __device__ __forceinline__ float act(float x) {
    if (ACT == 0) return fmaxf(x, 0.f);
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 144.26950408889634f);            // softplus(beta = 100), as mlp_f16_util.h
    return fmaf(__builtin_amdgcn_logf(1.0f + t), 0.0069314718055994531f, fmaxf(x, 0.f));
}
template <int ACT>
__device__ __forceinline__ float epilogue_old(const f32x16 (&a)[2], char* dst, float* rmax, int wave, int i, int h, float bias) {
    float m[2] = {0.f, 0.f};
    float v[2][16];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[r][k] = act<ACT>(fmaf(a[r][k], 1.25f, bias)); m[r] = fmaxf(m[r], fabsf(v[r][k])); }
    m[0] = fmaxf(m[0], __shfl_xor(m[0], 32)); m[1] = fmaxf(m[1], __shfl_xor(m[1], 32));
    if (h == 0) { rmax[i * 8 + wave] = m[0]; rmax[(32 + i) * 8 + wave] = m[1]; }
    __syncthreads();
    float mm_[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float4 p = *(const float4*)(rmax + (32 * r + i) * 8), q = *(const float4*)(rmax + (32 * r + i) * 8 + 4);
        mm_[r] = fmaxf(fmaxf(fmaxf(p.x, p.y), fmaxf(p.z, p.w)), fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float inv = __uint_as_float((unsigned)(127 + 14 + 127 - (int)((__float_as_uint(mm_[r]) >> 23) & 0xff)) << 23);
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            unsigned hp[2], lp[2];
            split2(v[r][k] * inv, v[r][k + 1] * inv, hp[0], lp[0]);
            split2(v[r][k + 2] * inv, v[r][k + 3] * inv, hp[1], lp[1]);
            *(uint2*)(dst + r * 32 * SA + k * 4) = make_uint2(hp[0], hp[1]);
            *(uint2*)(dst + PLANE + r * 32 * SA + k * 4) = make_uint2(lp[0], lp[1]);
        }
    }
    return m[0];
}
template <int ACT, bool EPI>
__global__ __launch_bounds__(512, 1) void probe_old(const uint4* W, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 31, h = lane >> 5;
    float* rmax = (float*)(smem + TILE);
    CYC_BEGIN;
    for (int idx = tid; idx < TILE / 16; idx += 512) {
        unsigned s0 = (blockIdx.x * 9781u + idx) * 2654435761u;
        uint4 q; unsigned* qq = (unsigned*)&q;
        for (int j = 0; j < 4; ++j) { s0 = s0 * 1664525u + 1013904223u; qq[j] = (s0 & 0x83ff83ffu) | 0x38003800u; }
        ((uint4*)smem)[idx] = q;
    }
    __syncthreads();
    const char* xp = smem + i * SA + 16 * h;
    char* dst = smem + i * SA + (32 * wave + 4 * h) * 2;
    float s = 0.f;
    for (int l = 0; l < layers; ++l) {
        f32x16 a[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) a[r][v] = 0.f;
        const uint4* wp = W + ((size_t)(l & 7) * 8 + wave) * 16 * 128 + lane;
        gemm_old(a, wp, xp);
        if (EPI) s += epilogue_old<ACT>(a, dst, rmax, wave, i, h, 0.01f * l);
        else { s += a[0][0] + a[1][3]; __syncthreads(); }
        __syncthreads();
    }
    CYC_END;
    if (s == 12345.678f) out[tid] = s;
}

// ---------------------------------------------------------------- MODE R: a wave owns 32 rows
constexpr int RING = 4, SLOT = 32768;
__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "{m0}"(lds_addr) : "memory");
}
// the 16 values of a finished feature tile (accumulator layout: v = 4 g + j <-> feature 8 g + 4 h + j of row i) -> the (h, l) plane
// fragments of the next layer's k-steps 2t, 2t + 1 (B layout: lane (i, h') holds features 16 c + 8 h' + 0..7)
__device__ __forceinline__ void to_planes(const float (&v)[16], float inv, uint4& xh0, uint4& xl0, uint4& xh1, uint4& xl1) {
    unsigned hp[8], lp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split2(v[2 * k] * inv, v[2 * k + 1] * inv, hp[k], lp[k]);
    // g = 0 (hp[0..1]) / g = 1 (hp[2..3]) -> k-step 2t;  g = 2 (hp[4..5]) / g = 3 (hp[6..7]) -> k-step 2t + 1
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        unsigned e0 = hp[4 * gp], e1 = hp[4 * gp + 1], o0 = hp[4 * gp + 2], o1 = hp[4 * gp + 3];
        unsigned f0 = lp[4 * gp], f1 = lp[4 * gp + 1], p0 = lp[4 * gp + 2], p1 = lp[4 * gp + 3];
        auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        auto s2 = __builtin_amdgcn_permlane32_swap(f0, p0, false, false);
        auto s3 = __builtin_amdgcn_permlane32_swap(f1, p1, false, false);
        const uint4 H = make_uint4(s0[0], s1[0], s0[1], s1[1]), L = make_uint4(s2[0], s3[0], s2[1], s3[1]);
        if (gp == 0) { xh0 = H; xl0 = L; } else { xh1 = H; xl1 = L; }
    }
}
// NT feature tiles at once (NT independent accumulators: a dependent MFMA on the SAME accumulator does not issue back to back at full
// rate).  The ring is filled in CONSUMPTION order, 32 KB = 32 chunks per stage: stage g of a layer covers the k-steps
// [KS * (g % NT), + KS) of the tile group g / NT, KS = 16 / NT; chunk j = ((k-step, tile of the group), plane).
template <int ACT, bool LAZY, int NT>
__global__ __launch_bounds__(256, 1) void probe_r(const char* W, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 16 / NT, SPG = 16 / KS;                   // k-steps per stage, stages per tile group
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 xh[16], xl[16];
    {
        unsigned s0 = (blockIdx.x * 9781u + tid) * 2654435761u;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            unsigned* a = (unsigned*)&xh[c]; unsigned* b = (unsigned*)&xl[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = s0 * 1664525u + 1013904223u; a[j] = (s0 & 0x83ff83ffu) | 0x38003800u;
                s0 = s0 * 1664525u + 1013904223u; b[j] = (s0 & 0x83ff83ffu) | 0x28002800u;
            }
        }
    }
    const unsigned ring0 = (unsigned)(size_t)smem;
    CYC_BEGIN;
    auto issue_one = [&](int l, int g, int jj) {              // chunk 8 wave + jj of stage g (0..7, may run past the layer) of layer l
        l += g >> 3; g &= 7;
        const int j = wave * 8 + jj;
        const int c = KS * (g % SPG) + j / (2 * NT), tile = NT * (g / SPG) + (j >> 1) % NT, plane = j & 1;
        const char* src = W + (size_t)(l & 7) * 262144 + (size_t)((tile * 16 + c) * 2 + plane) * 1024 + lane * 16;
        lds_dma16(src, ring0 + ((l * 8 + g) & (RING - 1)) * SLOT + j * 1024);
    };
#pragma unroll
    for (int g = 0; g < RING - 1; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) issue_one(0, g, j);
    float s = 0.f;
    float inv_lazy = 1.f / 64.f;
    for (int l = 0; l < layers; ++l) {
        float vals[8][16];
        uint4 nh[16], nl[16];
        float m = 0.f;
        const float bias = 0.01f * l;
        f32x16 acc[NT];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int grp = g / SPG, c0 = KS * (g % SPG);
            // this wave's quarter of the stage has landed (two younger request groups may still be in flight); then everybody's has,
            // and everybody is done reading the slot of the previous stage, which the requests of stage g + 3 overwrite
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // (the requests run three stages past the end: the image index wraps)
            __builtin_amdgcn_s_barrier();
            const char* slot = smem + ((l * 8 + g) & (RING - 1)) * SLOT + lane * 16;
            if (c0 == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[n][v] = 0.f;
            }
            uint4 w[3][NT][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int n = 0; n < NT; ++n) { w[p][n][0] = *(const uint4*)(slot + ((p * NT + n) * 2) * 1024); w[p][n][1] = *(const uint4*)(slot + ((p * NT + n) * 2 + 1) * 1024); }
            FENCE();
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int c = c0 + k;
                if (k + 2 < KS) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        w[(k + 2) % 3][n][0] = *(const uint4*)(slot + (((k + 2) * NT + n) * 2) * 1024);
                        w[(k + 2) % 3][n][1] = *(const uint4*)(slot + (((k + 2) * NT + n) * 2 + 1) * 1024);
                    }
                }
                if ((k * 8) % KS == 0) {
#pragma unroll
                    for (int jj = 0; jj < (8 + KS - 1) / KS; ++jj)
                        if (k * 8 / KS + jj < 8) issue_one(l, g + RING - 1, k * 8 / KS + jj);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) MF(acc[n], w[k % 3][n][1], xh[c]);
#pragma unroll
                for (int n = 0; n < NT; ++n) MF(acc[n], w[k % 3][n][0], xl[c]);
#pragma unroll
                for (int n = 0; n < NT; ++n) MF(acc[n], w[k % 3][n][0], xh[c]);
                if (grp > 0) {                                    // NT elements of the previous tile group's epilogue per k-step
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int tp = NT * (grp - 1) + n;
                        const float y = act<ACT>(fmaf(vals[tp][c], 1.25f, bias));
                        vals[tp][c] = y;
                        m = fmaxf(m, fabsf(y));
                    }
                }
                FENCE();
            }
            if (c0 + KS == 16) {
                if (LAZY && grp > 0) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int tp = NT * (grp - 1) + n;
                        to_planes(vals[tp], inv_lazy, nh[2 * tp], nl[2 * tp], nh[2 * tp + 1], nl[2 * tp + 1]);
                    }
                    FENCE();
                }
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int v = 0; v < 16; ++v) vals[NT * grp + n][v] = acc[n][v];
            }
        }
#pragma unroll
        for (int tp = 8 - NT; tp < 8; ++tp)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float y = act<ACT>(fmaf(vals[tp][c], 1.25f, bias));
                vals[tp][c] = y;
                m = fmaxf(m, fabsf(y));
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float inv = __uint_as_float((unsigned)(127 + 14 + 127 - (int)((__float_as_uint(m) >> 23) & 0xff)) << 23);
        if (LAZY) {
#pragma unroll
            for (int tp = 8 - NT; tp < 8; ++tp) to_planes(vals[tp], inv_lazy, nh[2 * tp], nl[2 * tp], nh[2 * tp + 1], nl[2 * tp + 1]);
            inv_lazy = inv;                                   // (stand-in for the bound of the next layer: known before its GEMM)
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) to_planes(vals[t], inv, nh[2 * t], nl[2 * t], nh[2 * t + 1], nl[2 * t + 1]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) { xh[c] = nh[c]; xl[c] = nl[c]; }
        s += m;
        FENCE();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CYC_END;
    if (s == 12345.678f) out[tid] = s + __uint_as_float(xh[3].x);
}

static int g_wgs = 1024;
static void report(const char* name, float ms, int rows_per_wg, int layers) {
    const double rounds = (g_wgs + 255) / 256;
    const double us = ms * 1e3 / (rounds * layers * (rows_per_wg / 64));
    unsigned long long cyc[2] = {0, 0}, z[2] = {0, 0};
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
    hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
    const double cyc_tl = (double)cyc[0] / cyc[1] / (layers * (rows_per_wg / 64));      // shader-clock cycles per 64-row tile-layer
    printf("%-62s %7.3f ms %6.2f us / 64-row tile-layer / CU %6.1f TF  %6.0f cyc -> %.2f GHz\n", name, ms, us,
           (double)g_wgs * layers * rows_per_wg * 256.0 * 256 * 2 / (ms * 1e-3) / 1e12, cyc_tl, cyc_tl / us * 1e-3);
}
template <class K, class WP> static float time_kernel(K k, int block, int lds, WP W, float* out, int layers) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(g_wgs), dim3(block), lds, 0, W, out, layers);
    hipDeviceSynchronize();
    hipEventRecord(a);
    { unsigned long long z[2] = {0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z)); }
    hipEventRecord(a);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(g_wgs), dim3(block), lds, 0, W, out, layers);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
    return ms / 3;
}
int main(int argc, char** argv) {
    if (argc > 1) g_wgs = atoi(argv[1]);
    printf("== %d workgroups (one per CU at a time)\n", g_wgs);
    char* W; float* out;
    const size_t wn = (size_t)8 * 8 * 16 * 128 * 4;      // dwords: 8 layers x 8 feature tiles x 16 k-steps x (64 lanes x 2 planes) uint4
    hipMalloc(&W, wn * 4);
    {
        std::vector<unsigned> hw(wn);
        unsigned s0 = 12345u;
        for (size_t j = 0; j < wn; ++j) { s0 = s0 * 1664525u + 1013904223u; hw[j] = (s0 & 0x83ff83ffu) | 0x34003400u; }
        hipMemcpy(W, hw.data(), wn * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 4096);
    const int layers = 64, lds_old = TILE + 64 * 8 * 4, lds_r = RING * SLOT;
    report("today, ONE workgroup per CU: GEMM only", time_kernel(probe_old<0, false>, 512, 2 * TILE, (const uint4*)W, out, layers), 64, layers);
    report("today, ONE workgroup per CU: GEMM + relu epilogue", time_kernel(probe_old<0, true>, 512, 2 * TILE, (const uint4*)W, out, layers), 64, layers);
    report("today, ONE workgroup per CU: GEMM + softplus epilogue", time_kernel(probe_old<1, true>, 512, 2 * TILE, (const uint4*)W, out, layers), 64, layers);
    report("today, TWO workgroups per CU: GEMM only", time_kernel(probe_old<0, false>, 512, lds_old, (const uint4*)W, out, layers), 64, layers);
    report("today, TWO workgroups per CU: GEMM + relu epilogue", time_kernel(probe_old<0, true>, 512, lds_old, (const uint4*)W, out, layers), 64, layers);
    report("today, TWO workgroups per CU: GEMM + softplus epilogue", time_kernel(probe_old<1, true>, 512, lds_old, (const uint4*)W, out, layers), 64, layers);
#define RUN_R(ACT, LAZY, NT, name) report(name, time_kernel(probe_r<ACT, LAZY, NT>, 256, lds_r, (const char*)W, out, layers), 128, layers)
    RUN_R(0, false, 1, "row owner, 1 tile at once : relu, exact row maximum");
    RUN_R(0, true, 1, "row owner, 1 tile at once : relu, lazy scale");
    RUN_R(0, false, 2, "row owner, 2 tiles at once: relu, exact row maximum");
    RUN_R(0, true, 2, "row owner, 2 tiles at once: relu, lazy scale");
    RUN_R(1, false, 2, "row owner, 2 tiles at once: softplus, exact row maximum");
    RUN_R(1, true, 2, "row owner, 2 tiles at once: softplus, lazy scale");
    RUN_R(0, false, 4, "row owner, 4 tiles at once: relu, exact row maximum");
    RUN_R(0, true, 4, "row owner, 4 tiles at once: relu, lazy scale");
    RUN_R(1, false, 4, "row owner, 4 tiles at once: softplus, exact row maximum");
    RUN_R(1, true, 4, "row owner, 4 tiles at once: softplus, lazy scale");
    return 0;
}
