// phase_shift_probe.hip -- the question DESIGN.md 9.5 ends with: if the two waves of a SIMD ran DIFFERENT phases of the chain walk (one in
// its GEMM, the other in its epilogue), would the matrix pipe and the VALU overlap -- and can ONE wave per SIMD keep the pipe fed?
//   MODE 0  "lock step" (today's structure): 8 waves own one 64-row tile; per layer every wave multiplies ONE 32-feature tile
//           (16 k-steps x [2 weight loads (L2) + 4 fragment reads (LDS) + 6 MFMAs]), then runs its epilogue (~450 VALU + 16 LDS stores),
//           two barriers.
//   MODE 1  "phase shifted": the workgroup owns TWO tiles; waves 0-3 own tile A, waves 4-7 tile B, each wave TWO feature tiles
//           (16 k-steps x [4 weight loads + 4 fragment reads + 12 MFMAs], then ~900 VALU + 32 LDS stores).  Every wave runs the same
//           sequence GEMM, barrier, epilogue, barrier -- group B one half-step late -- so that in every half-step one wave of a SIMD is in
//           its GEMM and the other in its epilogue.
//   MODE 4  "one barrier": MODE 0 with the planes DOUBLE BUFFERED (layer l reads buffer l & 1, its epilogue writes the other) and no
//           barrier between GEMM and epilogue -- what a block scale known before the GEMM (no row-maximum exchange) would allow.
//   MODE 2  MODE 1's GEMM alone (no epilogue work): the lone-wave GEMM rate.      MODE 3  MODE 0's GEMM alone.
// Same arithmetic per tile-layer in all modes (768 MFMAs of v_mfma_f32_32x32x16_f16 per tile and layer: three plane products).
// hipcc --offload-arch=gfx950 -O3 -o phase_shift_probe phase_shift_probe.hip && ./phase_shift_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SA = 528, PLANE = 64 * SA, TILE = 2 * PLANE;       // two fp16 planes of a 64 x 256 tile
#define MF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
struct WF { uint4 wh, wl; };
struct XF { uint4 xh0, xl0, xh1, xl1; };
__device__ __forceinline__ void load_w(WF& o, const uint4* wp, int c) { const uint4* w = wp + (size_t)c * 128; o.wh = w[0]; o.wl = w[64]; }
__device__ __forceinline__ void load_x(XF& o, const char* xp, int c) {
    const char* x = xp + c * 32;
    o.xh0 = *(const uint4*)x; o.xl0 = *(const uint4*)(x + PLANE);
    x += 32 * SA;
    o.xh1 = *(const uint4*)x; o.xl1 = *(const uint4*)(x + PLANE);
}
__device__ __forceinline__ void mm(f32x16 (&aH)[2], f32x16 (&aL)[2], const WF& w, const XF& x) {
    MF(aL[0], w.wl, x.xh0); MF(aL[1], w.wl, x.xh1); MF(aH[0], w.wh, x.xh0); MF(aH[1], w.wh, x.xh1); MF(aL[0], w.wh, x.xl0); MF(aL[1], w.wh, x.xl1);
}
// NF feature tiles per wave, weights three steps ahead in a ring of four, fragments one step ahead
template <int NF>
__device__ __forceinline__ void gemm(f32x16 (&aH)[NF][2], f32x16 (&aL)[NF][2], const uint4* wp, size_t ft_stride, const char* xp) {
    WF w[4][NF];
    XF x[2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int f = 0; f < NF; ++f) load_w(w[c][f], wp + f * ft_stride, c);
    load_x(x[0], xp, 0);
    FENCE();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c + 3 < 16) {
#pragma unroll
            for (int f = 0; f < NF; ++f) load_w(w[(c + 3) & 3][f], wp + f * ft_stride, c + 3);
        }
        if (c + 1 < 16) load_x(x[(c + 1) & 1], xp, c + 1);
        FENCE();
#pragma unroll
        for (int f = 0; f < NF; ++f) mm(aH[f], aL[f], w[c & 3][f], x[c & 1]);
        FENCE();
    }
}
// stand-in for the epilogue of one feature tile: combine, bias, ReLU, row maximum, scale, split into two fp16 planes, 16 plane stores
__device__ __forceinline__ float epilogue(const f32x16 (&aH)[2], const f32x16 (&aL)[2], char* dst, float bias) {
    float m = 0.f;
    float v[2][16];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[r][k] = fmaxf(fmaf(fmaf(aL[r][k], 4.8828125e-4f, aH[r][k]), 1.25f, bias), 0.f); m = fmaxf(m, v[r][k]); }
    m = fmaxf(m, __shfl_xor(m, 32));
    const float inv = 1.0f / (m + 1.0f);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            unsigned hp[2], lp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = v[r][k + 2 * j] * inv, b = v[r][k + 2 * j + 1] * inv;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 ab = {a, b};
                h2 hh = __builtin_convertvector(ab, h2);
                f2 rem = {(a - (float)hh[0]) * 2048.f, (b - (float)hh[1]) * 2048.f};
                h2 ll = __builtin_convertvector(rem, h2);
                hp[j] = __builtin_bit_cast(unsigned, hh); lp[j] = __builtin_bit_cast(unsigned, ll);
            }
            *(uint2*)(dst + r * 32 * SA + k * 4) = make_uint2(hp[0], hp[1]);
            *(uint2*)(dst + PLANE + r * 32 * SA + k * 4) = make_uint2(lp[0], lp[1]);
        }
    return m;
}
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const uint4* W, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NF = (MODE == 1 || MODE == 2) ? 2 : 1;
    constexpr bool EPI = MODE <= 1 || MODE == 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 31, h = lane >> 5;
    for (int idx = tid; idx < 2 * TILE / 16; idx += 512) {          // pseudo-random fp16 planes (realistic toggling)
        unsigned s0 = (blockIdx.x * 9781u + idx) * 2654435761u;
        uint4 q; unsigned* qq = (unsigned*)&q;
        for (int j = 0; j < 4; ++j) { s0 = s0 * 1664525u + 1013904223u; qq[j] = (s0 & 0x83ff83ffu) | 0x38003800u; }
        ((uint4*)smem)[idx] = q;
    }
    __syncthreads();
    const int group = NF == 2 ? (wave >> 2) : 0;                   // tile of this wave
    const int ft0 = NF == 2 ? 2 * (wave & 3) : wave;               // first feature tile of this wave
    const char* xp = smem + group * TILE + i * SA + 16 * h;
    char* dst = smem + group * TILE + i * SA + (32 * ft0 + 4 * h) * 2;
    float s = 0.f;
    if (NF == 2 && group == 1) __syncthreads();                    // group B runs one half-step behind
    for (int l = 0; l < layers; ++l) {
        f32x16 aH[NF][2], aL[NF][2];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) { aH[f][r][v] = 0.f; aL[f][r][v] = 0.f; }
        const uint4* wp = W + ((size_t)(l & 7) * 8 + ft0) * 16 * 128 + lane;
        if (MODE == 4) {
            gemm<NF>(aH, aL, wp, (size_t)16 * 128, xp + (l & 1) * TILE);
            s += epilogue(aH[0], aL[0], dst + ((l + 1) & 1) * TILE, 0.01f * l);
            __syncthreads();
            continue;
        }
        gemm<NF>(aH, aL, wp, (size_t)16 * 128, xp);
        __syncthreads();                                           // (lock step: every wave is done reading the planes)
        if (EPI) {
#pragma unroll
            for (int f = 0; f < NF; ++f) s += epilogue(aH[f], aL[f], dst + f * 64, 0.01f * l);
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) s += aH[f][0][0] + aL[f][1][3];
        }
        __syncthreads();
    }
    if (NF == 2 && group == 0) __syncthreads();
    if (s == 12345.678f) out[tid] = s;
}
static int g_wgs = 1024;                         // workgroups (argv[1]): 1024 = four rounds over all 256 CUs (the loaded chip), 32 = one workgroup on 32 CUs (a cold chip)
template <int MODE>
void run(const char* name, const uint4* W, float* out) {
    const int layers = 64, grid = g_wgs, lds = 2 * TILE;         // 135 KB: one workgroup per CU in every mode
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<grid, 512, lds>>>(W, out, layers);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < 3; ++it) probe<MODE><<<grid, 512, lds>>>(W, out, layers);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double tile_layers = (double)grid * layers * ((MODE == 1 || MODE == 2) ? 2 : 1);
    const double rounds = (g_wgs + 255) / 256;                       // workgroups a CU runs one after the other
    const double us = ms * 1e3 / (rounds * layers * ((MODE == 1 || MODE == 2) ? 2 : 1));
    printf("%-52s %8.3f ms  %6.2f us per tile-layer per CU  %6.1f TFLOP/s fp32-equivalent (x3 issued)\n", name, ms, us,
           tile_layers * 64.0 * 256 * 256 * 2 / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    if (argc > 1) g_wgs = atoi(argv[1]);
    printf("== %d workgroups (one per CU at a time)\n", g_wgs);
    uint4* W; float* out;
    const size_t wn = (size_t)8 * 8 * 16 * 128 * 4;      // dwords: 8 layers x 8 feature tiles x 16 k-steps x (64 lanes x 2 planes) uint4
    hipMalloc(&W, wn * 4);
    {
        std::vector<unsigned> hw(wn);
        unsigned s0 = 12345u;
        for (size_t j = 0; j < wn; ++j) { s0 = s0 * 1664525u + 1013904223u; hw[j] = (s0 & 0x83ff83ffu) | 0x34003400u; }
        hipMemcpy(W, hw.data(), wn * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 4096);
    run<3>("lock step, GEMM only (8 waves x 1 feature tile)", W, out);
    run<2>("two tiles, GEMM only (4 + 4 waves x 2 feature tiles)", W, out);
    run<0>("lock step: GEMM, epilogue, 2 barriers (today)", W, out);
    run<1>("phase shifted: GEMM || epilogue of the other tile", W, out);
    run<4>("one barrier per layer, planes double buffered", W, out);
    return 0;
}
