// probe of the split GEMM inner loop on gfx950: which operand stream stalls the MFMA pipe?  (tuning aid, not part of the library)
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/probe scripts/probe/mfma_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SA = 528, PLANE_A = 64 * SA;
struct OpS { uint4 w0, w1, w2, x00, x01, x02, x10, x11, x12; };
#define MF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

template <int MODE>   // bit0: load W from global, bit1: load X from LDS
__device__ __forceinline__ void ops_load(OpS& o, const uint4* wp, const char* xp, int c) {
    if (MODE & 1) { const uint4* w = wp + (size_t)c * 192; o.w0 = w[0]; o.w1 = w[64]; o.w2 = w[128]; }
    if (MODE & 2) {
        const char* x = xp + c * 32;
        o.x00 = *(const uint4*)(x); o.x01 = *(const uint4*)(x + PLANE_A); o.x02 = *(const uint4*)(x + 2 * PLANE_A);
        x += 32 * SA;
        o.x10 = *(const uint4*)(x); o.x11 = *(const uint4*)(x + PLANE_A); o.x12 = *(const uint4*)(x + 2 * PLANE_A);
    }
}
__device__ __forceinline__ void ops_compute(f32x16 (&acc)[2], const OpS& o) {
    MF(acc[0], o.w2, o.x00); MF(acc[1], o.w2, o.x10); MF(acc[0], o.w1, o.x01); MF(acc[1], o.w1, o.x11);
    MF(acc[0], o.w0, o.x02); MF(acc[1], o.w0, o.x12); MF(acc[0], o.w1, o.x00); MF(acc[1], o.w1, o.x10);
    MF(acc[0], o.w0, o.x01); MF(acc[1], o.w0, o.x11); MF(acc[0], o.w0, o.x00); MF(acc[1], o.w0, o.x10);
}
template <int MODE, bool BARRIER>
__global__ __launch_bounds__(512, 1) void probe(const uint4* W, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    // pseudo-random bf16 values in [-2, 2): realistic toggling (DVFS clocks constant data higher)
    for (int idx = tid; idx < 3 * PLANE_A / 16; idx += 512) {
        unsigned s0 = (blockIdx.x * 9781u + idx) * 2654435761u;
        uint4 q; unsigned* qq = (unsigned*)&q;
        for (int j = 0; j < 4; ++j) { s0 = s0 * 1664525u + 1013904223u; qq[j] = (s0 & 0x807f807fu) | 0x3f003f00u | ((s0 >> 9) & 0x00800080u); }
        ((uint4*)smem)[idx] = q;
    }
    __syncthreads();
    f32x16 acc[2];
    for (int r = 0; r < 2; ++r) for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
    OpS u, v, w;
    {
        unsigned s0 = (tid * 7919u + 13u) * 2654435761u;
        unsigned* p = (unsigned*)&u;
        for (int j = 0; j < 36; ++j) { s0 = s0 * 1664525u + 1013904223u; p[j] = (s0 & 0x807f807fu) | 0x3f003f00u; }
    }
    v = u; w = u;
    const char* xp = smem + i * SA + 16 * h;
    for (int l = 0; l < layers; ++l) {
        const uint4* wp = W + ((size_t)l * 8 + wave) * 16 * 192 + lane;
        const int n = 16, last = 15;
        ops_load<MODE>(u, wp, xp, 0); ops_load<MODE>(v, wp, xp, 1);
        __builtin_amdgcn_sched_barrier(0);
        int c = 0;
        for (; c + 3 <= n; c += 3) {
            ops_load<MODE>(w, wp, xp, c + 2 < last ? c + 2 : last); __builtin_amdgcn_sched_barrier(0);
            ops_compute(acc, u); __builtin_amdgcn_sched_barrier(0);
            ops_load<MODE>(u, wp, xp, c + 3 < last ? c + 3 : last); __builtin_amdgcn_sched_barrier(0);
            ops_compute(acc, v); __builtin_amdgcn_sched_barrier(0);
            ops_load<MODE>(v, wp, xp, c + 4 < last ? c + 4 : last); __builtin_amdgcn_sched_barrier(0);
            ops_compute(acc, w); __builtin_amdgcn_sched_barrier(0);
        }
        if (c < n) ops_compute(acc, u);
        if (BARRIER) { __syncthreads(); __syncthreads(); }
    }
    float s = 0.f;
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 16; ++q) s += acc[r][q];
    if (s == 12345.678f) out[tid] = s;
}
template <int MODE, bool BARRIER>
void run(const char* name, const uint4* W, float* out) {
    const int layers = 8, grid = 256 * 32, lds = 3 * PLANE_A + 3 * 64 * 112;
    hipFuncSetAttribute((const void*)probe<MODE, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE, BARRIER><<<grid, 512, lds>>>(W, out, layers);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < 3; ++it) probe<MODE, BARRIER><<<grid, 512, lds>>>(W, out, layers);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double tiles = (double)grid * layers;                       // layer-tiles
    const double cyc = ms * 1e-3 * 2.4e9 / (tiles / 256.0);          // nominal cycles per layer-tile per CU
    printf("%-44s %7.3f ms  %6.0f nominal cyc / layer-tile (MFMA ideal 12288)  %6.1f TF-equiv\n", name, ms, cyc,
           tiles * 64.0 * 256 * 256 * 2 / (ms * 1e-3) / 1e12);
}
int main() {
    uint4* W; float* out;
    const size_t wn = (size_t)8 * 8 * 16 * 192 * 4;      // dwords
    hipMalloc(&W, wn * 4);
    {
        std::vector<unsigned> hw(wn);
        unsigned s0 = 12345u;
        for (size_t j = 0; j < wn; ++j) { s0 = s0 * 1664525u + 1013904223u; hw[j] = (s0 & 0x807f807fu) | 0x3c003c00u; }
        hipMemcpy(W, hw.data(), wn * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 4096);
    run<0, false>("pure MFMA, no barriers", W, out);
    run<0, true>("pure MFMA, layer barriers", W, out);
    run<2, false>("X from LDS, no barriers", W, out);
    run<1, false>("W from L2, no barriers", W, out);
    run<3, false>("W + X, no barriers", W, out);
    run<3, true>("W + X, layer barriers (product loop)", W, out);
    return 0;
}
