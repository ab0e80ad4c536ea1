// read_pattern_probe.hip -- does the SHAPE of a wave's 1 KB read matter for the HBM-bound chain passes?
// The reverse / tangent kernels fetch saved activations "in the lane's fragment order": a wave instruction reads 32 rows x 32 B
// (lane (i, h): 16 B at row i, column 4 h + 8 g), four instructions cover the wave's 32 rows x 128 B.  The alternative is row-major:
// an instruction reads 8 rows x 128 B (lane: row l >> 3, piece l & 7).  Both move the same bytes; this probe streams a [rows][256] fp32
// matrix tile by tile (64 rows per workgroup step, 8 waves x 32 columns, as the kernels do) with either pattern, as register loads and
// as LDS-DMA, and prints GB/s.  One workgroup per CU walking tiles (persistent) or one tile per workgroup.
// hipcc --offload-arch=gfx950 -O3 -o read_pattern_probe read_pattern_probe.hip && ./read_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "{m0}"(lds_addr) : "memory");
}

// MODE 0: fragment-order register loads   1: row-major register loads   2: fragment-order LDS-DMA   3: row-major LDS-DMA
// `work`: dummy FMA iterations between tiles (stands for the GEMM + epilogue of a layer-tile, so that loads of the NEXT tile are in flight
// behind compute as in the kernels)
template <int MODE>
__global__ __launch_bounds__(512, 1) void read_kernel(const float* __restrict__ src, int n_tiles, int layers, size_t layer_stride, int work, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    float acc = 0.f;
    char* pa = smem + wave * 8192;
    const unsigned pa_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pa);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int l = 0; l < layers; ++l) {
            const float* base = src + (size_t)l * layer_stride + (size_t)tile * 64 * 256 + 32 * wave;
            float4 v[8];
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[r * 4 + g] = *reinterpret_cast<const float4*>(base + (size_t)(32 * r + i) * 256 + 4 * h + 8 * g);
            } else if (MODE == 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const float4*>(base + (size_t)(8 * p + (lane >> 3)) * 256 + 4 * (lane & 7));
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) lds_dma16(base + (size_t)(32 * r + i) * 256 + 4 * h + 8 * g, pa_addr + (r * 4 + g) * 1024);
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) lds_dma16(base + (size_t)(8 * p + (lane >> 3)) * 256 + 4 * (lane & 7), pa_addr + p * 1024);
            }
            // "compute" while the loads fly
            float t = (float)lane;
            for (int k = 0; k < work; ++k) t = fmaf(t, 1.0001f, 0.5f);
            acc += t;
            if (MODE >= 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const float4*>(pa + p * 1024 + lane * 16);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += v[p].x + v[p].y + v[p].z + v[p].w;
            if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, const float* src, int n_tiles, int layers, size_t layer_stride, int work, int grid, float* out) {
    CK(hipFuncSetAttribute((const void*)read_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(read_kernel<MODE>, dim3(grid), dim3(512), 65536 * 2, 0, src, n_tiles, layers, layer_stride, work, out);   // (128 KB LDS: one workgroup per CU, as the chain kernels)
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)n_tiles * layers * 65536.0;
    printf("  %-34s grid %5d work %5d: %7.3f ms  %6.2f TB/s\n", name, grid, work, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int rows = 298000 / 64 * 64, n_tiles = rows / 64, layers = 8;
    const size_t layer_stride = (size_t)rows * 256;
    float *src, *out;
    CK(hipMalloc(&src, layer_stride * layers * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 0, layer_stride * layers * 4));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%d rows x 256 fp32 x %d layers = %.2f GB read per launch, %d CUs\n", rows, layers, layer_stride * layers * 4 / 1e9, cus);
    for (int work : {0, 2000, 6000}) {
        for (int persistent : {0, 1}) {
            const int grid = persistent ? cus : n_tiles;
            run<0>("fragment order, register loads", src, n_tiles, layers, layer_stride, work, grid, out);
            run<1>("row major,      register loads", src, n_tiles, layers, layer_stride, work, grid, out);
            run<2>("fragment order, LDS-DMA", src, n_tiles, layers, layer_stride, work, grid, out);
            run<3>("row major,      LDS-DMA", src, n_tiles, layers, layer_stride, work, grid, out);
        }
    }
    return 0;
}
