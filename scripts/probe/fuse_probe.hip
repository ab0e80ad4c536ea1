// fuse_probe.hip -- two hardware questions behind "accumulate dW inside the reverse chain kernel" (DESIGN.md 7c):
//   (1) how fast can 256 workgroups flush 256 KB fp32 partial tiles (one per layer-tile) with hardware float atomics, and where do
//       those atomics execute (XCD-private accumulators stay in that XCD's L2; one device-wide accumulator is fabric traffic)?
//   (2) the exact lane mapping of ds_read_b64_tr_b16, which turns the row-major [row][feature] fp16 planes of the chain kernels into
//       the "8 consecutive rows of one feature per lane" fragments a contraction over the batch rows needs.
// Standalone: hipcc --offload-arch=gfx950 -O3 -o fuse_probe fuse_probe.hip && ./fuse_probe      (scripts/probe/run_fuse_probe.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

// mode 0: one accumulator region for the whole device (agent scope)      mode 1: one region per XCD (workgroup-scope atomics)
// mode 2: one region per XCD, agent-scope atomics                          mode 3: plain stores into a per-workgroup region (baseline)
// mode 4: per-XCD, `layers` regions cycled (working set layers x 256 KB per XCD)
// Each "flush" = the workgroup's 8 waves add 8192 floats each (128 wave-instructions of 64 consecutive floats) = 256 KB.
template <int MODE>
__global__ __launch_bounds__(512) void flush_kernel(float* acc, int flushes, int layers, unsigned* xcc_hist) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc], 1u);
    const size_t REG = 65536;                                 // floats per 256 x 256 accumulator
    for (int f = 0; f < flushes; ++f) {
        float* base;
        if (MODE == 0) base = acc + (size_t)(f % layers) * REG;
        else if (MODE == 3) base = acc + ((size_t)blockIdx.x * layers + (f % layers)) * REG;
        else base = acc + ((size_t)xcc * layers + (f % layers)) * REG;
        float* p = base + wave * 8192 + lane;
#pragma unroll 16
        for (int i = 0; i < 128; ++i) {
            const float v = 1.0f;
            if (MODE == 3) p[i * 64] = v;
            else if (MODE == 1 || MODE == 4) __hip_atomic_fetch_add(p + i * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(p + i * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
static void run_flush(const char* name, int flushes, int layers) {
    const int WG = 256;
    const size_t REG = 65536;
    const size_t regions = (MODE == 0) ? layers : (MODE == 3 ? (size_t)WG * layers : (size_t)8 * layers);
    float* acc;
    unsigned* hist;
    CK(hipMalloc(&acc, regions * REG * 4));
    CK(hipMalloc(&hist, 16 * 4));
    CK(hipMemset(acc, 0, regions * REG * 4));
    CK(hipMemset(hist, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    flush_kernel<MODE><<<WG, 512>>>(acc, 2, layers, hist);    // warm-up
    CK(hipDeviceSynchronize());
    CK(hipMemset(acc, 0, regions * REG * 4));
    CK(hipMemset(hist, 0, 64));
    CK(hipEventRecord(e0));
    flush_kernel<MODE><<<WG, 512>>>(acc, flushes, layers, hist);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h(regions * REG);
    CK(hipMemcpy(h.data(), acc, regions * REG * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (float v : h) sum += v;
    unsigned hh[16];
    CK(hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost));
    const double bytes = (double)WG * flushes * REG * 4;
    const double expect = (MODE == 3) ? (double)regions * REG : (double)WG * flushes * REG;
    printf("%-44s flushes/WG=%3d layers=%2d : %8.3f ms  %7.2f TB/s of fp32 payload  (%.2f us per 256 KB flush per CU)  sum %s  xcc hist %u %u %u %u %u %u %u %u\n",
           name, flushes, layers, ms, bytes / ms * 1e-9, ms * 1e3 / flushes, sum == expect ? "ok" : "MISMATCH",
           hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
    if (sum != expect) printf("   sum %.1f expected %.1f\n", sum, expect);
    CK(hipFree(acc));
    CK(hipFree(hist));
}

// ---- ds_read_b64_tr_b16 lane mapping ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr_kernel(unsigned short* out, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < 64 * row_stride; i += 64) lds[i] = (unsigned short)(((i / row_stride) << 8) | (i % row_stride));   // (row << 8) | col
    __syncthreads();
    const int l = threadIdx.x;
    // lane p of each 16-lane group points at 4 consecutive columns 4 (p & 3).. of row (p >> 2); groups: +16 columns; upper half-wave: +8 rows
    const int p = l & 15, grp = (l >> 4) & 1, h = l >> 5;
    const unsigned short* a = lds + ((p >> 2) + 8 * h) * row_stride + 16 * grp + 4 * (p & 3);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, L2 %d KB\n", prop.name, prop.multiProcessorCount, prop.l2CacheSize / 1024);
    // (2) first: the transposing read
    unsigned short* out;
    CK(hipMalloc(&out, 64 * 4 * 2));
    for (int stride : {64, 264}) {
        tr_kernel<<<1, 64, 64 * stride * 2>>>(out, stride);
        CK(hipDeviceSynchronize());
        unsigned short h[256];
        CK(hipMemcpy(h, out, 512, hipMemcpyDeviceToHost));
        printf("ds_read_b64_tr_b16, row stride %d halves: lane -> 4 x (row,col)\n", stride);
        int ok = 1;
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                const int row = h[l * 4 + j] >> 8, col = h[l * 4 + j] & 255;
                printf(" (%2d,%2d)", row, col);
                // prediction: element j of lane l = row 8 (l >> 5) + j, column 16 ((l >> 4) & 1) + (l & 15)
                if (row != 8 * (l >> 5) + j || col != 16 * ((l >> 4) & 1) + (l & 15)) ok = 0;
            }
            printf("\n");
        }
        printf("  prediction out[l][j] = X[8 (l>>5) + j][16 ((l>>4)&1) + (l&15)] : %s\n", ok ? "CONFIRMED" : "WRONG");
    }
    CK(hipFree(out));
    // (1) the flush
    run_flush<3>("plain stores, per-workgroup region", 36, 1);
    run_flush<0>("atomics agent scope, ONE region", 36, 1);
    run_flush<2>("atomics agent scope, per-XCD region", 36, 1);
    run_flush<1>("atomics workgroup scope, per-XCD region", 36, 1);
    run_flush<4>("atomics wg scope, per-XCD, 9 regions cycled", 36, 9);
    run_flush<4>("atomics wg scope, per-XCD, 16 regions cycled", 32, 16);
    run_flush<0>("atomics agent scope, 9 regions cycled", 36, 9);
    return 0;
}
