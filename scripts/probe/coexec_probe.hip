// coexec_probe.hip -- does a wave's VALU epilogue lose results while the wave sharing its SIMD, from ANOTHER workgroup, retires MFMAs?
// (the hypothesis left open in DESIGN.md 3i for fwd_p_kernel).  256-thread workgroups with 78 KB of LDS each: two per CU, one wave of
// each on every SIMD.  Even workgroups run the epilogue pattern  val = fma(fma(aL, 2^-11, aH), U, b)  on 32 register values per lane,
// twice from separately laundered copies of the same inputs, and count lanes whose two results differ; odd workgroups run back-to-back
// v_mfma_f32_32x32x16_f16.  Modes: 0 = VALU + MFMA workgroups interleaved, 1 = VALU workgroups only (control), 2 = every workgroup
// alternates both phases out of step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float launder(float x) { asm volatile("" : "+v"(x)); return x; }

__device__ unsigned long long valu_phase(int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    float aH[32], aL[32], b[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        aH[j] = (float)((lane * 37 + j * 11 + seed) % 251) * 0.013f - 1.1f;
        aL[j] = (float)((lane * 17 + j * 29 + seed) % 241) * 3.1f;
        b[j] = (float)((lane + 3 * j + seed) % 97) * 0.01f + 0.02f;
    }
    float U = 0.5f + (float)(lane & 7) * 0.125f;
    for (int it = 0; it < iters; ++it) {
        float v1[32], v2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v1[j] = fmaf(fmaf(launder(aL[j]), 4.8828125e-4f, launder(aH[j])), launder(U), launder(b[j]));
#pragma unroll
        for (int j = 0; j < 32; ++j) v2[j] = fmaf(fmaf(launder(aL[j]), 4.8828125e-4f, launder(aH[j])), launder(U), launder(b[j]));
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (v1[j] != v2[j]) ++bad;
            aH[j] = v1[j] * 0.5f + 0.1f;            // (new inputs every iteration, bounded)
            aL[j] = aL[j] * 0.999f + 1.0f;
        }
        U = U * 0.99f + 0.01f;
    }
    return bad;
}

__device__ float mfma_phase(int iters, unsigned seed) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    f16x8 A, B;
    for (int k = 0; k < 8; ++k) { A[k] = (_Float16)(0.01f * (float)((threadIdx.x + k + seed) % 13)); B[k] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + k) % 7)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int v = 0; v < 16; ++v) s += acc[a][v];
    return s;
}

__global__ __launch_bounds__(256, 2) void probe(int mode, int iters, unsigned long long* bad, float* sink) {
    extern __shared__ char smem[];
    (void)smem;
    const bool valu_wg = mode == 1 || (mode == 0 && (blockIdx.x & 1) == 0);
    unsigned long long nb = 0;
    float s = 0.f;
    if (mode == 2) {
        for (int rep = 0; rep < 8; ++rep) {
            if (((blockIdx.x + rep) & 1) == 0) nb += valu_phase(iters / 8, rep); else s += mfma_phase(iters * 4, rep);
        }
    } else if (valu_wg) nb = valu_phase(iters, 1);
    else s = mfma_phase(iters * 32, 1);
    if (nb) atomicAdd(bad, nb);
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    unsigned long long* bad; float* sink;
    hipMalloc(&bad, 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 79360);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 79360, 0, mode, 4000, bad, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %llu lane-values whose two evaluations differ, %.1f ms, %s\n", mode,
               mode == 0 ? "VALU and MFMA workgroups interleaved, two per CU" : mode == 1 ? "VALU workgroups only" : "alternating phases", h, ms,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
