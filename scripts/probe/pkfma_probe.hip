// pkfma_probe.hip -- second probe for the fwd_p_kernel fault.  The failing build's epilogue is, per pair of columns,
//     v_pk_fma_f32 T, aL, K(sgpr pair), aH ; s_nop 0 ; v_pk_fma_f32 val, T, U, bias op_sel:[0,1,0] ; v_pk_fma_f32 T, aL', K, aH' ; ...
// with ONE temporary register pair T rewritten by the instruction that follows its reader, and the build with every fma issued on its
// own (no v_pk_fma_f32) does not fail.  This probe issues exactly that instruction sequence from inline assembly on 8 register pairs and
// compares every value with the scalar v_fma_f32 evaluation of the same inputs (both fused, same rounding: must be bit-equal), in four
// settings: 0 = packed-epilogue workgroups next to MFMA workgroups (two per CU, one wave of each per SIMD), 1 = packed only (control),
// 2 = every workgroup alternates MFMA and epilogue phases out of step, 3 = as the kernel: each wave runs an MFMA chain and puts the
// epilogue on ITS accumulators while the other workgroup of the CU is somewhere else in the same loop.
// Counts: values that differ from the scalar evaluation, and of those the ones that equal the bias (the signature of the fault).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float launder(float x) { asm volatile("" : "+v"(x)); return x; }

struct Cnt { unsigned long long bad, eq_bias; };

// 8 pairs through the packed sequence with one shared temporary; U.y is the unit used (op_sel picks the high half for both results)
__device__ __forceinline__ void packed8(f32x2 (&o)[8], const f32x2 (&l)[8], const f32x2 (&h)[8], const f32x2 (&b)[8], f32x2 u, f32x2 k) {
    f32x2 t;
    asm volatile(
        "s_nop 7\n s_nop 7\n s_nop 3\n"
        "v_pk_fma_f32 %[t], %[l0], %[k], %[h0]\n s_nop 0\n v_pk_fma_f32 %[o0], %[t], %[u], %[b0] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l1], %[k], %[h1]\n s_nop 0\n v_pk_fma_f32 %[o1], %[t], %[u], %[b1] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l2], %[k], %[h2]\n s_nop 0\n v_pk_fma_f32 %[o2], %[t], %[u], %[b2] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l3], %[k], %[h3]\n s_nop 0\n v_pk_fma_f32 %[o3], %[t], %[u], %[b3] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l4], %[k], %[h4]\n s_nop 0\n v_pk_fma_f32 %[o4], %[t], %[u], %[b4] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l5], %[k], %[h5]\n s_nop 0\n v_pk_fma_f32 %[o5], %[t], %[u], %[b5] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l6], %[k], %[h6]\n s_nop 0\n v_pk_fma_f32 %[o6], %[t], %[u], %[b6] op_sel:[0,1,0]\n"
        "v_pk_fma_f32 %[t], %[l7], %[k], %[h7]\n s_nop 0\n v_pk_fma_f32 %[o7], %[t], %[u], %[b7] op_sel:[0,1,0]\n"
        "s_nop 1\n"
        : [t] "=&v"(t), [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]),
          [o6] "=&v"(o[6]), [o7] "=&v"(o[7])
        : [l0] "v"(l[0]), [l1] "v"(l[1]), [l2] "v"(l[2]), [l3] "v"(l[3]), [l4] "v"(l[4]), [l5] "v"(l[5]), [l6] "v"(l[6]), [l7] "v"(l[7]),
          [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]),
          [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]),
          [u] "v"(u), [k] "s"(k));
}

__device__ __forceinline__ void check8(Cnt& c, const f32x2 (&o)[8], const f32x2 (&l)[8], const f32x2 (&h)[8], const f32x2 (&b)[8], f32x2 u, float kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r0 = fmaf(fmaf(launder(l[j].x), kk, launder(h[j].x)), launder(u.y), launder(b[j].x));
        const float r1 = fmaf(fmaf(launder(l[j].y), kk, launder(h[j].y)), launder(u.y), launder(b[j].y));
        if (__float_as_uint(r0) != __float_as_uint(o[j].x)) { ++c.bad; if (o[j].x == b[j].x) ++c.eq_bias; }
        if (__float_as_uint(r1) != __float_as_uint(o[j].y)) { ++c.bad; if (o[j].y == b[j].y) ++c.eq_bias; }
    }
}

__device__ void valu_phase(Cnt& c, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    f32x2 l[8], h[8], b[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j].x = (float)((lane * 37 + j * 11 + seed) % 251) * 0.013f - 1.1f;  h[j].y = (float)((lane * 31 + j * 7 + seed) % 239) * 0.017f - 1.3f;
        l[j].x = (float)((lane * 17 + j * 29 + seed) % 241) * 3.1f;           l[j].y = (float)((lane * 13 + j * 23 + seed) % 233) * 2.7f;
        b[j].x = (float)((lane + 3 * j + seed) % 97) * 0.01f + 0.02f;         b[j].y = (float)((lane + 5 * j + seed) % 89) * 0.01f + 0.03f;
    }
    f32x2 u = {0.25f, 0.5f + (float)(lane & 7) * 0.125f};
    f32x2 k = {4.8828125e-4f, 4.8828125e-4f};
    for (int it = 0; it < iters; ++it) {
        packed8(o, l, h, b, u, k);
        check8(c, o, l, h, b, u, 4.8828125e-4f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = o[j] * 0.5f + 0.1f; l[j] = l[j] * 0.999f + 1.0f; }
        u.y = u.y * 0.99f + 0.01f;
    }
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

// as the kernel: 16 MFMA steps into two accumulators (hi / lo products), then the packed epilogue on the 16 + 16 accumulator registers
__device__ void fused_phase(Cnt& c, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    f16x8 A, B, B2;
    for (int k = 0; k < 8; ++k) {
        A[k] = (_Float16)(0.01f * (float)((threadIdx.x + k + seed) % 13) - 0.05f);
        B[k] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + k) % 7) - 0.04f);
        B2[k] = (_Float16)(0.5f * (float)((threadIdx.x * 5 + k) % 11));
    }
    f32x2 u = {0.25f, 0.5f + (float)(lane & 7) * 0.125f};
    f32x2 k = {4.8828125e-4f, 4.8828125e-4f};
    for (int it = 0; it < iters; ++it) {
        f32x16 aH, aL;
        for (int v = 0; v < 16; ++v) { aH[v] = 0.f; aL[v] = 0.f; }
        const int steps = 8 + ((blockIdx.x + it) & 7);                           // (the two workgroups of a CU drift against each other)
        for (int s = 0; s < steps; ++s) {
            aH = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, aH, 0, 0, 0);
            aL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B2, aL, 0, 0, 0);
        }
        f32x2 l[8], h[8], b[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j].x = aH[2 * j]; h[j].y = aH[2 * j + 1];
            l[j].x = aL[2 * j]; l[j].y = aL[2 * j + 1];
            b[j].x = (float)((lane + 3 * j + it) % 97) * 0.01f + 0.02f; b[j].y = (float)((lane + 5 * j + it) % 89) * 0.01f + 0.03f;
        }
        packed8(o, l, h, b, u, k);
        check8(c, o, l, h, b, u, 4.8828125e-4f);
        A[it & 7] = (_Float16)(0.003f * (float)((it + lane) % 17));
    }
}

__global__ __launch_bounds__(256, 2) void probe(int mode, int iters, unsigned long long* out, float* sink) {
    extern __shared__ char smem[];
    (void)smem;
    Cnt c = {0, 0};
    float s = 0.f;
    if (mode == 0) { if ((blockIdx.x & 1) == 0) valu_phase(c, iters, 1); else s = mfma_phase(iters * 24, 1); }
    else if (mode == 1) valu_phase(c, iters, 1);
    else if (mode == 2) {
        for (int rep = 0; rep < 8; ++rep) { if (((blockIdx.x + rep) & 1) == 0) valu_phase(c, iters / 8, rep); else s += mfma_phase(iters * 3, rep); }
    } else fused_phase(c, iters / 4, 1);
    if (c.bad) { atomicAdd(out, c.bad); atomicAdd(out + 1, c.eq_bias); }
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    unsigned long long* out; float* sink;
    if (hipMalloc(&out, 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 79360);
    const char* names[4] = {"packed-epilogue and MFMA workgroups interleaved, two per CU", "packed-epilogue workgroups only", "alternating phases",
                            "MFMA chain then packed epilogue on its accumulators (as the kernel)"};
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(out, 0, 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 79360, 0, mode, 4000, out, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2] = {0, 0}; (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double vals = 4096.0 * 256 * 16 * (mode == 0 ? 0.5 : mode == 2 ? 0.5 : 1.0) * (mode == 3 ? 1000 : 4000);
        printf("mode %d (%s): %llu of %.2e values differ from the scalar evaluation, %llu of them equal the bias; %.1f ms, %s\n", mode, names[mode],
               h[0], vals, h[1], ms, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
