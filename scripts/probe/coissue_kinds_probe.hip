// coissue_kinds_probe.hip -- WHICH VALU instructions of one wave overlap with the MFMAs of the other wave of the same SIMD on gfx950?
// Waves 0-3 (one per SIMD) issue back-to-back v_mfma_f32_32x32x16_f16 on four accumulators; waves 4-7 issue ONE kind of VALU instruction
// (inline asm, 8 independent registers, unrolled 32x per loop trip so that loop overhead is small).  Alone and together, on `argv[1]`
// workgroups (8 = a cold chip at full clock, 256 = every CU busy: the MFMA stream alone is then power-limited to ~64 % of its rate).
// hipcc --offload-arch=gfx950 -O3 -o coissue_kinds_probe coissue_kinds_probe.hip && ./coissue_kinds_probe 8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(int nm, int nv, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
        f16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.001f * (lane + j)); y[j] = (_Float16)(0.002f * (lane - j)); }
        for (int i = 0; i < nm; i += 4) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, acc[3], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) r += acc[a][3];
    } else {
        float c[8];
        double d[8];
        for (int j = 0; j < 8; ++j) { c[j] = 0.5f + 0.001f * (lane + j); d[j] = c[j]; }
        const float m = 0.9999f, a = 0.0001f;
        for (int i = 0; i < nv; i += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#define ONE(J) \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[J]) : "v"(m), "v"(a)); \
    else if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c[J]) : "v"(m)); \
    else if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c[J]) : "v"(a)); \
    else if (KIND == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(c[J]) : "v"(a)); \
    else if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(c[J]) : "v"(m)); \
    else if (KIND == 5) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(c[J])); \
    else if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[J]) : "v"(d[(J + 1) & 7])); \
    else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[J]) : "v"(d[(J + 1) & 7]), "v"(d[(J + 2) & 7])); \
    else if (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c[J]) : "v"(m), "v"(a)); \
    else if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(c[J]) : "v"(m)); \
    else if (KIND == 10) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(c[J]) : "v"(a)); \
    else if (KIND == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c[J]) : "v"(m), "v"(a)); \
    else if (KIND == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(c[J])); \
    else if (KIND == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[J]) : "v"(d[(J + 1) & 7]));
                REP8(ONE)
#undef ONE
            }
        }
        for (int j = 0; j < 8; ++j) r += c[j] + (float)d[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
static int g_grid = 8;
template <int KIND>
float timeit(int nm, int nv, float* out) {
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<g_grid, 512, 100 * 1024>>>(nm, nv, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<g_grid, 512, 100 * 1024>>>(nm, nv, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int KIND>
void one(const char* name, float* out) {
    const int NM = 100000, NV = 800000;
    const float m = timeit<KIND>(NM, 0, out), v = timeit<KIND>(0, NV, out), both = timeit<KIND>(NM, NV, out);
    printf("%-28s MFMA alone %.3f ms (%.1f cyc)  VALU alone %.3f ms (%.1f cyc/instr)  together %.3f  sum %.3f  overlap %3.0f %%\n", name, m,
           m * 1e-3 * 2.4e9 / NM, v, v * 1e-3 * 2.4e9 / NV, both, m + v, 100.0 * (m + v - both) / fminf(m, v));
}
int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    printf("== %d workgroups (one per CU)\n", g_grid);
    float* out; hipMalloc(&out, 4096);
    one<0>("v_fma_f32 (3 vgpr src)", out);
    one<10>("v_fma_f32 v, 1.0, v (2 vgpr)", out);
    one<1>("v_mul_f32", out);
    one<2>("v_add_f32", out);
    one<3>("v_max_f32", out);
    one<8>("v_max3_f32", out);
    one<11>("v_mad_u32_u24", out);
    one<4>("v_cvt_pk_f16_f32", out);
    one<5>("v_cvt_f32_f16", out);
    one<12>("v_exp_f32", out);
    one<9>("v_mov_b32", out);
    one<6>("v_pk_mul_f32", out);
    one<13>("v_pk_add_f32", out);
    one<7>("v_pk_fma_f32", out);
    return 0;
}
