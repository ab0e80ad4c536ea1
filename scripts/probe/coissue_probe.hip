// coissue_probe.hip -- do MFMAs of one wave and VALU instructions of ANOTHER wave of the same SIMD overlap on gfx950?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) issue NM back-to-back v_mfma_f32_32x32x16_f16 on four independent
// accumulators; waves 4-7 (the second wave of each SIMD) issue NV VALU instructions (v_fma_f32 or v_pk_fma_f32 on 8 independent chains).
// Timed alone and together: together ~ max(alone) = the pipes overlap; together ~ sum = they exclude each other.
// CAUTION (found later the same round): under plain -O3 hipcc SLP-packs the eight independent fmaf chains of KIND 0 into v_pk_fma_f32, so
// rows 0 and 1 both measure the PACKED instruction -- which never overlaps MFMAs -- and at 256 workgroups the MFMA stream alone is already
// power-limited (51 cycles per MFMA instead of 32).  coissue_kinds_probe.hip pins one instruction per row with inline asm and takes the
// workgroup count as an argument; read its numbers, not this file's, for "what overlaps".
// hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip && ./coissue_probe [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>   // VALU kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cvt_pk_f16_f32 + v_max (conversion-like mix)
__global__ __launch_bounds__(512, 1) void k(int nm, int nv, float* out) {
    extern __shared__ char smem[];
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
        if (KIND == 0) {
            float c[8];
            for (int j = 0; j < 8; ++j) c[j] = 0.5f + 0.001f * (lane + j);
            for (int i = 0; i < nv; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = fmaf(c[j], 0.9999f, 0.0001f);
            }
            for (int j = 0; j < 8; ++j) r += c[j];
        } else if (KIND == 1) {
            f32x2 c[8];
            for (int j = 0; j < 8; ++j) { c[j][0] = 0.5f + 0.001f * (lane + j); c[j][1] = 0.25f; }
            const f32x2 m = {0.9999f, 0.9998f}, a = {0.0001f, 0.0002f};
            for (int i = 0; i < nv; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = __builtin_elementwise_fma(c[j], m, a);
            }
            for (int j = 0; j < 8; ++j) r += c[j][0] + c[j][1];
        } else {
            float c[8]; unsigned acc = 0;
            for (int j = 0; j < 8; ++j) c[j] = 0.5f + 0.001f * (lane + j);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            for (int i = 0; i < nv; i += 16) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 ab = {c[j], c[j + 1]};
                    h2 hh = __builtin_convertvector(ab, h2);
                    acc ^= __builtin_bit_cast(unsigned, hh);
                    c[j] = fmaxf(c[j] * 0.999f, (float)hh[0]); c[j + 1] = fmaxf(c[j + 1] * 0.999f, (float)hh[1]);
                }
            }
            r = (float)acc;
            for (int j = 0; j < 8; ++j) r += c[j];
        }
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
static int g_grid = 256;                         // workgroups = CUs kept busy (argv[1]): 256 = the whole chip (power-limited clocks), 8 = a cold chip
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
int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    printf("== %d workgroups (one per CU)\n", g_grid);
    float* out; hipMalloc(&out, 4096);
    const int NM = 200000;                       // MFMAs per wave: 200000 x 32 cycles = 6.4 M cycles
    const char* names[3] = {"v_fma_f32", "v_pk_fma_f32", "cvt_pk_f16 + cvt_f32_f16 + mul + max mix"};
    const float m = timeit<0>(NM, 0, out);
    printf("MFMA alone (one wave per SIMD, %d back-to-back v_mfma_f32_32x32x16_f16): %.3f ms = %.1f cycles per MFMA at 2.4 GHz\n", NM, m, m * 1e-3 * 2.4e9 / NM);
    for (int kind = 0; kind < 3; ++kind) {
        const int NV = 1600000;
        float v, both;
        if (kind == 0) { v = timeit<0>(0, NV, out); both = timeit<0>(NM, NV, out); }
        else if (kind == 1) { v = timeit<1>(0, NV, out); both = timeit<1>(NM, NV, out); }
        else { v = timeit<2>(0, NV, out); both = timeit<2>(NM, NV, out); }
        printf("%-44s VALU alone %.3f ms (%.1f cycles per instruction-slot), together %.3f ms: max %.3f  sum %.3f  -> overlap %.0f %%\n", names[kind], v,
               v * 1e-3 * 2.4e9 / NV, both, fmaxf(m, v), m + v, 100.0 * (m + v - both) / fminf(m, v));
    }
    return 0;
}
