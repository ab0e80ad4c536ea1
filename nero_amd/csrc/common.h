// common.h -- error plumbing shared by the translation units of libnero_hip.so
#pragma once
// RULE (round 5): no packed fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in this library.  On gfx950 a wave executing
// them while ANOTHER wave of the same SIMD executes MFMAs can get a quarter-wave (16 lanes) of wrong results -- a few ulp off, rarely, and only
// under co-residency, which is how it hid until a third stream put the narrow weight-gradient kernel beside sdf_alpha_bwd (DESIGN.md 9.3;
// round 3 met the same signature in the two-workgroups-per-CU forward kernel).  The build passes -fno-slp-vectorize and the sources do
// arithmetic on ext_vector_type(float) values component by component; tests/test_no_packed_fp32.py scans the compiled ISA.
#include <hip/hip_runtime.h>

// x * y rounded to fp32 and NEVER contracted into a neighbouring add / subtract (HIP compiles with -ffp-contract=fast, and __fmul_rn is a
// plain product in its headers): the operand packers split fl(W * scale) into planes whose sum must be that fp32 value exactly.
__device__ __forceinline__ float nero_mul_rn(float x, float y) {
#pragma clang fp contract(off)
    return x * y;
}

int nero_fail(int code, const char* msg);          // records msg for nero_last_error(), returns code
int nero_check_launch(const char* what);           // hipGetLastError() -> NERO_OK / NERO_ERR_LAUNCH

#define NERO_ONCE(expr)                                   \
    do {                                                  \
        static bool done_ = false;                        \
        if (!done_) { (void)(expr); done_ = true; }       \
    } while (0)

// optional per-launch timing of the four MFMA kernel classes with HIP events on the launch stream (bench.py roofline leg)
enum { NERO_K_FWD = 0, NERO_K_TAN = 1, NERO_K_BWD = 2, NERO_K_DW = 3, NERO_K_COUNT = 4 };
void nero_prof_begin(int kind, double flops, hipStream_t s);
void nero_prof_mark_paired();                      // the record just opened is a launch of mlp_f16p.hip's kernels (class + NERO_K_COUNT)
void nero_prof_note(long long rows, unsigned sig);  // annotates the record just opened (NERO_PROF_DUMP lines)
void nero_prof_end(int kind, hipStream_t s);
bool nero_prof_is_on();                            // (the step drivers keep ONE stream while launches are being timed)
