// common.h -- error plumbing shared by the translation units of libnero_hip.so
#pragma once
#include <hip/hip_runtime.h>

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
void nero_prof_end(int kind, hipStream_t s);
bool nero_prof_is_on();                            // (the step drivers keep ONE stream while launches are being timed)
