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
