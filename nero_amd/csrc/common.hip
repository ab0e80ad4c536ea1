// common.hip -- error state + version of libnero_hip.so
#include <stdio.h>
#include <string.h>
#include "../../include/nero_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

int nero_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int nero_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return NERO_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return NERO_ERR_LAUNCH;
}

extern "C" const char* nero_last_error(void) { return g_err; }
extern "C" int nero_version(void) { return 100; }
