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

extern "C" int nero_check_device_memory(size_t need_bytes, size_t reusable_bytes, const char* what) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return nero_fail(NERO_ERR_LAUNCH, "nero_check_device_memory: hipMemGetInfo failed (no device?)");
    }
    if (need_bytes <= free_b + reusable_bytes) return NERO_OK;
    const double G = 1.0 / (1024.0 * 1024.0 * 1024.0);
    snprintf(g_err, sizeof(g_err), "%s: needs %zu bytes (%.1f GiB); the device has %zu free + %zu re-usable by the caller = %.1f GiB of %.1f GiB. "
             "Fewer rays / points per GPU, or NERO_STREAMS=2 (no third stream: chain deltas are released early), shrink it.",
             what ? what : "workspace", need_bytes, need_bytes * G, free_b, reusable_bytes, (free_b + reusable_bytes) * G, total_b * G);
    return NERO_ERR_NOMEM;
}
extern "C" int nero_version(void) { return 100; }

// ---- per-launch kernel timing (disabled by default) ----------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops; long long rows; unsigned sig; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
hipEvent_t g_cur_a;
}
void nero_prof_begin(int kind, double flops, hipStream_t s) {
    if (!g_prof_on) return;
    ProfRec r; r.kind = kind; r.flops = flops; r.rows = 0; r.sig = 0u;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void nero_prof_end(int kind, hipStream_t s) {
    if (!g_prof_on || g_recs.empty()) return;
    (void)kind;
    (void)hipEventRecord(g_recs.back().b, s);
}
// annotate the record just opened (NERO_PROF_DUMP only): rows of the launch and a caller-defined signature word
void nero_prof_note(long long rows, unsigned sig) {
    if (!g_prof_on || g_recs.empty()) return;
    g_recs.back().rows = rows; g_recs.back().sig = sig;
}
bool nero_prof_is_on() { return g_prof_on; }
extern "C" int nero_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
// annotate the record just opened as a launch of the two-workgroups-per-CU kernels (mlp_f16p.hip): reported as its own class
void nero_prof_mark_paired() {
    if (!g_prof_on || g_recs.empty() || g_recs.back().kind >= NERO_K_COUNT) return;
    g_recs.back().kind += NERO_K_COUNT;
}
static int prof_report(double* out, int n_kinds);
// out[kind*3 + {0,1,2}] = {launches, total milliseconds, total algorithmic flops}; clears the records.  Four classes (forward, tangent,
// reverse, weight gradient): the paired-kernel launches are counted with their pass
extern "C" int nero_prof_report(double* out /*host, 12 doubles*/) { return prof_report(out, NERO_K_COUNT); }
// the same per KERNEL: classes 0-3 as above but only the 512-thread / weight-gradient kernels, 4-6 = fwd_p / tan_p / bwd_p kernel, 7 unused
extern "C" int nero_prof_report_kernels(double* out /*host, 24 doubles*/) { return prof_report(out, 2 * NERO_K_COUNT); }
static int prof_report(double* out, int n_kinds) {
    for (int i = 0; i < n_kinds * 3; ++i) out[i] = 0.0;
    // NERO_PROF_DUMP=<file>: additionally append one "kind ms flops rows signature" line per launch (tuning aid)
    const char* dump = getenv("NERO_PROF_DUMP");
    FILE* df = dump ? fopen(dump, "a") : nullptr;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        const int k = r.kind % n_kinds;
        out[k * 3 + 0] += 1.0; out[k * 3 + 1] += ms; out[k * 3 + 2] += r.flops;
        if (df) fprintf(df, "%d %.4f %.0f %lld 0x%x\n", r.kind, ms, r.flops, r.rows, r.sig);
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    if (df) fclose(df);
    g_recs.clear();
    return 0;
}
