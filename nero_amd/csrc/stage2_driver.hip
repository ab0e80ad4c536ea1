// stage2_driver.hip -- the C-level driver of the Stage-II (material) shading step (nero_mc_shade_fwd / _bwd):
// predict_materials (network/field.py:915-922, MaterialFeatsNetwork :660-689) and MCShadingNetwork.shade_mixed / get_lights
// (:856-880, 950-1012) with their backward, sequenced in the library exactly as nero_amd/material_step.py does from Python.
// Host code only (the kernels are the library's own entry points).  The mesh tracer stays the CALLER's call between
// nero_stage2_rays and nero_stage2_shade_fwd -- nero_bvh_trace in production, anything with the same contract in tests -- which is
// also where the reference calls its third-party tracer (raytracing/raytracer.py:49 from network/field.py:860).
// Memory: one caller-owned workspace, carved by the same arena as the Stage-I driver; host synchronisation: one read-back of the
// (miss, hit) ray counts, which size the light-MLP launches.
#include "chain_host.h"
#include <stdlib.h>
#ifndef NERO_STREAMS_DEFAULT
#define NERO_STREAMS_DEFAULT 2
#endif

enum { M_FEATS = 0, M_METALLIC = 8, M_ROUGHNESS = 12, M_ALBEDO = 16, M_OUTER = 20, M_INNER = 24, M_HUMAN = 28 };

namespace {

__global__ void x8_from_x3_kernel(const float* __restrict__ x, int n, int n_pad, float* __restrict__ x8) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) v = make_float4(x[(size_t)r * 3], x[(size_t)r * 3 + 1], x[(size_t)r * 3 + 2], 0.f);
    reinterpret_cast<float4*>(x8)[2 * r] = v;
    reinterpret_cast<float4*>(x8)[2 * r + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// raw5[r] = (metallic head, roughness head, albedo head[0..2]) of row r
__global__ void gather_raw5_kernel(const float* __restrict__ hm, const float* __restrict__ hr, const float* __restrict__ ha, int n,
                                   float* __restrict__ raw5) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    raw5[(size_t)r * 5] = hm[(size_t)r * 4];
    raw5[(size_t)r * 5 + 1] = hr[(size_t)r * 4];
    raw5[(size_t)r * 5 + 2] = ha[(size_t)r * 4];
    raw5[(size_t)r * 5 + 3] = ha[(size_t)r * 4 + 1];
    raw5[(size_t)r * 5 + 4] = ha[(size_t)r * 4 + 2];
}
// dh[r] = (d_raw5[r][c0 .. c0 + nc), 0...) for r < n, 0 for the padding rows
__global__ void scatter_dhead_kernel(const float* __restrict__ d_raw5, int n, int n_pad, int c0, int nc, float* __restrict__ dh) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n) for (int c = 0; c < nc; ++c) v[c] = d_raw5[(size_t)r * 5 + c0 + c];
    reinterpret_cast<float4*>(dh)[r] = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

struct nero_stage2 {
    nero_stage2_cfg cfg;
    Modes M;
    int kout = 72;
    bool packed = false;
    Chain feats, mat[3], outer_light, inner_light, human_light;
    int device = -1;                                   // the device the private stream / events were created on (nero_stage2_create)
    Arena A;
    size_t predict_mark = 0, shade_mark = 0;
    // predict state
    int n_pred = 0;
    Fwd f_feats, f_mat[3];
    float *pe = nullptr, *x8 = nullptr;
    // shade state
    int P = 0, n_miss = 0, n_hit = 0;
    float *pt = nullptr, *Xm = nullptr, *Xh = nullptr, *Xhum = nullptr, *hmask = nullptr;
    int *slot = nullptr, *miss_idx = nullptr, *hit_idx = nullptr, *counts = nullptr;
    // rays whose estimator weight is exactly zero (mc_shade.hip, DEAD_SLOT): flagged in nero_stage2_rays, skipped by the tracer the caller hands
    // the flags to, left out of both light MLPs.  NERO_MC_SKIP_DEAD=0 (read at create) keeps every ray, as rounds 1-5 did.
    bool skip_dead = true;
    unsigned char* dead = nullptr;
    // the human-light MLP's output is multiplied by the plane-hit mask of its ray (field.py:829): with the flags of nero_mc_human_flags the
    // miss list is partitioned (nero_mc_split_classes) and that MLP runs on the miss rows [0, n_hum) only.  Off with skip_dead: n_hum = n_miss.
    unsigned char* hum = nullptr;
    int n_hum = 0;
    Fwd f_out, f_in, f_hum;
    const float *dirs = nullptr, *depth = nullptr, *fnrm = nullptr, *poses = nullptr, *tab_s = nullptr;
    // a second stream for the HIT rows (inner-light MLP) beside the MISS rows (outer / human light): the two row sets share nothing
    // between the compaction and the estimator (forward), resp. between its backward and the direction backward (as stage1_driver.hip)
    int n_streams = 1;
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace {

// (as stage1_driver.hip) a failing exit between a fork and its join drains the private stream; calls under another current device are refused
int drain_on_error(nero_stage2* h, int rc) {
    if (rc == NERO_OK) return rc;
    if (h->s2) (void)hipStreamSynchronize(h->s2);
    (void)hipGetLastError();
    return rc;
}
bool wrong_device(const nero_stage2* h) {
    int dev = -1;
    return h->s2 && hipGetDevice(&dev) == hipSuccess && h->device >= 0 && dev != h->device;
}
hipStream_t fork_side(nero_stage2* h, const Arena& A, hipStream_t main) {
    if (A.dry || h->n_streams < 2 || !h->s2 || nero_prof_is_on()) return main;
    (void)hipEventRecord(h->ev_fork, main);
    (void)hipStreamWaitEvent(h->s2, h->ev_fork, 0);
    return h->s2;
}
void join_side(nero_stage2* h, hipStream_t side, hipStream_t main) {
    if (side == main) return;
    (void)hipEventRecord(h->ev_join, side);
    (void)hipStreamWaitEvent(main, h->ev_join, 0);
}

void build_chains(nero_stage2* h, const nero_stage2_weights* w) {
    const nero_linear* L = w->lin;
    Chain f;                         // MaterialFeatsNetwork: PE-8(p) (51) -> 4 x 256 ReLU -> cat[h, pe] -> 3 x 256 ReLU + Linear 256
    for (int i = 0; i < 8; ++i) {
        if (i == 0) f.e.push_back(dense(L[M_FEATS], 51, 256, NERO_ACT_RELU, 51));
        else if (i == 4) f.e.push_back(dense(L[M_FEATS + 4], 307, 256, NERO_ACT_RELU, 256, 0, 51, 256));
        else f.e.push_back(dense(L[M_FEATS + i], 256, 256, i == 7 ? NERO_ACT_NONE : NERO_ACT_RELU, 256));
    }
    f.k_init = 56; f.k_aux = 56; f.aux_wide = 1;
    h->feats = f;
    h->mat[0] = predictor(L + M_METALLIC, 256, 3, 256, 8, 1);
    h->mat[1] = predictor(L + M_ROUGHNESS, 256, 3, 256, 8, 1);
    h->mat[2] = predictor(L + M_ALBEDO, 256, 3, 256, 8, 3);
    h->kout = h->cfg.sphere_direction ? 144 : 72;
    h->outer_light = predictor(L + M_OUTER, h->kout, 0, h->kout, 0, 3);
    h->inner_light = predictor(L + M_INNER, 123, 0, 128, 0, 3);
    if (h->cfg.human_lights) h->human_light = predictor(L + M_HUMAN, 24, 0, 24, 0, 4);
}

std::vector<Chain*> all_chains(nero_stage2* h) {
    std::vector<Chain*> v = {&h->feats, &h->outer_light, &h->inner_light};
    if (h->cfg.human_lights) v.push_back(&h->human_light);
    v.push_back(&h->mat[0]); v.push_back(&h->mat[1]); v.push_back(&h->mat[2]);
    return v;
}

size_t pack_floats_total(nero_stage2* h) {
    size_t t = 0;
    for (Chain* c : all_chains(h)) t += (c->pack_floats() + 63) / 64 * 64;
    return t;
}

int do_predict_fwd(nero_stage2* h, Arena& A, const float* x, int n, float* raw5, void* stream) {
    hipStream_t hs = (hipStream_t)stream;
    const int rp = rpad(n);
    h->n_pred = n;
    h->pe = A.f32((size_t)rp * 56);
    h->x8 = A.f32((size_t)rp * 8);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_predict_fwd: workspace too small");
    LAUNCH(nero_encode_pe(x, 3, 3, 8, n, h->pe, 56, stream));
    if (!A.dry) hipLaunchKernelGGL(x8_from_x3_kernel, dim3((rp + 255) / 256), dim3(256), 0, hs, x, n, rp, h->x8);
    h->f_feats = Fwd();
    RC(h->feats.forward(A, h->M, h->pe, 56, h->pe, 56, n, true, h->f_feats, stream));
    for (int j = 0; j < 3; ++j) {
        h->f_mat[j] = Fwd();
        RC(h->mat[j].forward(A, h->M, h->f_feats.saves[7], NERO_HID, h->x8, 8, n, true, h->f_mat[j], stream));
    }
    if (!A.dry) hipLaunchKernelGGL(gather_raw5_kernel, dim3((n + 255) / 256), dim3(256), 0, hs, h->f_mat[0].heads[3], h->f_mat[1].heads[3],
                                   h->f_mat[2].heads[3], n, raw5);
    return NERO_OK;
}

int do_predict_bwd(nero_stage2* h, Arena& A, const float* d_raw5, const nero_stage2_grads* G, void* stream) {
    hipStream_t hs = (hipStream_t)stream;
    const int n = h->n_pred, rp = rpad(n);
    const nero_linear_grad* g = G->lin;
    float* partials = A.f32((size_t)nero_dw_workspace_floats(n > 1 ? n : 1));
    float* d_feats = A.f32((size_t)rp * NERO_HID);
    const int c0[3] = {0, 1, 2}, nc[3] = {1, 1, 3}, lidx[3] = {M_METALLIC, M_ROUGHNESS, M_ALBEDO};
    for (int j = 0; j < 3; ++j) {
        const size_t mk = A.mark();
        float* dh = A.f32((size_t)rp * 4);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_predict_bwd: workspace too small");
        if (!A.dry) hipLaunchKernelGGL(scatter_dhead_kernel, dim3((rp + 255) / 256), dim3(256), 0, hs, d_raw5, n, rp, c0[j], nc[j], dh);
        predictor_grads(h->mat[j], g + lidx[j], 259);
        const float* hd[MAXL] = {};
        hd[3] = dh;
        Bwd mb;
        RC(h->mat[j].backward(A, h->M, h->f_mat[j], n, nullptr, 0, hd, true, false, nullptr, d_feats, NERO_HID, j > 0, false, mb, stream));
        RC(h->mat[j].weight_grads(A, h->M, h->f_mat[j], mb, n, h->f_feats.saves[7], NERO_HID, h->x8, 8, hd, nullptr, nullptr, partials, stream));
        A.release(mk);
    }
    for (int i = 0; i < 8; ++i) set_dense_grad(h->feats.e[i], g[M_FEATS + i], i == 0 ? 51 : (i == 4 ? 307 : 256));
    Bwd fb;
    RC(h->feats.backward(A, h->M, h->f_feats, n, d_feats, NERO_HID, nullptr, false, false, nullptr, nullptr, 0, false, false, fb, stream));
    RC(h->feats.weight_grads(A, h->M, h->f_feats, fb, n, h->pe, 56, h->pe, 56, nullptr, nullptr, nullptr, partials, stream));
    return nero_check_launch("nero_stage2_predict_bwd");
}

// light MLPs on the compacted miss / hit rows + the microfacet estimator; n_miss / n_hit known
int do_shade_lights(nero_stage2* h, Arena& A, const float* pos, float* rgb, float* dl, float* sl, float* sp, void* stream) {
    const nero_stage2_cfg& c = h->cfg;
    const int D = c.diffuse_sample_num + c.specular_sample_num;
    const int n_miss = h->n_miss, n_hit = h->n_hit;
    const int rpm = rpad(n_miss), rph = rpad(n_hit);
    h->Xm = A.f32((size_t)(rpm > 64 ? rpm : 64) * h->kout);
    h->Xh = A.f32((size_t)(rph > 64 ? rph : 64) * 128);
    h->Xhum = h->hmask = nullptr;
    h->f_out = Fwd(); h->f_in = Fwd(); h->f_hum = Fwd();
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: workspace too small");
    // the hit branch starts behind everything the caller's stream holds NOW (before the miss branch is issued on it)
    const hipStream_t fork_side_early = (n_miss > 0 && n_hit > 0) ? fork_side(h, A, (hipStream_t)stream) : (hipStream_t)stream;
    if (n_miss > 0) {
        LAUNCH(nero_mc_encode_miss(h->dirs, h->miss_idx, h->pt, D, c.sphere_direction, n_miss, h->Xm, stream));
        RC(h->outer_light.forward(A, h->M, h->Xm, h->kout, nullptr, 0, n_miss, true, h->f_out, stream));
        if (c.human_lights && h->n_hum > 0) {                 // (the miss rows [0, n_hum): the rays that reach the photographer's region)
            const int rpu = rpad(h->n_hum) > 64 ? rpad(h->n_hum) : 64;
            h->Xhum = A.f32((size_t)rpu * 24); h->hmask = A.f32(rpu);
            if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: workspace too small");
            LAUNCH(nero_mc_human_encode(h->dirs, h->miss_idx, h->pt, D, h->poses, h->n_hum, h->Xhum, h->hmask, stream));
            RC(h->human_light.forward(A, h->M, h->Xhum, 24, nullptr, 0, h->n_hum, true, h->f_hum, stream));
        }
    }
    hipStream_t side = (hipStream_t)stream;
    if (n_hit > 0) {
        if (n_miss > 0) side = fork_side_early;            // (forked before the miss branch was issued, see above)
        LAUNCH(nero_mc_encode_hit(h->dirs, pos, h->fnrm, h->hit_idx, n_hit, h->Xh, (void*)side));
        RC(h->inner_light.forward(A, h->M, h->Xh, 128, nullptr, 0, n_hit, true, h->f_in, (void*)side));
    }
    if (!A.dry) join_side(h, side, (hipStream_t)stream);
    LAUNCH(nero_mc_combine_fwd_h(h->pt, h->dirs, h->depth, h->slot, n_miss > 0 ? h->f_out.heads[3] : nullptr, n_hit > 0 ? h->f_in.heads[3] : nullptr,
                                 (n_miss > 0 && c.human_lights && h->n_hum > 0) ? h->f_hum.heads[3] : nullptr, h->hmask, h->n_hum, c.light_exp_max,
                                 c.inner_light_exp_max, h->P, c.diffuse_sample_num, c.specular_sample_num, c.geometry_type, rgb, dl, sl, sp, stream));
    return NERO_OK;
}

int do_shade_bwd(nero_stage2* h, Arena& A, const float* d_rgb, const float* d_dl, const nero_stage2_grads* G, float* d_mat5, void* stream) {
    hipStream_t hs = (hipStream_t)stream;
    const nero_stage2_cfg& c = h->cfg;
    const nero_linear_grad* g = G->lin;
    const int n_miss = h->n_miss, n_hit = h->n_hit, P = h->P, Dd = c.diffuse_sample_num, Ds = c.specular_sample_num;
    const int rm = rpad(n_miss) > 64 ? rpad(n_miss) : 64, rh = rpad(n_hit) > 64 ? rpad(n_hit) : 64;
    const int n_hum = h->n_hum;
    const bool hum = c.human_lights && n_miss > 0 && n_hum > 0;
    const int ru = rpad(n_hum) > 64 ? rpad(n_hum) : 64;
    float* d_hr = c.human_lights ? A.f32((size_t)ru * 4) : nullptr;
    float* d_or = A.f32((size_t)rm * 4);
    float* d_ir = A.f32((size_t)rh * 4);
    float* d_w = A.f32((size_t)P * Ds * 3);
    const int mx = n_miss > n_hit ? n_miss : n_hit;
    float* partials = A.f32((size_t)nero_dw_workspace_floats(mx > 1 ? mx : 1));
    const bool two = h->n_streams >= 2;
    float* partials_h = two ? A.f32((size_t)nero_dw_workspace_floats(n_hit > 1 ? n_hit : 1)) : partials;     // (the hit branch's own partial sums)
    float* dXm = n_miss > 0 ? A.f32((size_t)rpad(n_miss) * h->kout) : nullptr;
    float* dXh = n_hit > 0 ? A.f32((size_t)rpad(n_hit) * 128) : nullptr;
    float* dXhum = hum ? A.f32((size_t)rpad(n_hum) * 24) : nullptr;
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_bwd: workspace too small");
    if (!A.dry) {
        if (d_hr) (void)hipMemsetAsync(d_hr, 0, (size_t)ru * 16, hs);
        (void)hipMemsetAsync(d_or, 0, (size_t)rm * 16, hs);
        (void)hipMemsetAsync(d_ir, 0, (size_t)rh * 16, hs);
        (void)hipMemsetAsync(d_w, 0, (size_t)P * Ds * 12, hs);
    }
    LAUNCH(nero_mc_combine_bwd_h(h->pt, h->dirs, h->depth, h->slot, n_miss > 0 ? h->f_out.heads[3] : nullptr, n_hit > 0 ? h->f_in.heads[3] : nullptr,
                                 hum ? h->f_hum.heads[3] : nullptr, h->hmask, n_hum, c.light_exp_max, c.inner_light_exp_max, P, Dd, Ds, c.geometry_type,
                                 d_rgb, d_dl, d_or, d_ir, d_hr, d_mat5, d_w, stream));
    const float* hd[MAXL] = {};
    // the hit branch (inner light) starts behind the estimator's backward, beside the miss branch
    const hipStream_t side = (n_miss > 0 && n_hit > 0) ? fork_side(h, A, hs) : hs;
    if (n_miss > 0) {
        size_t mk = A.mark();
        predictor_grads(h->outer_light, g + M_OUTER, h->kout);
        hd[3] = d_or;
        Bwd ob;
        RC(h->outer_light.backward(A, h->M, h->f_out, n_miss, nullptr, 0, hd, true, false, nullptr, dXm, h->kout, false, false, ob, stream));
        RC(h->outer_light.weight_grads(A, h->M, h->f_out, ob, n_miss, h->Xm, h->kout, nullptr, 0, hd, nullptr, nullptr, partials, stream));
        if (!two) A.release(mk);                           // (concurrent branches carve on while these deltas are still read)
        if (hum) {
            predictor_grads(h->human_light, g + M_HUMAN, 24);
            hd[3] = d_hr;
            Bwd hb;
            RC(h->human_light.backward(A, h->M, h->f_hum, n_hum, nullptr, 0, hd, true, false, nullptr, dXhum, 24, false, false, hb, stream));
            RC(h->human_light.weight_grads(A, h->M, h->f_hum, hb, n_hum, h->Xhum, 24, nullptr, 0, hd, nullptr, nullptr, partials, stream));
            if (!two) A.release(mk);
        }
    }
    if (n_hit > 0) {
        const size_t mk = A.mark();
        predictor_grads(h->inner_light, g + M_INNER, 123);
        const float* hdh[MAXL] = {};
        hdh[3] = d_ir;
        Bwd ib;
        RC(h->inner_light.backward(A, h->M, h->f_in, n_hit, nullptr, 0, hdh, true, false, nullptr, dXh, 128, false, false, ib, (void*)side));
        RC(h->inner_light.weight_grads(A, h->M, h->f_in, ib, n_hit, h->Xh, 128, nullptr, 0, hdh, nullptr, nullptr, partials_h, (void*)side));
        if (!two) A.release(mk);
    }
    if (!A.dry) join_side(h, side, hs);
    LAUNCH(nero_mc_dir_bwd_h(h->pt, h->dirs, h->fnrm, h->slot, h->tab_s, dXm, dXh, d_w, P, Dd, Ds, d_mat5, c.sphere_direction, dXhum, h->poses, n_hum, stream));
    return nero_check_launch("nero_stage2_shade_bwd");
}

}  // namespace

extern "C" {

int nero_stage2_create(const nero_stage2_cfg* cfg, nero_stage2** out) {
    if (!cfg || !out || cfg->diffuse_sample_num < 1 || cfg->specular_sample_num < 1) return nero_fail(NERO_ERR_ARG, "nero_stage2_create: bad argument");
    if (cfg->geometry_type != 0 && cfg->geometry_type != 1) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_stage2_create: geometry_type must be 0 (schlick) or 1 (ggx_smith)");
    if (!is_f16(cfg->gemm_fwd) || cfg->gemm_bwd != NERO_GEMM_F16X3 || !is_f16(cfg->gemm_dw))
        return nero_fail(NERO_ERR_UNSUPPORTED, "nero_stage2_create: the C-level driver packs fp16 two-plane operands only (F16X3)");
    nero_stage2* h = new (std::nothrow) nero_stage2();
    if (!h) return nero_fail(NERO_ERR_ARG, "nero_stage2_create: out of host memory");
    h->cfg = *cfg;
    h->M = {cfg->gemm_fwd, NERO_GEMM_F16X3, cfg->gemm_bwd, cfg->gemm_dw};     // (validated above: NERO_GEMM_F16X3; the tangent slot is unused in Stage II)
    nero_stage2_weights zero;
    memset(&zero, 0, sizeof(zero));
    build_chains(h, &zero);
    const char* e = getenv("NERO_STREAMS");
    h->n_streams = e ? atoi(e) : NERO_STREAMS_DEFAULT;
    const char* sd = getenv("NERO_MC_SKIP_DEAD");
    h->skip_dead = !(sd && atoi(sd) == 0);
    if (hipGetDevice(&h->device) != hipSuccess) { h->device = -1; (void)hipGetLastError(); }
    if (h->n_streams >= 2) {
        if (hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();             // (no device in reach -- the size queries still work -- or out of handles: one stream)
            h->s2 = nullptr;
        }
    }
    *out = h;
    return NERO_OK;
}

void nero_stage2_destroy(nero_stage2* h) {
    if (!h) return;
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->s2) (void)hipStreamDestroy(h->s2);
    delete h;
}

size_t nero_stage2_pack_bytes(nero_stage2* h) { return h ? pack_floats_total(h) * 4 : 0; }

int nero_stage2_pack(nero_stage2* h, const nero_stage2_weights* w, void* pack_buf, void* stream) {
    if (!h || !w || !pack_buf) return nero_fail(NERO_ERR_ARG, "nero_stage2_pack: bad argument");
    build_chains(h, w);
    (void)hipMemsetAsync(pack_buf, 0, pack_floats_total(h) * 4, (hipStream_t)stream);
    std::vector<nero_pack_job> jobs;
    float* p = static_cast<float*>(pack_buf);
    for (Chain* c : all_chains(h)) {
        float* q = p;
        c->pack(q, jobs);
        p += (c->pack_floats() + 63) / 64 * 64;
    }
    for (size_t i0 = 0; i0 < jobs.size(); i0 += NERO_MAX_PACK_JOBS) {
        const int n = (int)(jobs.size() - i0 < NERO_MAX_PACK_JOBS ? jobs.size() - i0 : NERO_MAX_PACK_JOBS);
        RC(nero_pack_batch(jobs.data() + i0, n, stream));
    }
    h->packed = true;
    return NERO_OK;
}

// worst case over the hit / miss split of the P * D light rays, for n_pred rows through predict_materials.  The carve is linear in the
// two partitions' row counts AFTER each is padded to whole 64-row tiles, so the maximum sits at an end of the split (all miss / all hit)
// or one ray short of it (N - 1 / 1: both partitions padded -- what decides for a handful of points; tests/test_edge_cases.py).
size_t nero_stage2_workspace_bytes(nero_stage2* h, int n_pred, int P) {
    if (!h || n_pred < 0 || P < 0) return 0;
    const int N = P * (h->cfg.diffuse_sample_num + h->cfg.specular_sample_num);
    size_t worst = 0;
    const int hits[4] = {0, N, N > 1 ? 1 : 0, N > 1 ? N - 1 : N};
    for (int k = 0; k < 4; ++k) {
        nero_stage2 tmp = *h;
        Arena& A = tmp.A;
        A = Arena();
        A.dry = true;
        nero_stage2_grads G;
        for (int i = 0; i < NERO_S2_LINEARS; ++i) { G.lin[i].dW = reinterpret_cast<float*>(0x1000); G.lin[i].db = reinterpret_cast<float*>(0x1000); }
        (void)do_predict_fwd(&tmp, A, nullptr, n_pred, nullptr, nullptr);
        tmp.P = P;
        tmp.pt = A.f32((size_t)P * 32);
        tmp.slot = A.i32(N); tmp.miss_idx = A.i32(N); tmp.hit_idx = A.i32(N); tmp.counts = A.i32(4);
        (void)A.i32((N + 3) / 4);                                      // the dead-ray flags (bytes)
        if (tmp.cfg.human_lights) (void)A.i32((N + 3) / 4);            // the human-plane flags (bytes)
        (void)A.i32(nero_mc_split_tmp_ints(N));
        tmp.n_hit = hits[k]; tmp.n_miss = N - hits[k]; tmp.n_hum = tmp.n_miss;
        (void)do_shade_lights(&tmp, A, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        const size_t mk = A.mark();
        (void)do_shade_bwd(&tmp, A, nullptr, nullptr, &G, nullptr, nullptr);
        A.release(mk);
        (void)do_predict_bwd(&tmp, A, nullptr, &G, nullptr);
        worst = A.peak > worst ? A.peak : worst;
    }
    return worst + 65536;
}

int nero_stage2_predict_fwd(nero_stage2* h, const float* x, int n, float* raw5, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !h->packed || !x || !raw5 || !ws || n <= 0) return nero_fail(NERO_ERR_ARG, "nero_stage2_predict_fwd: bad argument (pack the weights first)");
    Arena& A = h->A;
    A = Arena();
    A.base = static_cast<char*>(ws); A.cap = ws_bytes;
    RC(do_predict_fwd(h, A, x, n, raw5, stream));
    h->predict_mark = h->shade_mark = A.mark();
    h->P = 0;
    return nero_check_launch("nero_stage2_predict_fwd");
}

int nero_stage2_rays(nero_stage2* h, int P, const float* pts, const float* view, const float* normals, const float* mat5, const float* rand_d,
                     const float* rand_s, const float* tab_d, const float* tab_s, float* origins, float* dirs, void* stream) {
    if (!h || !h->A.base || !pts || !view || !normals || !mat5 || !tab_d || !tab_s || !origins || !dirs || P <= 0)
        return nero_fail(NERO_ERR_ARG, "nero_stage2_rays: bad argument (call nero_stage2_predict_fwd first: it opens the step's workspace)");
    Arena& A = h->A;
    A.release(h->predict_mark);
    const nero_stage2_cfg& c = h->cfg;
    const int N = P * (c.diffuse_sample_num + c.specular_sample_num);
    h->P = P;
    h->pt = A.f32((size_t)P * 32);
    h->slot = A.i32(N); h->miss_idx = A.i32(N); h->hit_idx = A.i32(N); h->counts = A.i32(4);
    h->dead = reinterpret_cast<unsigned char*>(A.i32((N + 3) / 4));
    h->hum = c.human_lights ? reinterpret_cast<unsigned char*>(A.i32((N + 3) / 4)) : nullptr;
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_rays: workspace too small");
    RC(nero_mc_point_setup(pts, view, normals, mat5, rand_d, rand_s, P, h->pt, stream));
    RC(nero_mc_dirs(h->pt, tab_d, tab_s, P, c.diffuse_sample_num, c.specular_sample_num, dirs, origins, stream));
    if (h->skip_dead && c.geometry_type == 0)
        RC(nero_mc_dead_rays(h->pt, dirs, P, c.diffuse_sample_num, c.specular_sample_num, c.geometry_type, h->dead, stream));
    h->dirs = dirs; h->tab_s = tab_s;
    h->shade_mark = A.mark();
    return NERO_OK;
}

int nero_stage2_counts(nero_stage2* h, int* n_miss, int* n_hit, int* n_hum) {
    if (!h) return nero_fail(NERO_ERR_ARG, "nero_stage2_counts: bad argument");
    if (n_miss) *n_miss = h->n_miss;
    if (n_hit) *n_hit = h->n_hit;
    if (n_hum) *n_hum = h->cfg.human_lights ? h->n_hum : 0;
    return NERO_OK;
}

const unsigned char* nero_stage2_dead_rays(nero_stage2* h) {
    return (h && h->P > 0 && h->skip_dead && h->cfg.geometry_type == 0) ? h->dead : nullptr;
}

int nero_stage2_shade_fwd(nero_stage2* h, const float* pos, const float* face_normals, const float* depth, const float* poses, float* rgb,
                          float* dl, float* sl, float* sp, int* n_miss_out, int* n_hit_out, void* stream) {
    if (!h || !h->P || !pos || !face_normals || !depth || !rgb || !dl || !sl || !sp) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: bad argument (nero_stage2_rays first)");
    if (h->cfg.human_lights && !poses) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: human_lights needs poses [P,3,4]");
    if (wrong_device(h)) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: the current device is not the one the handle was created on");
    Arena& A = h->A;
    A.release(h->shade_mark);
    const int N = h->P * (h->cfg.diffuse_sample_num + h->cfg.specular_sample_num);
    int* tmp = A.i32(nero_mc_split_tmp_ints(N));
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_fwd: workspace too small");
    const unsigned char* dead = (h->skip_dead && h->cfg.geometry_type == 0) ? h->dead : nullptr;
    const bool part = h->skip_dead && h->cfg.human_lights && h->hum;       // (the miss list partitioned by the human-plane mask)
    if (part) RC(nero_mc_human_flags(h->pt, h->dirs, poses, h->P, h->cfg.diffuse_sample_num + h->cfg.specular_sample_num, h->hum, stream));
    RC(nero_mc_split_classes(depth, dead, part ? h->hum : nullptr, N, h->slot, h->miss_idx, h->hit_idx, h->counts, tmp, stream));
    int counts[3] = {0, 0, 0};
    if (hipMemcpyAsync(counts, h->counts, 12, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return nero_fail(NERO_ERR_LAUNCH, "nero_stage2_shade_fwd: reading the ray counts failed");
    h->n_miss = counts[0]; h->n_hit = counts[1];
    h->n_hum = part ? counts[2] : counts[0];
    if (n_miss_out) *n_miss_out = counts[0];
    if (n_hit_out) *n_hit_out = counts[1];
    h->depth = depth; h->fnrm = face_normals; h->poses = poses;
    RC(drain_on_error(h, do_shade_lights(h, A, pos, rgb, dl, sl, sp, stream)));
    h->shade_mark = A.mark();
    return nero_check_launch("nero_stage2_shade_fwd");
}

int nero_stage2_shade_bwd(nero_stage2* h, const float* d_rgb, const float* d_dl, const nero_stage2_grads* grads, float* d_mat5, void* stream) {
    if (!h || !h->P || !d_rgb || !grads || !d_mat5) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_bwd: bad argument (no forward state)");
    Arena& A = h->A;
    const size_t mk = h->shade_mark;
    A.release(mk);
    if (wrong_device(h)) return nero_fail(NERO_ERR_ARG, "nero_stage2_shade_bwd: the current device is not the one the handle was created on");
    const int rc = drain_on_error(h, do_shade_bwd(h, A, d_rgb, d_dl, grads, d_mat5, stream));
    A.release(mk);
    return rc;
}

int nero_stage2_predict_bwd(nero_stage2* h, const float* d_raw5, const nero_stage2_grads* grads, void* stream) {
    if (!h || !h->n_pred || !d_raw5 || !grads) return nero_fail(NERO_ERR_ARG, "nero_stage2_predict_bwd: bad argument (no forward state)");
    // the predict state sits BELOW the shading state in the arena; its temporaries go above whatever is live
    Arena& A = h->A;
    const size_t mk = A.mark() > h->shade_mark ? A.mark() : h->shade_mark;
    A.release(mk);
    const int rc = do_predict_bwd(h, A, d_raw5, grads, stream);
    A.release(mk);
    return rc;
}

}  // extern "C"
