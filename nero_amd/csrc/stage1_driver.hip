// stage1_driver.hip -- the C-level driver of the Stage-I render step (nero_stage1_render_fwd / _bwd,
// nero_stage1_workspace_bytes).  Host code only, plus four trivial kernels: it sequences the library's own entry points (chain
// kernels, sampler, shader algebra, compositing, weight-gradient GEMMs) exactly as nero_amd/shape_step.py does, so that a host in
// ANY language can run sample_ray + render_core + their backward (network/renderer.py:403-443, 445-463, 550-606 and autograd's
// double backward through SDFNetwork.gradient, network/field.py:155-167) with one call each and no Python in between.
//
// Memory: the caller hands over ONE workspace (size from nero_stage1_workspace_bytes); every intermediate -- saved activations,
// deltas, encodings, partial sums -- is carved from it by a bump allocator with stack-style release, nothing is hipMalloc'ed.  The
// forward keeps its state there for the backward; nero_stage1_get_state exposes the pieces the loss needs (occlusion-loss march,
// validation extras).  The only host synchronisation is the read-back of the inner / outer sample counts after render_prep
// (they size every later launch), as in the Python driver.
#include "chain_host.h"
#include <stdlib.h>
#ifndef NERO_STREAMS_DEFAULT
#define NERO_STREAMS_DEFAULT 3
#endif


// ---- the handle -----------------------------------------------------------------------------------------------------------------
enum { L_SDF = 0, L_NERF_PTS = 9, L_NERF_VIEWS = 17, L_NERF_FEATURE = 18, L_NERF_ALPHA = 19, L_NERF_RGB = 20, L_METALLIC = 21,
       L_ROUGHNESS = 25, L_ALBEDO = 29, L_OUTER = 33, L_INNER = 37, L_WEIGHT = 41, L_HUMAN = 45 };
constexpr int D_PE = 39, LD_PE = 40, N_FREQ = 6;

struct nero_stage1 {
    nero_stage1_cfg cfg;
    Modes M;
    int ld_outer = 72;
    bool packed = false;
    Chain sdf_full, sdf_value, nerf_trunk, nerf_head, mat[3], outer_light, inner_light, inner_weight, human_light;
    // ---- state of the current step (pointers into the caller's workspace) ----
    Arena A;
    nero_stage1_state st;
    size_t step_mark = 0;
    // forward state kept for the backward
    Fwd f_trunk, f_head, f_sdf, f_mat[3], f_out, f_in, f_w, f_h;
    Bwd b_normal;
    float *pe88 = nullptr, *pev32 = nullptr, *dist_o = nullptr, *pe40 = nullptr, *x8 = nullptr, *mat8 = nullptr, *Xo2 = nullptr, *Xi = nullptr,
          *Xo = nullptr, *Xh = nullptr, *hmask = nullptr, *alphaRT = nullptr, *colorRT = nullptr;
    const float *o = nullptr, *d = nullptr, *variance = nullptr, *lut = nullptr, *poses = nullptr;
    float anneal = 0.f;
    // ---- a second stream for the NeRF++ (outer-sample) branch (round 4): it shares no intermediate with the SDF / shading branch between
    // the compaction and the compositing (forward), resp. between the compositing backward and the end of the step (backward), so the
    // two run concurrently: one branch's partial last rounds of workgroups and its ~300 kernel boundaries are filled by the other's
    // work.  NERO_STREAMS=1 restores the single-stream order (also used while launches are timed: nero_prof_enable).
    int n_streams = 1;
    int device = -1;                                   // the device the private streams / events were created on (nero_stage1_create)
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // ---- a third stream for the weight-gradient jobs of the SDF / shading branch (round 5, NERO_STREAMS=3): the jobs of a chain depend on
    // that chain's reverse pass only and nothing in the step depends on them, so they run BESIDE the next chain's reverse pass.  The two
    // kernel classes are bound by different things -- the weight-gradient GEMM streams its operands from HBM (4.9 TB/s), the chain kernels
    // issue MFMA + VALU -- and their workgroups interleave over the CUs: fewer weight-gradient workgroups at a time compete for HBM.  The
    // price is memory: a chain's deltas cannot be released while its jobs may still read them (no arena release in this mode).
    // debug: (pointer, bytes) of the backward's intermediates of the last step, for scripts/r05/dbg_streams.py (nero_stage1_debug_buffers)
    const void* dbg_ptr[12] = {};
    size_t dbg_bytes[12] = {};
    hipStream_t s3 = nullptr;
    static constexpr int N_DW_EV = 12;                     // one event per fork of a step (never re-recorded while a wait on it may be pending)
    hipEvent_t ev_dw[N_DW_EV] = {}, ev_dw_done = nullptr;
    int dw_fork = 0;
};

namespace {

// the stream of the side branch: s2 behind everything `main` holds so far -- or `main` itself (dry runs, one-stream mode, timed launches)
hipStream_t fork_side(nero_stage1* h, const Arena& A, hipStream_t main) {
    if (A.dry || h->n_streams < 2 || !h->s2 || nero_prof_is_on()) return main;
    (void)hipEventRecord(h->ev_fork, main);
    (void)hipStreamWaitEvent(h->s2, h->ev_fork, 0);
    return h->s2;
}
void join_side(nero_stage1* h, hipStream_t side, hipStream_t main) {
    if (side == main) return;
    (void)hipEventRecord(h->ev_join, side);
    (void)hipStreamWaitEvent(main, h->ev_join, 0);
}

// the stream of the weight-gradient jobs: s3 behind everything `main` holds so far (the reverse pass that produced their operands)
hipStream_t fork_dw(nero_stage1* h, const Arena& A, hipStream_t main) {
    if (A.dry || h->n_streams < 3 || !h->s3 || nero_prof_is_on()) return main;
    hipEvent_t ev = h->ev_dw[h->dw_fork++ % nero_stage1::N_DW_EV];
    (void)hipEventRecord(ev, main);
    (void)hipStreamWaitEvent(h->s3, ev, 0);
    return h->s3;
}
void join_dw(nero_stage1* h, const Arena& A, hipStream_t main) {
    if (A.dry || h->n_streams < 3 || !h->s3 || nero_prof_is_on()) return;
    (void)hipEventRecord(h->ev_dw_done, h->s3);
    (void)hipStreamWaitEvent(main, h->ev_dw_done, 0);
    h->dw_fork = 0;
}

// A call that fails between a fork and its join would leave work queued on the private streams with nothing ordering it in front of
// whatever the caller does next with the workspace: every failing exit of the entry points drains them (host-blocking, error path only).
int drain_on_error(nero_stage1* h, int rc) {
    if (rc == NERO_OK) return rc;
    if (h->s2) (void)hipStreamSynchronize(h->s2);
    if (h->s3) (void)hipStreamSynchronize(h->s3);
    h->dw_fork = 0;
    (void)hipGetLastError();
    return rc;
}
// the private streams belong to the device that was current at create time: a call under another current device would fork onto it
bool wrong_device(const nero_stage1* h) {
    int dev = -1;
    return (h->s2 || h->s3) && hipGetDevice(&dev) == hipSuccess && h->device >= 0 && dev != h->device;
}

void build_chains(nero_stage1* h, const nero_stage1_weights* w) {
    const nero_linear* L = w->lin;
    // SDF (nero_amd/sdf.py::sdf_entries): PE-6 -> 9 layers, skip into layer 4 with 1/sqrt2 folded in, last layer = [head row 0 | dense rows 1..256]
    Chain s;
    for (int l = 0; l < 9; ++l) {
        if (l == 0) s.e.push_back(dense(L[L_SDF], D_PE, 256, NERO_ACT_SOFTPLUS100, D_PE));
        else if (l == 3) s.e.push_back(dense(L[L_SDF + 3], 256, 256 - D_PE, NERO_ACT_SOFTPLUS100, 256));
        else if (l == 4) s.e.push_back(dense(L[L_SDF + 4], 256, 256, NERO_ACT_SOFTPLUS100, 256 - D_PE, 0, D_PE, 256 - D_PE, (float)(1.0 / sqrt(2.0))));
        else if (l == 8) {
            nero_linear rows1 = {L[L_SDF + 8].W + 256, L[L_SDF + 8].b + 1};
            Entry x = dense(rows1, 256, 256, NERO_ACT_NONE, 256);
            x.h.has = true; x.h.W = L[L_SDF + 8].W; x.h.ldw = 256; x.h.b = L[L_SDF + 8].b; x.h.n_head = 1; x.h.k = 256;
            s.e.push_back(x);
        } else s.e.push_back(dense(L[L_SDF + l], 256, 256, NERO_ACT_SOFTPLUS100, 256));
    }
    s.k_init = LD_PE; s.k_aux = LD_PE;
    h->sdf_full = s;
    // NeRF++ trunk: 8 ReLU layers, layer 5 takes [pe(84) | h(256)] (network/field.py:239-241, 265-269), sigma head on the trunk output
    Chain t;
    for (int i = 0; i < 8; ++i) {
        if (i == 0) t.e.push_back(dense(L[L_NERF_PTS], 84, 256, NERO_ACT_RELU, 84));
        else if (i == 5) t.e.push_back(dense(L[L_NERF_PTS + 5], 340, 256, NERO_ACT_RELU, 256, 84, 84, 0));
        else t.e.push_back(dense(L[L_NERF_PTS + i], 256, 256, NERO_ACT_RELU, 256));
    }
    t.e.push_back(head_only(L[L_NERF_ALPHA], 256, 1, 256));
    t.k_init = 88; t.k_aux = 88; t.aux_wide = 1;
    h->nerf_trunk = t;
    Chain hd;
    hd.e.push_back(dense(L[L_NERF_FEATURE], 256, 256, NERO_ACT_NONE, 256));
    hd.e.push_back(dense(L[L_NERF_VIEWS], 283, 128, NERO_ACT_RELU, 256, 0, 27, 256));
    hd.e.push_back(head_only(L[L_NERF_RGB], 128, 3, 128));
    hd.k_init = 256; hd.k_aux = 32;
    h->nerf_head = hd;
    h->mat[0] = predictor(L + L_METALLIC, 256, 3, 256, 8, 1);
    h->mat[1] = predictor(L + L_ROUGHNESS, 256, 3, 256, 8, 1);
    h->mat[2] = predictor(L + L_ALBEDO, 256, 3, 256, 8, 3);
    h->ld_outer = h->cfg.sphere_direction ? 144 : 72;
    h->outer_light = predictor(L + L_OUTER, h->ld_outer, 0, h->ld_outer, 0, 3);
    h->inner_light = predictor(L + L_INNER, 123, 0, 128, 0, 3);
    h->inner_weight = predictor(L + L_WEIGHT, 90, 0, 96, 0, 1);
    if (h->cfg.human_light) h->human_light = predictor(L + L_HUMAN, 24, 0, 24, 0, 4);
}

// value-only SDF chain shares the images of the full one: entries 0..7 + the head of entry 8 as a pseudo layer
void make_value_chain(nero_stage1* h) {
    Chain v;
    for (int l = 0; l < 8; ++l) v.e.push_back(h->sdf_full.e[l]);
    Entry x;
    x.h = h->sdf_full.e[8].h;
    x.hw = h->sdf_full.e[8].hw; x.hb = h->sdf_full.e[8].hb;
    v.e.push_back(x);
    v.k_init = LD_PE; v.k_aux = LD_PE;
    h->sdf_value = v;
}

std::vector<Chain*> all_chains(nero_stage1* h) {
    std::vector<Chain*> v = {&h->sdf_full, &h->nerf_trunk, &h->nerf_head, &h->outer_light, &h->inner_light, &h->inner_weight};
    if (h->cfg.human_light) v.push_back(&h->human_light);
    v.push_back(&h->mat[0]); v.push_back(&h->mat[1]); v.push_back(&h->mat[2]);
    return v;
}

size_t pack_floats_total(nero_stage1* h) {
    size_t t = 0;
    for (Chain* c : all_chains(h)) t += (c->pack_floats() + 63) / 64 * 64;
    return t;
}

// no-grad SDF values of PE rows: -> heads [rows_pad, 4], column 0 = sdf   (SDFField.sdf_from_pe)
int sdf_from_pe(nero_stage1* h, Arena& A, const float* pe, int n, float*& out4, void* stream) {
    Fwd F;
    RC(h->sdf_value.forward(A, h->M, pe, LD_PE, pe, LD_PE, n, false, F, stream));
    out4 = F.heads[8];
    return NERO_OK;
}

// ---- sample_ray (network/renderer.py:403-443; nero_amd/shape_step.py::sample_ray) ---------------------------------------------------
int do_sample(nero_stage1* h, Arena& A, int R, const float* o, const float* d, const float* near, const float* far, const float* variance,
              const float* rand1, const float* rand_bg, float* z, void* stream) {
    const nero_stage1_cfg& c = h->cfg;
    const int ns = c.n_samples, nb = c.n_bg_samples, up = c.up_sample_steps, m = c.n_importance / up;
    const int n_in = ns + m * up, T = n_in + nb;
    float* tab = A.f32((size_t)R * n_in);
    LAUNCH(nero_coarse_z(near, far, rand1, R, ns, z, T, stream));
    float* pe = A.f32((size_t)rpad(R * ns) * LD_PE);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_sample: workspace too small");
    LAUNCH(nero_ray_points_pe(o, d, z, T, 0, ns, R, pe, stream));
    float* s4 = nullptr;
    RC(sdf_from_pe(h, A, pe, R * ns, s4, stream));
    LAUNCH(nero_scatter_sdf(s4, 4, R, ns, tab, n_in, stream));
    int n = ns;
    float* z_new = A.f32((size_t)R * m);
    float* pe_new = A.f32((size_t)rpad(R * m) * LD_PE);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_sample: workspace too small");
    const float* var_ptr = c.clip_sample_variance ? variance : nullptr;
    for (int i = 0; i < up; ++i) {
        LAUNCH(nero_upsample(o, d, z, T, tab, n_in, n, var_ptr, 64.0f * (float)(1 << i), m, R, z_new, nullptr, nullptr, stream));
        if (i + 1 < up) {
            const size_t mk = A.mark();
            LAUNCH(nero_ray_points_pe(o, d, z_new, m, 0, m, R, pe_new, stream));
            RC(sdf_from_pe(h, A, pe_new, R * m, s4, stream));
            LAUNCH(nero_merge_sorted(z, T, n, tab, n_in, z_new, m, s4, 4, R, nullptr, stream));
            A.release(mk);
        } else {
            LAUNCH(nero_merge_sorted(z, T, n, nullptr, 0, z_new, m, nullptr, 0, R, nullptr, stream));
        }
        n += m;
    }
    LAUNCH(nero_background_z(far, rand_bg, R, nb, z, T, n_in, stream));
    return NERO_OK;
}

// ---- render_core forward (nero_amd/shape_step.py::RenderCore.forward) ---------------------------------------------------------------
int do_forward(nero_stage1* h, Arena& A, int R, int T, int n_in, int n_out, const float* d, const float* variance, const float* lut,
               const float* poses, float anneal, float* rgb, float* gerr, float* occ_prob, void* stream) {
    nero_stage1_state& S = h->st;
    const Modes& M = h->M;
    hipStream_t hs = (hipStream_t)stream;
    const int rpi = rpad(n_in), rpo = rpad(n_out);
    S.inner_idx = A.i32(n_in > 0 ? n_in : 1);
    S.outer_idx = A.i32(n_out > 0 ? n_out : 1);
    h->alphaRT = A.f32((size_t)R * T);
    h->colorRT = A.f32((size_t)R * T * 3);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
    LAUNCH(nero_compact(S.pts4, S.ray_off, R, T, S.inner_idx, S.outer_idx, stream));
    if (!A.dry) {
        (void)hipMemsetAsync(h->alphaRT, 0, (size_t)R * T * 4, hs);
        (void)hipMemsetAsync(h->colorRT, 0, (size_t)R * T * 12, hs);
    }
    const bool two = h->n_streams >= 2;                // (memory decisions follow the handle, not the moment: the size query is a dry run)
    hipStream_t so_ = hs;
    if (n_out > 0) {
        so_ = fork_side(h, A, hs);                     // the NeRF++ branch: behind the compaction and the two memsets
        void* so = (void*)so_;
        h->pe88 = A.f32((size_t)rpo * 88); h->pev32 = A.f32((size_t)rpo * 32); h->dist_o = A.f32(rpo);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_gather_outer(S.pts4, d, S.outer_idx, T, n_out, h->pe88, h->pev32, h->dist_o, so));
        h->f_trunk = Fwd(); h->f_head = Fwd();
        RC(h->nerf_trunk.forward(A, M, h->pe88, 88, h->pe88, 88, n_out, true, h->f_trunk, so));
        RC(h->nerf_head.forward(A, M, h->f_trunk.saves[7], NERO_HID, h->pev32, 32, n_out, true, h->f_head, so));
        const size_t mk = A.mark();
        float* alpha_o = A.f32(rpo);
        float* color_o = A.f32((size_t)rpo * 3);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_nerf_head_fwd(h->f_trunk.heads[8], h->f_head.heads[2], h->dist_o, n_out, alpha_o, color_o, so));
        LAUNCH(nero_scatter_samples(alpha_o, color_o, S.outer_idx, n_out, h->alphaRT, h->colorRT, so));
        if (!two) A.release(mk);                       // (concurrent branches: the other one carves on while these are still read)
    }
    if (n_in > 0) {
        S.x4 = A.f32((size_t)rpi * 4);
        h->pe40 = A.f32((size_t)rpi * LD_PE);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_gather_inner(S.pts4, S.inner_idx, n_in, S.x4, h->pe40, stream));
        // SDFField.forward_normal: value + feature, then the first-order reverse pass seeded by the sdf row of W_8 = the normal
        h->f_sdf = Fwd();
        RC(h->sdf_full.forward(A, M, h->pe40, LD_PE, h->pe40, LD_PE, n_in, true, h->f_sdf, stream));
        float* ones = A.f32((size_t)rpi * 4);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        if (!A.dry) hipLaunchKernelGGL(ones_col0_kernel, dim3((rpi + 255) / 256), dim3(256), 0, hs, ones, rpi);
        const float* hd[MAXL] = {};
        hd[8] = ones;
        h->b_normal = Bwd();
        RC(h->sdf_full.backward(A, M, h->f_sdf, n_in, nullptr, 0, hd, true, true, nullptr, nullptr, 0, false, true, h->b_normal, stream));
        S.normal = A.f32((size_t)n_in * 3);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_pe_vjp(S.x4, 4, h->b_normal.d_init, h->b_normal.ld_dinit, h->b_normal.d_aux, h->b_normal.ld_daux, N_FREQ, n_in, S.normal, 3, stream));
        S.sdf4 = h->f_sdf.heads[8];
        S.feat = h->f_sdf.saves[8];
        float* alpha_i = A.f32(rpi);
        S.geo = A.f32((size_t)rpi * 8);
        h->x8 = A.f32((size_t)rpi * 8);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_sdf_alpha_fwd(S.sdf4, S.normal, S.x4, S.inner_idx, d, T, variance, anneal, n_in, alpha_i, S.geo, gerr, stream));
        if (!A.dry) hipLaunchKernelGGL(x8_from_x4_kernel, dim3((rpi + 255) / 256), dim3(256), 0, hs, S.x4, h->x8, n_in, rpi);
        for (int j = 0; j < 3; ++j) {
            h->f_mat[j] = Fwd();
            RC(h->mat[j].forward(A, M, S.feat, NERO_HID, h->x8, 8, n_in, true, h->f_mat[j], stream));
        }
        h->mat8 = A.f32((size_t)rpi * 8);
        h->Xo2 = A.f32((size_t)2 * rpi * h->ld_outer);
        h->Xi = A.f32((size_t)rpi * 128);
        h->Xo = A.f32((size_t)rpi * 96);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        LAUNCH(nero_shade_encode(S.x4, S.geo, h->f_mat[0].heads[3], h->f_mat[1].heads[3], h->f_mat[2].heads[3], n_in, h->mat8, h->Xo2,
                                 h->Xo2 + (size_t)rpi * h->ld_outer, h->Xi, h->Xo, h->cfg.sphere_direction, stream));
        h->f_out = Fwd(); h->f_in = Fwd(); h->f_w = Fwd(); h->f_h = Fwd();
        RC(h->outer_light.forward(A, M, h->Xo2, h->ld_outer, nullptr, 0, rpi + n_in, true, h->f_out, stream));
        RC(h->inner_light.forward(A, M, h->Xi, 128, nullptr, 0, n_in, true, h->f_in, stream));
        RC(h->inner_weight.forward(A, M, h->Xo, 96, nullptr, 0, n_in, true, h->f_w, stream));
        h->Xh = h->hmask = nullptr;
        if (h->cfg.human_light) {
            h->Xh = A.f32((size_t)rpi * 24); h->hmask = A.f32(rpi);
            if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
            LAUNCH(nero_human_encode(S.x4, S.geo, h->mat8, S.inner_idx, T, poses, n_in, h->Xh, h->hmask, stream));
            RC(h->human_light.forward(A, M, h->Xh, 24, nullptr, 0, n_in, true, h->f_h, stream));
        }
        const size_t mk = A.mark();
        float* color_i = A.f32((size_t)rpi * 3);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
        const float* Lh = h->f_out.heads[3];
        LAUNCH(nero_shade_combine_fwd(S.geo, h->mat8, Lh, Lh + (size_t)rpi * 4, h->f_in.heads[3], h->f_w.heads[3], lut, h->cfg.light_exp_max, n_in,
                                      color_i, occ_prob, h->cfg.human_light ? h->f_h.heads[3] : nullptr, h->hmask, stream));
        LAUNCH(nero_scatter_samples(alpha_i, color_i, S.inner_idx, n_in, h->alphaRT, h->colorRT, stream));
        A.release(mk);
    }
    S.weights = A.f32((size_t)R * T);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
    if (!A.dry) join_side(h, so_, hs);                 // both branches have scattered their alpha / colour
    LAUNCH(nero_composite_fwd(h->alphaRT, h->colorRT, R, T, S.weights, rgb, stream));
    return NERO_OK;
}

// ---- render_core backward (RenderCore.backward + SDFField.backward) ------------------------------------------------------------------
int do_backward(nero_stage1* h, Arena& A, const float* d_rgb, const float* d_gerr, const float* d_occ, const nero_stage1_grads* G,
                float* d_inv_s_sum, void* stream) {
    nero_stage1_state& S = h->st;
    const Modes& M = h->M;
    hipStream_t hs = (hipStream_t)stream;
    const int R = S.R, T = S.T, n_in = S.n_in, n_out = S.n_out;
    const int rpi = rpad(n_in), rpo = rpad(n_out);
    const nero_linear_grad* g = G->lin;
    float* d_aRT = A.f32((size_t)R * T);
    float* d_cRT = A.f32((size_t)R * T * 3);
    const int ws_rows = (n_in + rpi) > n_out ? (n_in + rpi) : n_out;
    float* partials = A.f32((size_t)nero_dw_workspace_floats(ws_rows > 1 ? ws_rows : 1));
    const bool two = h->n_streams >= 2;
    const bool three = h->n_streams >= 3;              // (weight-gradient jobs of the inner branch on their own stream: no delta is released)
    float* partials_o = two ? A.f32((size_t)nero_dw_workspace_floats(n_out > 1 ? n_out : 1)) : partials;      // (the side branch's own partial sums)
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
    LAUNCH(nero_composite_bwd(h->alphaRT, h->colorRT, S.weights, d_rgb, R, T, d_aRT, d_cRT, stream));
    hipStream_t so_ = hs;
    if (n_out > 0) {
        so_ = fork_side(h, A, hs);                     // the NeRF++ branch of the backward: behind the compositing backward
        void* const main_stream = stream;
        stream = (void*)so_;                           // (the block below issues everything on `stream`)
        float* const main_partials = partials;
        partials = partials_o;
        const size_t mk = A.mark();
        float* d_ao = A.f32(rpo);
        float* d_co = A.f32((size_t)rpo * 3);
        float* d_sig4 = A.f32((size_t)rpo * 4);
        float* d_rgb4 = A.f32((size_t)rpo * 4);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
        LAUNCH(nero_gather_sample_grads(d_aRT, d_cRT, S.outer_idx, n_out, d_ao, d_co, stream));
        LAUNCH(nero_nerf_head_bwd(h->f_trunk.heads[8], h->f_head.heads[2], h->dist_o, n_out, d_ao, d_co, d_sig4, d_rgb4, stream));
        Chain& hc = h->nerf_head;
        set_dense_grad(hc.e[0], g[L_NERF_FEATURE], 256);
        set_dense_grad(hc.e[1], g[L_NERF_VIEWS], 283);
        set_head_grad(hc.e[2], g[L_NERF_RGB], 128);
        const float* hd[MAXL] = {};
        hd[2] = d_rgb4;
        Bwd hb;
        RC(hc.backward(A, M, h->f_head, n_out, nullptr, 0, hd, true, false, nullptr, nullptr, 0, false, false, hb, stream));
        RC(hc.weight_grads(A, M, h->f_head, hb, n_out, h->f_trunk.saves[7], NERO_HID, h->pev32, 32, hd, nullptr, nullptr, partials, stream));
        Chain& tc = h->nerf_trunk;
        for (int i = 0; i < 8; ++i) set_dense_grad(tc.e[i], g[L_NERF_PTS + i], i == 0 ? 84 : (i == 5 ? 340 : 256));
        set_head_grad(tc.e[8], g[L_NERF_ALPHA], 256);
        const float* td[MAXL] = {};
        td[8] = d_sig4;
        Bwd tb;
        RC(tc.backward(A, M, h->f_trunk, n_out, hb.d_init, hb.ld_dinit, td, false, false, nullptr, nullptr, 0, false, false, tb, stream));
        RC(tc.weight_grads(A, M, h->f_trunk, tb, n_out, h->pe88, 88, h->pe88, 88, td, nullptr, nullptr, partials, stream));
        if (!two) A.release(mk);
        stream = main_stream;
        partials = main_partials;
    }
    if (n_in > 0) {
        float* d_ai = A.f32(rpi);
        float* d_ci = A.f32((size_t)rpi * 3);
        float* dLh = A.f32((size_t)2 * rpi * 4);
        float* dLi = A.f32((size_t)rpi * 4);
        float* dLo = A.f32((size_t)rpi * 4);
        float* dmat = A.f32((size_t)rpi * 8);
        float* d_geo = A.f32((size_t)rpi * 8);
        float* dLhum = h->cfg.human_light ? A.f32((size_t)rpi * 4) : nullptr;
        float* d_feat = A.f32((size_t)rpi * NERO_HID);
        float* dmr = A.f32((size_t)rpi * 4);
        float* drr = A.f32((size_t)rpi * 4);
        float* dar = A.f32((size_t)rpi * 4);
        float* extra = h->cfg.human_light ? A.f32((size_t)rpi * 4) : nullptr;
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
        LAUNCH(nero_gather_sample_grads(d_aRT, d_cRT, S.inner_idx, n_in, d_ai, d_ci, stream));
        if (!A.dry) (void)hipMemsetAsync(d_geo, 0, (size_t)rpi * 32, hs);
        const float* Lh = h->f_out.heads[3];
        LAUNCH(nero_shade_combine_bwd(S.geo, h->mat8, Lh, Lh + (size_t)rpi * 4, h->f_in.heads[3], h->f_w.heads[3], h->lut, h->cfg.light_exp_max, n_in,
                                      d_ci, d_occ, dLh, dLh + (size_t)rpi * 4, dLi, dLo, dmat, d_geo,
                                      h->cfg.human_light ? h->f_h.heads[3] : nullptr, h->hmask, dLhum, stream));
        const int n2 = rpi + n_in;
        const float* dX_outer = nullptr; int ld_dxo = 0;
        const float* dX_inner = nullptr;
        {   // light MLPs: reverse passes for the input gradients + weight gradients; their deltas are released right away, the
            // input gradients stay until the IDE backward has consumed them
            const float* hd[MAXL] = {};
            predictor_grads(h->outer_light, g + L_OUTER, h->ld_outer);
            float* dxo = A.f32((size_t)rpad(n2) * h->ld_outer);
            float* dxi = A.f32((size_t)rpi * 128);
            float* dxh = h->cfg.human_light ? A.f32((size_t)rpi * 24) : nullptr;
            size_t mk = A.mark();
            hd[3] = dLh;
            Bwd ob;
            RC(h->outer_light.backward(A, M, h->f_out, n2, nullptr, 0, hd, true, false, nullptr, dxo, h->ld_outer, false, false, ob, stream));
            RC(h->outer_light.weight_grads(A, M, h->f_out, ob, n2, h->Xo2, h->ld_outer, nullptr, 0, hd, nullptr, nullptr, partials, (void*)fork_dw(h, A, hs)));
            if (!three) A.release(mk);
            predictor_grads(h->inner_light, g + L_INNER, 123);
            hd[3] = dLi;
            Bwd ib;
            RC(h->inner_light.backward(A, M, h->f_in, n_in, nullptr, 0, hd, true, false, nullptr, dxi, 128, false, false, ib, stream));
            RC(h->inner_light.weight_grads(A, M, h->f_in, ib, n_in, h->Xi, 128, nullptr, 0, hd, nullptr, nullptr, partials, (void*)fork_dw(h, A, hs)));
            if (!three) A.release(mk);
            predictor_grads(h->inner_weight, g + L_WEIGHT, 90);
            hd[3] = dLo;
            Bwd wb;
            RC(h->inner_weight.backward(A, M, h->f_w, n_in, nullptr, 0, hd, false, false, nullptr, nullptr, 0, false, false, wb, stream));
            RC(h->inner_weight.weight_grads(A, M, h->f_w, wb, n_in, h->Xo, 96, nullptr, 0, hd, nullptr, nullptr, partials, (void*)fork_dw(h, A, hs)));
            if (!three) A.release(mk);
            if (h->cfg.human_light) {
                predictor_grads(h->human_light, g + L_HUMAN, 24);
                hd[3] = dLhum;
                Bwd hb;
                RC(h->human_light.backward(A, M, h->f_h, n_in, nullptr, 0, hd, true, false, nullptr, dxh, 24, false, false, hb, stream));
                RC(h->human_light.weight_grads(A, M, h->f_h, hb, n_in, h->Xh, 24, nullptr, 0, hd, nullptr, nullptr, partials, (void*)fork_dw(h, A, hs)));
                if (!three) A.release(mk);
                LAUNCH(nero_human_encode_bwd(S.x4, S.geo, h->mat8, S.inner_idx, T, h->poses, n_in, dxh, extra, stream));
            }
            dX_outer = dxo; ld_dxo = h->ld_outer; dX_inner = dxi;
        }
        LAUNCH(nero_shade_encode_bwd(S.geo, h->mat8, dX_outer, dX_outer + (size_t)rpi * ld_dxo, dX_inner, dmat, n_in, d_geo, dmr, drr, dar, extra,
                                     S.x4, h->cfg.sphere_direction, stream));
        const float* dhs[3] = {dmr, drr, dar};
        const int lidx[3] = {L_METALLIC, L_ROUGHNESS, L_ALBEDO};
        for (int j = 0; j < 3; ++j) {
            const size_t mk = A.mark();
            predictor_grads(h->mat[j], g + lidx[j], 259);
            const float* hd[MAXL] = {};
            hd[3] = dhs[j];
            Bwd mb;
            RC(h->mat[j].backward(A, M, h->f_mat[j], n_in, nullptr, 0, hd, true, false, nullptr, d_feat, NERO_HID, j > 0, false, mb, stream));
            RC(h->mat[j].weight_grads(A, M, h->f_mat[j], mb, n_in, S.feat, NERO_HID, h->x8, 8, hd, nullptr, nullptr, partials, (void*)fork_dw(h, A, hs)));
            if (!three) A.release(mk);
        }
        // Where the weight-gradient stream is joined: behind the SDF network's own jobs (default), i.e. the tangent / second-order reverse
        // passes run beside the material MLPs' jobs.  HISTORY (round 5): the first version of this did not reproduce its SDF gradients bit for
        // bit (2 repeats of 9 differed by 1e-7 ... 1e-5).  scripts/r05/dbg_streams.py narrowed it to ONE kernel: sdf_alpha_bwd, with identical
        // inputs, returned 16-row blocks of d_grad a few ulp off whenever the narrow weight-gradient kernel shared its SIMDs -- packed fp32
        // VALU instructions next to another wave's MFMAs (common.h).  With the library built without them every join position is
        // bit-reproducible (8 x 9 repeats each).  NERO_DW_JOIN = e | j | t | b | l moves the join for experiments (early: in front of
        // sdf_alpha_bwd; j: in front of the tangent chain; t / b: behind the tangent / reverse launch; l: late, the default).
        static const char join_at = [] { const char* e = getenv("NERO_DW_JOIN"); return e ? e[0] : 'l'; }();
        const bool late_join = join_at == 'l';
        if (join_at == 'e') join_dw(h, A, hs);
        float* d_sdf4 = A.f32((size_t)rpi * 4);
        float* d_grad = A.f32((size_t)rpi * 3);
        float* dinv = A.f32(rpi);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
        LAUNCH(nero_sdf_alpha_bwd(S.sdf4, S.normal, S.x4, S.inner_idx, h->d, T, h->variance, h->anneal, n_in, d_ai, d_gerr, d_geo, d_sdf4, d_grad,
                                  dinv, stream));
        // ---- SDFField.backward: tangent chain, reverse chain with the sigma'' injections, weight gradients with two operand pairs ----
        Chain& sc = h->sdf_full;
        for (int l = 0; l < 8; ++l) set_dense_grad(sc.e[l], g[L_SDF + l], l == 0 ? D_PE : 256);
        {   // lin8: rows 1..256 = the dense part, row 0 = the sdf head, written in place into the [257, 256] gradient
            nero_linear_grad rows1 = {g[L_SDF + 8].dW ? g[L_SDF + 8].dW + 256 : nullptr, g[L_SDF + 8].db ? g[L_SDF + 8].db + 1 : nullptr};
            set_dense_grad(sc.e[8], rows1, 256);
            set_head_grad(sc.e[8], g[L_SDF + 8], 256);
        }
        float* ehat = A.f32((size_t)rpi * LD_PE);
        float* tbuf = A.f32((size_t)8 * rpi * NERO_HID);               // adot_0..7 (the injections are formed inside the reverse kernel)
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
        if (!A.dry) {
            const void* ps[9] = {d_geo, d_feat, d_sdf4, d_grad, dinv, ehat, tbuf, d_ai, dmr};
            const size_t bs[9] = {(size_t)rpi * 32, (size_t)rpi * NERO_HID * 4, (size_t)rpi * 16, (size_t)n_in * 12, (size_t)n_in * 4, (size_t)rpi * LD_PE * 4,
                                  (size_t)8 * rpi * NERO_HID * 4, (size_t)n_in * 4, (size_t)rpi * 16};
            for (int i = 0; i < 9; ++i) { h->dbg_ptr[i] = ps[i]; h->dbg_bytes[i] = bs[i]; }
        }
        LAUNCH(nero_pe_jvp(S.x4, 4, d_grad, 3, N_FREQ, n_in, ehat, LD_PE, stream));
        if (join_at == 'j') join_dw(h, A, hs);                 // (behind sdf_alpha_bwd + pe_jvp, in front of the tangent chain)
        nero_tan_chain tc;
        memset(&tc, 0, sizeof(tc));
        tc.init = ehat; tc.ld_init = LD_PE; tc.k_init = LD_PE; tc.aux = ehat; tc.ld_aux = LD_PE; tc.k_aux = LD_PE;
        tc.n_layers = 8; tc.aux_wide = 0; tc.gemm_mode = M.tan;
        const float* injs[MAXL] = {};
        const float* adots[MAXL] = {};
        Second second[MAXL];
        const float* head_extra[MAXL] = {};
        double macs = 0.0;
        for (int l = 0; l < 8; ++l) {
            const Dense& dd = sc.e[l].d;
            macs += (double)dd.n_out * (dd.k_main + dd.k_aux);
            nero_tan_layer& tl = tc.layer[l];
            tl.w_main = sc.e[l].hfm; tl.w_aux = sc.e[l].hfa;
            tl.a_saved = h->f_sdf.saves[l]; tl.gbar = nullptr;
            tl.adot = tbuf + (size_t)l * rpi * NERO_HID;
            tl.inj = nullptr;
            tl.k_main = r16(dd.k_main); tl.k_aux = dd.k_aux ? r16(dd.k_aux) : 0; tl.n_tiles = tiles(dd.n_out);
            injs[l] = h->b_normal.deltas[l];                   // (gbar_l: the reverse kernel forms gbar beta (1 - s) adot / s itself)
            adots[l] = tl.adot;
            second[l].D1 = h->b_normal.deltas[l]; second[l].ldd1 = NERO_HID;
            second[l].B1m = l == 0 ? ehat : tbuf + (size_t)(l - 1) * rpi * NERO_HID; second[l].ldb1m = l == 0 ? LD_PE : NERO_HID;
            second[l].B1a = ehat; second[l].ldb1a = LD_PE;
        }
        tc.macs_per_row = macs;
        head_extra[8] = tbuf + (size_t)7 * rpi * NERO_HID;
        LAUNCH(nero_mlp_tangent(&tc, n_in, stream));
        if (join_at == 't') join_dw(h, A, hs);
        const float* sd[MAXL] = {};
        sd[8] = d_sdf4;
        Bwd sb;
        RC(sc.backward(A, M, h->f_sdf, n_in, d_feat, NERO_HID, sd, false, false, injs, nullptr, 0, false, false, sb, stream, adots));
        if (join_at == 'b') join_dw(h, A, hs);
        RC(sc.weight_grads(A, M, h->f_sdf, sb, n_in, h->pe40, LD_PE, h->pe40, LD_PE, sd, second, head_extra, partials,
                           late_join ? (void*)fork_dw(h, A, hs) : stream));
        if (late_join) join_dw(h, A, hs);
        float* part = A.f32(128);
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: workspace too small");
        if (d_inv_s_sum && !A.dry) {
            hipLaunchKernelGGL(sum_kernel, dim3(128), dim3(256), 0, hs, dinv, n_in, part);
            hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, hs, part, 128, d_inv_s_sum);
        }
    } else if (d_inv_s_sum && !A.dry) {
        (void)hipMemsetAsync(d_inv_s_sum, 0, 4, hs);
    }
    if (!A.dry) join_side(h, so_, hs);                 // the caller's stream continues behind BOTH branches
    return nero_check_launch("nero_stage1_render_bwd");
}

}  // namespace

extern "C" {

int nero_stage1_create(const nero_stage1_cfg* cfg, nero_stage1** out) {
    if (!cfg || !out) return nero_fail(NERO_ERR_ARG, "nero_stage1_create: bad argument");
    if (cfg->n_importance % (cfg->up_sample_steps > 0 ? cfg->up_sample_steps : 1) || cfg->up_sample_steps < 1 || cfg->n_samples < 2)
        return nero_fail(NERO_ERR_ARG, "nero_stage1_create: n_importance must be a multiple of up_sample_steps");
    if (!is_f16(cfg->gemm_fwd) || cfg->gemm_tan != NERO_GEMM_F16X3 || cfg->gemm_bwd != NERO_GEMM_F16X3 || !is_f16(cfg->gemm_dw))
        return nero_fail(NERO_ERR_UNSUPPORTED, "nero_stage1_create: the C-level driver packs fp16 two-plane operands only (F16X3)");
    nero_stage1* h = new (std::nothrow) nero_stage1();
    if (!h) return nero_fail(NERO_ERR_ARG, "nero_stage1_create: out of host memory");
    h->cfg = *cfg;
    h->M = {cfg->gemm_fwd, cfg->gemm_tan, cfg->gemm_bwd, cfg->gemm_dw};      // (validated above: all four are NERO_GEMM_F16X3)
    memset(&h->st, 0, sizeof(h->st));
    nero_stage1_weights zero;
    memset(&zero, 0, sizeof(zero));
    build_chains(h, &zero);                      // shapes only: the size queries work before the first pack
    make_value_chain(h);
    const char* e = getenv("NERO_STREAMS");
    h->n_streams = e ? atoi(e) : NERO_STREAMS_DEFAULT;
    if (hipGetDevice(&h->device) != hipSuccess) { h->device = -1; (void)hipGetLastError(); }
    if (h->n_streams >= 2) {
        if (hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();             // (no device in reach -- the CPU-side size queries still work -- or out of handles: one stream)
            h->s2 = nullptr;
        }
    }
    const bool have_device = h->s2 != nullptr || h->n_streams < 2;
    if (h->n_streams >= 3) {
        bool ok = hipStreamCreateWithFlags(&h->s3, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&h->ev_dw_done, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < nero_stage1::N_DW_EV; ++i) ok = hipEventCreateWithFlags(&h->ev_dw[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            h->s3 = nullptr;
            // out of stream / event handles on a live device: the memory plan must follow the streams that exist -- in three-stream mode no
            // chain delta is released before the step ends (+6 GiB at 4096 rays), which only pays when the third stream runs.
            // Without a device (CPU-side size queries) the handle keeps the requested mode, so that the sizes match the GPU box's.
            if (have_device) h->n_streams = 2;
        }
    }
    *out = h;
    return NERO_OK;
}

void nero_stage1_destroy(nero_stage1* h) {
    if (!h) return;
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->s2) (void)hipStreamDestroy(h->s2);
    for (int i = 0; i < nero_stage1::N_DW_EV; ++i)
        if (h->ev_dw[i]) (void)hipEventDestroy(h->ev_dw[i]);
    if (h->ev_dw_done) (void)hipEventDestroy(h->ev_dw_done);
    if (h->s3) (void)hipStreamDestroy(h->s3);
    delete h;
}

size_t nero_stage1_pack_bytes(nero_stage1* h) { return h ? pack_floats_total(h) * 4 : 0; }

int nero_stage1_pack(nero_stage1* h, const nero_stage1_weights* w, void* pack_buf, void* stream) {
    if (!h || !w || !pack_buf) return nero_fail(NERO_ERR_ARG, "nero_stage1_pack: bad argument");
    build_chains(h, w);
    const size_t total = pack_floats_total(h);
    (void)hipMemsetAsync(pack_buf, 0, total * 4, (hipStream_t)stream);
    std::vector<nero_pack_job> jobs;
    float* p = static_cast<float*>(pack_buf);
    for (Chain* c : all_chains(h)) {
        float* q = p;
        c->pack(q, jobs);
        p += (c->pack_floats() + 63) / 64 * 64;
    }
    make_value_chain(h);
    for (size_t i0 = 0; i0 < jobs.size(); i0 += NERO_MAX_PACK_JOBS) {
        const int n = (int)(jobs.size() - i0 < NERO_MAX_PACK_JOBS ? jobs.size() - i0 : NERO_MAX_PACK_JOBS);
        RC(nero_pack_batch(jobs.data() + i0, n, stream));
    }
    h->packed = true;
    return NERO_OK;
}

static size_t workspace_bytes_impl(nero_stage1* h, int R, int n_in, int n_out, int with_sampler, bool with_backward);

size_t nero_stage1_workspace_bytes_for(nero_stage1* h, int R, int n_in, int n_out, int with_sampler) {
    return workspace_bytes_impl(h, R, n_in, n_out, with_sampler, true);
}

static size_t workspace_bytes_impl(nero_stage1* h, int R, int n_in, int n_out, int with_sampler, bool with_backward) {
    if (!h || R < 0) return 0;
    const nero_stage1_cfg& c = h->cfg;
    const int T = c.n_samples + c.n_importance + c.n_bg_samples;
    nero_stage1 tmp = *h;                         // dry run on a copy: same carve logic, no memory, no launches
    Arena& A = tmp.A;
    A = Arena();
    A.dry = true;
    size_t peak = 0;
    if (with_sampler) {
        float* z = A.f32((size_t)R * T);
        (void)do_sample(&tmp, A, R, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, z, nullptr);
        peak = A.peak;
        A.off = 0;
    }
    nero_stage1_state& S = tmp.st;
    S.R = R; S.T = T; S.n_in = n_in; S.n_out = n_out;
    S.pts4 = A.f32((size_t)R * T * 4); S.ray_counts = A.i32(R); S.ray_off = A.i32(R); S.counts = A.i32(2);
    (void)do_forward(&tmp, A, R, T, n_in, n_out, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr);
    nero_stage1_grads G;
    memset(&G, 0, sizeof(G));
    for (int i = 0; i < NERO_S1_LINEARS; ++i) { G.lin[i].dW = reinterpret_cast<float*>(0x1000); G.lin[i].db = reinterpret_cast<float*>(0x1000); }
    if (with_backward) (void)do_backward(&tmp, A, nullptr, nullptr, nullptr, &G, nullptr, nullptr);
    peak = A.peak > peak ? A.peak : peak;
    return peak + 4096;
}

static size_t workspace_worst(nero_stage1* h, int R, bool with_backward) {
    if (!h) return 0;
    const nero_stage1_cfg& c = h->cfg;
    const int T = c.n_samples + c.n_importance + c.n_bg_samples;
    // worst cases of the data-dependent split: every sample inner (the expensive kind) / every sample outer, and one sample short of
    // either end (both partitions padded to whole 64-row tiles: the carve is linear in the PADDED row counts)
    const int N = R * T;
    const int inner[4] = {N, 0, N > 1 ? N - 1 : N, N > 1 ? 1 : 0};
    size_t worst = 0;
    for (int k = 0; k < 4; ++k) {
        const size_t b = workspace_bytes_impl(h, R, inner[k], N - inner[k], 1, with_backward);
        worst = b > worst ? b : worst;
    }
    return worst + 65536;
}

size_t nero_stage1_workspace_bytes(nero_stage1* h, int R) { return workspace_worst(h, R, true); }

// sampler + render forward only (inference: nvs / test_step never call nero_stage1_render_bwd on this workspace)
size_t nero_stage1_workspace_bytes_fwd(nero_stage1* h, int R) { return workspace_worst(h, R, false); }

int nero_stage1_sample(nero_stage1* h, int R, const float* o, const float* d, const float* near, const float* far, const float* variance,
                       const float* rand1, const float* rand_bg, float* z_vals, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !h->packed || !o || !d || !near || !far || !variance || !z_vals || !ws)
        return nero_fail(NERO_ERR_ARG, "nero_stage1_sample: bad argument (pack the weights first)");
    if (R == 0) return NERO_OK;
    Arena A;
    A.base = static_cast<char*>(ws); A.cap = ws_bytes;
    RC(do_sample(h, A, R, o, d, near, far, variance, rand1, rand_bg, z_vals, stream));
    return nero_check_launch("nero_stage1_sample");
}

int nero_stage1_render_fwd(nero_stage1* h, int R, const float* o, const float* d, const float* z_vals, const float* variance,
                           const float* lut, const float* poses, float anneal, float* rgb, float* gerr, float* occ_prob,
                           int* n_in_out, int* n_out_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !h->packed || !o || !d || !z_vals || !variance || !lut || !rgb || !gerr || !occ_prob || !ws || R <= 0)
        return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: bad argument (pack the weights first)");
    if (h->cfg.human_light && !poses) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: human_light needs poses [R,3,4]");
    if (wrong_device(h)) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: the current device is not the one the handle was created on");
    const nero_stage1_cfg& c = h->cfg;
    const int T = c.n_samples + c.n_importance + c.n_bg_samples;
    Arena& A = h->A;
    A = Arena();
    A.base = static_cast<char*>(ws); A.cap = ws_bytes;
    nero_stage1_state& S = h->st;
    memset(&S, 0, sizeof(S));
    S.R = R; S.T = T;
    S.pts4 = A.f32((size_t)R * T * 4); S.ray_counts = A.i32(R); S.ray_off = A.i32(R); S.counts = A.i32(2);
    if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_fwd: workspace too small");
    RC(nero_render_prep(o, d, z_vals, R, T, S.pts4, S.ray_counts, S.ray_off, S.counts, stream));
    int counts[2] = {0, 0};
    if (hipMemcpyAsync(counts, S.counts, 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)                      // the step's one host synchronisation
        return nero_fail(NERO_ERR_LAUNCH, "nero_stage1_render_fwd: reading the sample counts failed");
    S.n_in = counts[0]; S.n_out = counts[1];
    if (n_in_out) *n_in_out = S.n_in;
    if (n_out_out) *n_out_out = S.n_out;
    h->o = o; h->d = d; h->variance = variance; h->lut = lut; h->poses = poses; h->anneal = anneal;
    if (S.n_in == 0) {
        (void)hipMemsetAsync(gerr, 0, 4, (hipStream_t)stream);
        (void)hipMemsetAsync(occ_prob, 0, 4, (hipStream_t)stream);
    }
    RC(drain_on_error(h, do_forward(h, A, R, T, S.n_in, S.n_out, d, variance, lut, poses, anneal, rgb, gerr, occ_prob, stream)));
    h->step_mark = A.mark();
    return nero_check_launch("nero_stage1_render_fwd");
}

int nero_stage1_render_bwd(nero_stage1* h, const float* d_rgb, const float* d_gerr, const float* d_occ, const nero_stage1_grads* grads,
                           float* d_inv_s_sum, void* stream) {
    if (!h || !h->A.base || !d_rgb || !grads) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: bad argument (no forward state)");
    if (wrong_device(h)) return nero_fail(NERO_ERR_ARG, "nero_stage1_render_bwd: the current device is not the one the handle was created on");
    h->A.release(h->step_mark);
    return drain_on_error(h, do_backward(h, h->A, d_rgb, d_gerr, d_occ, grads, d_inv_s_sum, stream));
}

// debug (scripts/r05/dbg_streams.py): device pointers + sizes of nine intermediates of the last render_bwd, in the order
// d_geo, d_feat, d_sdf4, d_grad, dinv, ehat, adot (8 layers), d_alpha_inner, d_metallic_raw
int nero_stage1_debug_buffers(nero_stage1* h, const void** ptrs, size_t* bytes) {
    if (!h || !ptrs || !bytes) return nero_fail(NERO_ERR_ARG, "nero_stage1_debug_buffers: bad argument");
    for (int i = 0; i < 9; ++i) { ptrs[i] = h->dbg_ptr[i]; bytes[i] = h->dbg_bytes[i]; }
    return NERO_OK;
}

int nero_stage1_get_state(nero_stage1* h, nero_stage1_state* out) {
    if (!h || !out) return nero_fail(NERO_ERR_ARG, "nero_stage1_get_state: bad argument");
    *out = h->st;
    return NERO_OK;
}

int nero_stage1_sdf_from_pe(nero_stage1* h, const float* pe, int n, float* out4, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !h->packed || !pe || !out4 || !ws) return nero_fail(NERO_ERR_ARG, "nero_stage1_sdf_from_pe: bad argument");
    if (n == 0) return NERO_OK;
    Arena A;
    A.base = static_cast<char*>(ws); A.cap = ws_bytes;
    float* res = nullptr;
    RC(sdf_from_pe(h, A, pe, n, res, stream));
    (void)hipMemcpyAsync(out4, res, (size_t)rpad(n) * 16, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    return nero_check_launch("nero_stage1_sdf_from_pe");
}

}  // extern "C"
