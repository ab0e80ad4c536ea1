// chain_host.h -- host-side machinery shared by the C-level step drivers (stage1_driver.hip, stage2_driver.hip): the workspace arena
// and the C++ twin of nero_amd/chain.py::Chain (one network as a list of dense / head entries: operand packing, forward, reverse and
// weight-gradient launches through the library's own entry points).  Include once per translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

constexpr int MAXL = NERO_MAX_LAYERS;
inline int r8(int x) { return (x + 7) / 8 * 8; }
inline int r16(int x) { return (x + 15) / 16 * 16; }
inline int tiles(int x) { return (x + 31) / 32; }
inline int rpad(int n) { return NERO_ROW_PAD(n); }
// contraction length of a layer's reverse GEMM: wide layers are zero-padded to 256 (nero_amd/chain.py::_rev_k)
inline int rev_k(int n_out) { return (n_out > 128 && n_out <= 256) ? 256 : r16(n_out); }
inline bool is_f16(int m) { return m == NERO_GEMM_F16X3; }

// ---- workspace arena -------------------------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;                                  // size query: no memory behind it, nothing is launched
    bool failed = false;
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        peak = off > peak ? off : peak;
        if (dry) return reinterpret_cast<void*>(0x1000 + a);       // a non-NULL token: descriptors are built, never dereferenced
        if (off > cap) { failed = true; return nullptr; }
        return base + a;
    }
    float* f32(size_t n) { return static_cast<float*>(take(n * 4)); }
    int* i32(size_t n) { return static_cast<int*>(take(n * 4)); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

__global__ void x8_from_x4_kernel(const float* __restrict__ x4, float* __restrict__ x8, int n, int n_pad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) { v = reinterpret_cast<const float4*>(x4)[r]; v.w = 0.f; }      // (rows_pad rows, as x8[:, :3] = x4[:, :3] over the padded buffer)
    else { v = reinterpret_cast<const float4*>(x4)[r]; v.w = 0.f; }
    reinterpret_cast<float4*>(x8)[2 * r] = v;
    reinterpret_cast<float4*>(x8)[2 * r + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void ones_col0_kernel(float* __restrict__ b, int n_pad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_pad) reinterpret_cast<float4*>(b)[r] = make_float4(1.f, 0.f, 0.f, 0.f);
}
// dst[r*ldd + c] = src[r*lds + c]
__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx - r * cols;
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}
// out[b] = sum of block b's grid-stride share of v[0..n) (deterministic two-stage reduction: 128 partials, then one block over them)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

#define RC(expr) do { const int rc_ = (expr); if (rc_ != NERO_OK) return rc_; } while (0)
#define LAUNCH(...) do { if (!A.dry) { RC(__VA_ARGS__); } } while (0)

// ---- one network as a list of entries (nero_amd/chain.py::Chain) ------------------------------------------------------------------
struct Dense {
    bool has = false;
    const float* W = nullptr; int ldw = 0; const float* b = nullptr;
    int n_out = 0, act = 0, k_main = 0, main_c0 = 0, k_aux = 0, aux_c0 = 0;
    float scale = 1.f;
    float* dW = nullptr; int ld_dw = 0; float* db = nullptr;          // gradient destinations (row-major like W), set per backward
};
struct Head {
    bool has = false;
    const float* W = nullptr; int ldw = 0; const float* b = nullptr;
    int n_head = 0, k = 0;
    float* dW = nullptr; int ld_dw = 0; float* db = nullptr;
};
struct Entry {
    Dense d; Head h;
    float *bias = nullptr, *hfm = nullptr, *hfa = nullptr, *hbm = nullptr, *hba = nullptr, *hw = nullptr, *hb = nullptr;   // packed images
};
struct Fwd {
    float* saves[MAXL]; uint32_t* masks[MAXL]; float* heads[MAXL];
    Fwd() { for (int i = 0; i < MAXL; ++i) { saves[i] = nullptr; masks[i] = nullptr; heads[i] = nullptr; } }
};
struct Bwd {
    const float* deltas[MAXL]; int ld_delta[MAXL];
    float* d_init = nullptr; int ld_dinit = 0; float* d_aux = nullptr; int ld_daux = 0;
    Bwd() { for (int i = 0; i < MAXL; ++i) { deltas[i] = nullptr; ld_delta[i] = NERO_HID; } }
};
struct Second { const float* D1 = nullptr; int ldd1 = 0; const float* B1m = nullptr; int ldb1m = 0; const float* B1a = nullptr; int ldb1a = 0; };

struct Modes { int fwd, tan, bwd, dw; };

struct Chain {
    std::vector<Entry> e;
    int k_init = 0, k_aux = 0, aux_wide = 0;
    int n() const { return (int)e.size(); }
    int last_dense() const { int l = -1; for (int i = 0; i < n(); ++i) if (e[i].d.has) l = i; return l; }

    // floats of the packed operand images (fp16 two-plane engine: 64-float header + 512 floats per (tile, 16-k step))
    size_t pack_floats() const {
        size_t t = 0;
        for (const Entry& x : e) {
            if (x.d.has) {
                const Dense& d = x.d;
                const int nt = tiles(d.n_out);
                t += 32 * nt;
                if (d.k_main) t += 64 + (size_t)(r16(d.k_main) / 16) * nt * 512 + 64 + (size_t)(rev_k(d.n_out) / 16) * tiles(d.k_main) * 512;
                if (d.k_aux) t += 64 + (size_t)(r16(d.k_aux) / 16) * nt * 512 + 64 + (size_t)(rev_k(d.n_out) / 16) * tiles(d.k_aux) * 512;
            }
            if (x.h.has) t += 4 * NERO_HID + 4;
        }
        return t;
    }
    // carve the images from `buf` (zero-filled by the caller) and append the pack jobs
    void pack(float*& buf, std::vector<nero_pack_job>& jobs) {
        auto job = [&](int kind, const float* W, float* out, int nrows, int ld, int col0, int ncols, int transpose, int kpad, int nt_count, float scale) {
            nero_pack_job j;
            j.W = W; j.out = out; j.kind = kind; j.nrows = nrows; j.ld = ld; j.col0 = col0; j.ncols = ncols; j.transpose = transpose;
            j.kpad = kpad; j.nt_count = nt_count; j.scale = scale; j.pad_ = 0;
            jobs.push_back(j);
        };
        for (Entry& x : e) {
            if (x.d.has) {
                const Dense& d = x.d;
                const int nt = tiles(d.n_out);
                x.bias = buf; buf += 32 * nt;
                x.hfm = x.hfa = x.hbm = x.hba = nullptr;
                if (d.k_main) {
                    x.hfm = buf; buf += 64 + (size_t)(r16(d.k_main) / 16) * nt * 512;
                    x.hbm = buf; buf += 64 + (size_t)(rev_k(d.n_out) / 16) * tiles(d.k_main) * 512;
                    job(3, d.W, x.hfm, d.n_out, d.ldw, d.main_c0, d.k_main, 0, r16(d.k_main), nt, d.scale);
                    job(3, d.W, x.hbm, d.n_out, d.ldw, d.main_c0, d.k_main, 1, rev_k(d.n_out), tiles(d.k_main), d.scale);
                }
                if (d.k_aux) {
                    x.hfa = buf; buf += 64 + (size_t)(r16(d.k_aux) / 16) * nt * 512;
                    x.hba = buf; buf += 64 + (size_t)(rev_k(d.n_out) / 16) * tiles(d.k_aux) * 512;
                    job(3, d.W, x.hfa, d.n_out, d.ldw, d.aux_c0, d.k_aux, 0, r16(d.k_aux), nt, d.scale);
                    job(3, d.W, x.hba, d.n_out, d.ldw, d.aux_c0, d.k_aux, 1, rev_k(d.n_out), tiles(d.k_aux), d.scale);
                }
                if (d.b) job(2, d.b, x.bias, 1, d.n_out, 0, d.n_out, 0, 32 * nt, 0, 1.f);
            }
            if (x.h.has) {
                x.hw = buf; buf += 4 * NERO_HID;
                x.hb = buf; buf += 4;
                job(2, x.h.W, x.hw, x.h.n_head, x.h.ldw, 0, x.h.k, 0, NERO_HID, 0, 1.f);
                if (x.h.b) job(2, x.h.b, x.hb, 1, x.h.n_head, 0, x.h.n_head, 0, 4, 0, 1.f);
            }
        }
    }

    // nero_amd/chain.py::Chain.forward
    int forward(Arena& A, const Modes& M, const float* init, int ld_init, const float* aux, int ld_aux, int n_rows, bool save, Fwd& F,
                void* stream) const {
        const int rp = rpad(n_rows);
        nero_fwd_chain ch;
        memset(&ch, 0, sizeof(ch));
        ch.init = init; ch.ld_init = init ? ld_init : 0; ch.k_init = k_init;
        ch.aux = aux; ch.ld_aux = aux ? ld_aux : 0; ch.k_aux = k_aux;
        ch.n_layers = n(); ch.aux_wide = aux_wide;
        ch.gemm_mode = M.fwd;
        double macs = 0.0;
        for (const Entry& x : e) if (x.d.has) macs += (double)x.d.n_out * (x.d.k_main + x.d.k_aux);
        ch.macs_per_row = macs;
        const int ld = last_dense();
        for (int i = 0; i < n(); ++i) {
            const Entry& x = e[i];
            nero_fwd_layer& fl = ch.layer[i];
            if (x.h.has) {
                F.heads[i] = A.f32((size_t)rp * 4);
                fl.head_w = x.hw; fl.head_b = x.hb; fl.head_out = F.heads[i];
                fl.n_head = x.h.n_head; fl.head_k = (x.h.k + 3) / 4 * 4;
            }
            if (x.d.has) {
                const Dense& d = x.d;
                fl.w_main = x.hfm; fl.w_aux = x.hfa; fl.bias = x.bias;
                fl.k_main = d.k_main ? r16(d.k_main) : 0; fl.k_aux = d.k_aux ? r16(d.k_aux) : 0;
                fl.n_tiles = tiles(d.n_out); fl.act = d.act;
                if (save || i == ld) { F.saves[i] = A.f32((size_t)rp * NERO_HID); fl.save = F.saves[i]; }
                if (save && d.act == NERO_ACT_RELU) { F.masks[i] = reinterpret_cast<uint32_t*>(A.i32((size_t)rp * 8)); fl.relu_mask = F.masks[i]; }
            }
        }
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1: workspace too small");
        LAUNCH(nero_mlp_forward(&ch, n_rows, stream));
        return NERO_OK;
    }

    // nero_amd/chain.py::Chain.backward
    int backward(Arena& A, const Modes& M, const Fwd& F, int n_rows, const float* dy, int ld_dy, const float* const* head_dys /*[MAXL] or NULL*/,
                 bool need_dinit, bool need_daux, const float* const* injs /*[MAXL] or NULL*/, float* dinit_out, int ld_dinit_out,
                 bool accumulate_dinit, bool skip_last_dense, Bwd& B, void* stream,
                 const float* const* inj_adots = nullptr /*[MAXL]: injs then holds gbar, nero_bwd_layer.inj_adot*/) const {
        const int rp = rpad(n_rows), last = n() - 1;
        nero_bwd_chain ch;
        memset(&ch, 0, sizeof(ch));
        ch.n_layers = n(); ch.aux_wide = 0; ch.gemm_mode = M.bwd;
        if (dy) {
            ch.dy = dy; ch.ld_dy = ld_dy;
            ch.k_dy = e[last].d.has ? r8(e[last].d.n_out) : NERO_HID;
        }
        if (need_dinit) {
            if (dinit_out) { B.d_init = dinit_out; B.ld_dinit = ld_dinit_out; }
            else { B.d_init = A.f32((size_t)rp * k_init); B.ld_dinit = k_init; }
            ch.d_init = B.d_init; ch.ld_dinit = B.ld_dinit; ch.accumulate_dinit = accumulate_dinit ? 1 : 0;
        }
        if (need_daux) {
            B.d_aux = A.f32((size_t)rp * k_aux); B.ld_daux = k_aux;
            if (!A.dry && !A.failed) (void)hipMemsetAsync(B.d_aux, 0, (size_t)rp * k_aux * 4, (hipStream_t)stream);
            ch.d_aux = B.d_aux; ch.ld_daux = k_aux;
        }
        int prev_dense[MAXL];
        int pd = -1;
        for (int i = 0; i < n(); ++i) { prev_dense[i] = pd; if (e[i].d.has) pd = i; }
        float* dw[MAXL];
        for (int i = 0; i < n(); ++i) {
            dw[i] = nullptr;
            if (e[i].d.has) { dw[i] = A.f32((size_t)rp * NERO_HID); B.deltas[i] = dw[i]; B.ld_delta[i] = NERO_HID; }
        }
        double macs = 0.0;
        for (int i = 0; i < n(); ++i) {
            const Entry& x = e[i];
            nero_bwd_layer& bl = ch.layer[i];
            const int j = prev_dense[i];
            if (x.d.has && !(skip_last_dense && i == last)) {
                const Dense& d = x.d;
                bl.w_main_t = x.hbm;
                bl.w_aux_t = need_daux ? x.hba : nullptr;
                bl.n_out = rev_k(d.n_out);
                bl.k_main_tiles = d.k_main ? tiles(d.k_main) : 0;
                bl.k_aux_tiles = d.k_aux ? tiles(d.k_aux) : 0;
                const bool first = j < 0;
                if (first && !need_dinit) macs += (need_daux && d.k_aux) ? (double)d.n_out * d.k_aux : 0.0;
                else macs += (double)d.n_out * (d.k_main + (need_daux ? d.k_aux : 0));
            } else {
                bl.n_out = 0;
                bl.k_main_tiles = j >= 0 ? tiles(e[j].d.n_out) : 0;
            }
            if (x.h.has && head_dys && head_dys[i]) { bl.head_w = x.hw; bl.head_dy = head_dys[i]; bl.n_head = x.h.n_head; }
            if (j >= 0) {
                bl.a_prev = F.saves[j];
                bl.act_prev = e[j].d.act;
                if (F.masks[j]) bl.mask_prev = F.masks[j];
                bl.delta_prev = dw[j];
                if (injs && injs[j]) { bl.inj = injs[j]; if (inj_adots && inj_adots[j]) bl.inj_adot = inj_adots[j]; }
            }
        }
        if (e[last].d.has && !skip_last_dense) { B.deltas[last] = dy; B.ld_delta[last] = ld_dy; }   // (the delta of the last dense entry is dy itself)
        ch.macs_per_row = macs;
        if (A.failed) return nero_fail(NERO_ERR_ARG, "nero_stage1: workspace too small");
        LAUNCH(nero_mlp_backward(&ch, n_rows, stream));
        return NERO_OK;
    }

    // nero_amd/chain.py::Chain.weight_grads -- every result goes straight to the entry's dW / db destination
    int weight_grads(Arena& A, const Modes& M, const Fwd& F, const Bwd& B, int n_rows, const float* init, int ld_init, const float* aux,
                     int ld_aux, const float* const* head_dys, const Second* second /*[MAXL] or NULL*/, const float* const* head_extra,
                     float* partials, void* stream) const {
        int prev = -1;
        std::vector<nero_dw_job> dw_jobs;           // every dense layer's job(s) of this chain: issued together at the end (nero_dw_gemm_batch)
        for (int i = 0; i < n(); ++i) {
            const Entry& x = e[i];
            if (x.h.has && head_dys && head_dys[i] && x.h.dW) {
                // (straight into the destination: n_head rows of k columns at pitch ld_dw; db may be NULL)
                LAUNCH(nero_head_dw_ld(head_dys[i], F.saves[prev], head_extra ? head_extra[i] : nullptr, x.h.n_head, n_rows, x.h.dW, x.h.ld_dw, x.h.k, x.h.db,
                                       partials, 0, stream));
            }
            if (x.d.has) {
                const Dense& d = x.d;
                if (d.dW) {
                    const float* main_in = prev < 0 ? init : F.saves[prev];
                    const int ld_main = prev < 0 ? ld_init : NERO_HID;
                    struct Part { const float* Bm; int ldb, kc, c0, which; } parts[2];
                    int np = 0;
                    if (d.k_main) parts[np++] = {main_in, ld_main, d.k_main, d.main_c0, 0};
                    if (d.k_aux) parts[np++] = {aux, ld_aux, d.k_aux, d.aux_c0, 1};
                    for (int pi = 0; pi < np; ++pi) {
                        nero_dw_job job;
                        memset(&job, 0, sizeof(job));
                        job.d0 = B.deltas[i]; job.ldd0 = B.ld_delta[i]; job.b0 = parts[pi].Bm; job.ldb0 = parts[pi].ldb;
                        if (second && second[i].D1) {
                            job.d1 = second[i].D1; job.ldd1 = second[i].ldd1;
                            job.b1 = parts[pi].which ? second[i].B1a : second[i].B1m;
                            job.ldb1 = parts[pi].which ? second[i].ldb1a : second[i].ldb1m;
                        }
                        job.n_out = d.n_out; job.k_cols = parts[pi].kc;
                        job.dW = d.dW; job.ldw = d.ld_dw; job.col0 = parts[pi].c0;
                        job.db = pi == 0 ? d.db : nullptr;
                        job.scale = d.scale; job.accumulate = 0; job.gemm_mode = M.dw;
                        dw_jobs.push_back(job);
                    }
                }
                prev = i;
            }
        }
        if (!dw_jobs.empty()) LAUNCH(nero_dw_gemm_batch(dw_jobs.data(), (int)dw_jobs.size(), n_rows, partials, stream));
        return NERO_OK;
    }
};

Entry dense(const nero_linear& L, int ldw, int n_out, int act, int k_main, int main_c0 = 0, int k_aux = 0, int aux_c0 = 0, float scale = 1.f) {
    Entry x;
    x.d.has = true; x.d.W = L.W; x.d.ldw = ldw; x.d.b = L.b; x.d.n_out = n_out; x.d.act = act;
    x.d.k_main = k_main; x.d.main_c0 = main_c0; x.d.k_aux = k_aux; x.d.aux_c0 = aux_c0; x.d.scale = scale;
    return x;
}
Entry head_only(const nero_linear& L, int ldw, int n_head, int k) {
    Entry x;
    x.h.has = true; x.h.W = L.W; x.h.ldw = ldw; x.h.b = L.b; x.h.n_head = n_head; x.h.k = k;
    return x;
}
void set_dense_grad(Entry& x, const nero_linear_grad& g, int ld) { x.d.dW = g.dW; x.d.ld_dw = ld; x.d.db = g.db; }
void set_head_grad(Entry& x, const nero_linear_grad& g, int ld) { x.h.dW = g.dW; x.h.ld_dw = ld; x.h.db = g.db; }

// the four-layer predictors (make_predictor, network/field.py:310-346): layer 0 may take [main | aux] columns
Chain predictor(const nero_linear* L, int k_main0, int k_aux0, int k_init, int k_aux, int n_head) {
    Chain c;
    const int k0 = k_main0 + k_aux0;
    c.e.push_back(dense(L[0], k0, 256, NERO_ACT_RELU, k_main0, 0, k_aux0, k_main0));
    c.e.push_back(dense(L[1], 256, 256, NERO_ACT_RELU, 256));
    c.e.push_back(dense(L[2], 256, 256, NERO_ACT_RELU, 256));
    c.e.push_back(head_only(L[3], 256, n_head, 256));
    c.k_init = k_init; c.k_aux = k_aux;
    return c;
}
void predictor_grads(Chain& c, const nero_linear_grad* g, int k0) {
    set_dense_grad(c.e[0], g[0], k0);
    set_dense_grad(c.e[1], g[1], 256);
    set_dense_grad(c.e[2], g[2], 256);
    set_head_grad(c.e[3], g[3], 256);
}

}  // namespace
