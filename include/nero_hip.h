/* nero_hip.h -- C ABI of libnero_hip.so, the MI355X (gfx950) implementation of the NeRO Stage-I/II render-step hot path.
 *
 * The reference (liuyuan-pal/NeRO) is pure Python/PyTorch: it has no FFI for this path except two third-party CUDA
 * extensions (`_raytracing.create_raytracer/trace`, raytracing/raytracer.py:19,49, and `nvdiffrast.torch.texture`,
 * network/field.py:612).  The entry points below are therefore the boundary a maintainer binds with ctypes from the
 * reference's own modules (INTEGRATION.md shows the stubs); each one cites the reference code it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 (or int32 where stated) unless marked host; matrices are
 * row-major with an explicit leading dimension; every row count `n` may be any value >= 0 but all row buffers must be
 * allocated for NERO_ROW_PAD(n) rows; every call is asynchronous on `stream` (a hipStream_t passed as void*); every
 * function returns 0 on success or a negative error code, with a message available from nero_last_error().
 * Nothing here allocates device memory or synchronises, except nero_bvh_create / nero_bvh_destroy (the library owns BVH
 * handles) and the one-time upload of the IDE coefficient table.
 */
#ifndef NERO_HIP_H
#define NERO_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERO_TILE_ROWS 64
#define NERO_ROW_PAD(n) (((n) + 63) / 64 * 64)
#define NERO_MAX_LAYERS 10
#define NERO_HID 256              /* leading dimension of every saved hidden activation matrix */

enum { NERO_ACT_NONE = 0, NERO_ACT_RELU = 1, NERO_ACT_SOFTPLUS100 = 2 };
/* arithmetic of the dense layers.  F32: v_mfma_f32_32x32x2_f32, an exact fp32 fmaf chain (157 TFLOP/s peak).
 * BF16X6: every fp32 operand is carried as three bf16 planes (exact 3-way split) and each product is the sum of the six
 * significant plane products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- dropped terms <= 3*2^-27 relative, i.e.
 * below fp32's own rounding; 417 TFLOP/s fp32-equivalent peak.  Packed operands are mode specific. */
/* F16X3: operands as two fp16 planes h = fp16(xs), l = fp16(xs - h) of their block-scaled value xs (per activation row / per weight
 * matrix, exact powers of two that put the block maximum into [2^14, 2^15), the top of fp16's range), three plane products
 * hw hx + hw lx + lw hx accumulated into ONE fp32 accumulator set -- half the MFMA count of BF16X6 at the same error class
 * (representation error <= 2^-24 of the block maximum, dropped term <= 2^-24 of the product).  (Rounds 1-4 scaled into [0.5, 1),
 * stored l * 2^11 and kept two accumulator sets; -DF16_TWO_ACC rebuilds that format.) */
/* (value 3 was NERO_GEMM_F16X3P, the two-workgroups-per-CU forward engine of rounds 2-3, removed in round 4 for a wrong partial sum in one
 * launch of three at size.  Round 5 found the mechanism -- packed fp32 beside another wave's MFMAs, DESIGN.md 9.3 -- and the kernels are
 * back as an execution detail of F16X3, nero_f16_paired below: not a mode, the value 3 stays refused.) */
enum { NERO_GEMM_F32 = 0, NERO_GEMM_BF16X6 = 1, NERO_GEMM_F16X3 = 2 };
enum { NERO_OK = 0, NERO_ERR_ARG = -1, NERO_ERR_LAUNCH = -2, NERO_ERR_UNSUPPORTED = -3, NERO_ERR_NOMEM = -4 };

const char* nero_last_error(void);
int nero_version(void);
/* Would a caller-owned workspace of `need_bytes` fit the current device?  The step drivers below never allocate -- the caller hands over ONE
 * workspace sized by nero_stage1_workspace_bytes / nero_stage2_workspace_bytes (31 GiB typical, 52 GiB worst case at 4096 rays) -- so the place
 * to fail cleanly is in front of the caller's allocation: NERO_OK when need_bytes <= free device memory + reusable_bytes (memory the caller
 * will release or re-use for it: its previous workspace, its allocator's cached blocks), else NERO_ERR_NOMEM with the byte counts and `what`
 * in nero_last_error().  (The reference has no counterpart: torch raises an allocator OOM from wherever the step happens to be.) */
int nero_check_device_memory(size_t need_bytes, size_t reusable_bytes, const char* what);
/* per-launch HIP-event timing of the MFMA kernel classes {0 forward, 1 tangent, 2 reverse, 3 weight-gradient GEMM}:
 * report fills out[kind*3 + {0,1,2}] = {launches, total ms, total algorithmic flops} (host array of 12 doubles). */
int nero_prof_enable(int on);
int nero_prof_report(double* out);
/* the same per kernel, 24 doubles: classes 0-3 = the 512-thread chain kernels and the weight-gradient GEMM alone, 4 / 5 / 6 = the forward /
 * tangent / reverse launches that ran on the two-workgroups-per-CU kernels (nero_f16_paired), 7 unused */
int nero_prof_report_kernels(double* out);

/* ---- weight packing -------------------------------------------------------------------------------------------
 * Packs (a column window of) an effective weight matrix W[rows, ld] into the MFMA B-operand order used by the chain
 * kernels:  out[((c*NT + nt)*64 + lane)*4 + t] = B[8c + 4(lane>>5) + t][32nt + (lane&31)] * scale, zero outside.
 *   transpose == 0 ("forward" operand):  B[k][n] = W[n][col0 + k],  K = ncols,  N = nrows
 *   transpose == 1 ("backward" operand): B[k][n] = W[k][col0 + n],  K = nrows,  N = ncols
 * K is padded to a multiple of 8 (kpad), N to 32*nt_count.  Replaces the per-call cuBLAS operand handling inside
 * nn.Linear (network/field.py:142, 266, 325-331). */
int nero_pack_weight(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale,
                     int kpad, int nt_count, float* out, void* stream);
/* NERO_GEMM_BF16X6 operand: same (W, window, transpose, scale) meaning; kpad is a multiple of 16 and the image holds three
 * bf16 planes in A-fragment order, nt_count * (kpad/16) * 3072 bytes:
 *   out[(((t*(kpad/16) + c)*3 + p)*64 + lane)*8 + j] = plane_p(A[32t + (lane&31)][16c + 8(lane>>5) + j]),
 *   transpose == 0: A[m][k] = W[m][col0 + k] (M = nrows, K = ncols);  transpose == 1: A[m][k] = W[k][col0 + m] */
int nero_pack_weight_split(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale,
                           int kpad, int nt_count, void* out, void* stream);

/* Batched form: up to NERO_MAX_PACK_JOBS packing / copy jobs of one network in ONE launch (a training step repacks ~90
 * operand images and copies ~60 bias / head vectors; as separate launches they are pure launch latency).
 *   kind 0: nero_pack_weight_split(W, nrows, ld, col0, ncols, transpose, scale, kpad, nt_count, out)
 *   kind 1: nero_pack_weight(...) with the same fields
 *   kind 2: 2-D copy  out[r*kpad + c] = W[r*ld + c],  r < nrows, c < ncols   (bias: nrows = 1; head weights [n_head][k] -> [4][256])
 *   kind 3: NERO_GEMM_F16X3 operand: out = [256-byte header: float 2^ew][two fp16 planes of A * 2^-ew in A-fragment order,
 *           (((t*(kpad/16) + c)*2 + p)*64 + lane)*16 bytes], ew = exponent of max|A|; `out` must be zero-filled by the caller */
#define NERO_MAX_PACK_JOBS 64
typedef struct {
    const float* W; void* out;
    int kind, nrows, ld, col0, ncols, transpose, kpad, nt_count;
    float scale; int pad_;
} nero_pack_job;
int nero_pack_batch(const nero_pack_job* jobs /*host*/, int n_jobs, void* stream);

/* ---- fused MLP chain ------------------------------------------------------------------------------------------
 * One workgroup owns a tile of 64 rows and walks the whole layer list with the activations resident in LDS; every
 * dense layer is computed with v_mfma_f32_32x32x2_f32.  Replaces the nn.Linear/softplus/ReLU sequences of
 * SDFNetwork.forward (network/field.py:130-147), NeRFNetwork.forward (:258-283), make_predictor (:310-346). */
typedef struct {
    const float* w_main;   /* packed fwd operand [k_main/8][n_tiles][64][4]; input = current LDS activation tile      */
    const float* w_aux;    /* packed fwd operand [k_aux/8][n_tiles][64][4];  input = aux tile (skip connections)      */
    const float* bias;     /* [32*n_tiles] or NULL                                                                    */
    float* save;           /* [rows_pad, 256] post-activation output, or NULL (forward-only evaluation)               */
    const float* head_w;   /* optional VALU head evaluated on this layer's INPUT tile: [n_head][256] row-major        */
    const float* head_b;   /* [n_head]                                                                                */
    float* head_out;       /* [rows_pad, 4]                                                                           */
    int k_main, k_aux;     /* multiples of 8 (0 = unused)                                                             */
    int n_tiles;           /* output width / 32 (0 = head-only pseudo layer)                                          */
    int n_head;            /* 0..4                                                                                    */
    int act;               /* NERO_ACT_*                                                                              */
    int head_k;            /* columns of the input tile the head reads (multiple of 4, <= 256)                        */
    uint32_t* relu_mask;   /* optional [rows_pad, 8] (F16X3 engine, ReLU layers): bit 16h + v of word [row][tile] = output
                              feature 32 tile + 4h + (v&3) + 8(v>>2) is > 0.  The reverse pass then reads 32 B per row
                              instead of the 1 KiB saved activation row (which only the weight-gradient GEMM still needs) */
} nero_fwd_layer;

typedef struct {
    const float* init;     /* [rows_pad, ld_init]: first k_init columns are loaded into the activation tile           */
    const float* aux;      /* [rows_pad, ld_aux]: first k_aux columns are loaded into the aux tile                    */
    int ld_init, k_init, ld_aux, k_aux;
    int n_layers, aux_wide; /* aux_wide: 0 -> aux tile holds <= 40 columns (2 workgroups/CU), 1 -> <= 88              */
    int gemm_mode, pad_;    /* NERO_GEMM_*: the packed operands must have been built for the same mode; BF16X6 wants
                               k_main / k_aux padded to multiples of 16                                               */
    double macs_per_row;    /* logical multiply-adds per row (profiling only)                                         */
    nero_fwd_layer layer[NERO_MAX_LAYERS];
} nero_fwd_chain;

int nero_mlp_forward(const nero_fwd_chain* chain /*host*/, int n_rows, void* stream);

/* Tangent (forward-mode) pass of a softplus chain: same layer walk with  adot_l = sigma'(a_l) * (W_l adot_{l-1})  and
 * inj_l = gbar_l * beta (1 - sigma'(a_l)) * zdot_l.  It is the "second-order" half of the backward of
 * SDFNetwork.gradient (network/field.py:155-167, create_graph=True); SURVEY.md App. E. */
typedef struct {
    const float* w_main; const float* w_aux;
    const float* a_saved;  /* [rows_pad,256] activations saved by the forward pass (for sigma')                       */
    const float* gbar;     /* [rows_pad,256] first-order backward signal saved by the normal pass; may be NULL when   */
                           /* inj is NULL (round 4: the F16X3 reverse kernel forms the injection itself, see          */
                           /* nero_bwd_layer.inj_adot)                                                                 */
    float* adot;           /* out [rows_pad,256]                                                                      */
    float* inj;            /* out [rows_pad,256], or NULL = not written (saves 1 KB read + 1 KB written per row and layer) */
    int k_main, k_aux, n_tiles, pad_;
} nero_tan_layer;

typedef struct {
    const float* init; const float* aux;
    int ld_init, k_init, ld_aux, k_aux;
    int n_layers, aux_wide;
    int gemm_mode, pad_;
    double macs_per_row;
    nero_tan_layer layer[NERO_MAX_LAYERS];
} nero_tan_chain;

int nero_mlp_tangent(const nero_tan_chain* chain /*host*/, int n_rows, void* stream);

/* Reverse pass of a chain: delta_{l-1} = (delta_l W_l [+ dy_head W_head]) * act'(a_{l-1}) [+ inj_{l-1}], every delta_l
 * written out for the weight-gradient GEMMs.  Replaces autograd's AddmmBackward / SoftplusBackward / ReluBackward
 * chain for the same modules. */
typedef struct {
    const float* w_main_t; /* packed bwd operand [n_out/8][k_main/32][64][4]                                          */
    const float* w_aux_t;  /* packed bwd operand [n_out/8][k_aux_tiles][64][4] or NULL                                */
    const float* a_prev;   /* [rows_pad,256] saved activation of the layer's INPUT (NULL for the first layer)          */
    const float* inj;      /* optional additive term for delta_{l-1} [rows_pad,256]                                   */
    float* delta_prev;     /* out [rows_pad,256] (NULL for the first layer)                                           */
    const float* head_w;   /* head on this layer's input tile: [n_head][256]                                          */
    const float* head_dy;  /* [rows_pad,4]                                                                            */
    int n_out;             /* K of the reverse GEMM (multiple of 8; 0 = head-only pseudo layer)                       */
    int k_main_tiles, k_aux_tiles, n_head;
    int act_prev;          /* activation that produced a_prev                                                         */
    int pad_;
    const uint32_t* mask_prev; /* optional relu_mask of the layer that produced a_prev (F16X3 engine): replaces reading a_prev */
    const float* inj_adot; /* F16X3 engine, softplus a_prev only.  When set, `inj` holds the first-order signal gbar_{l-1} and this the
                              tangent adot_{l-1} (both [rows_pad,256]) and the kernel forms the injection itself,
                              gbar * beta (1 - s) * adot / s  with s = sigma'(a_prev) -- the tangent pass then neither reads gbar
                              nor writes inj.  NULL = `inj` is the finished additive term (every engine).                    */
} nero_bwd_layer;

typedef struct {
    const float* dy;       /* [rows_pad, ld_dy] gradient w.r.t. the LAST dense layer's pre-activation output, or NULL */
    int ld_dy, k_dy;
    float* d_init;         /* out [rows_pad, ld_dinit] gradient w.r.t. the init columns (NULL = not needed)           */
    float* d_aux;          /* out [rows_pad, ld_daux]  gradient w.r.t. the aux columns, summed over the layers using it */
    int ld_dinit, ld_daux, accumulate_dinit;
    int n_layers, aux_wide;
    int gemm_mode;
    double macs_per_row;
    nero_bwd_layer layer[NERO_MAX_LAYERS];   /* in FORWARD order; walked from n_layers-1 down to 0                    */
} nero_bwd_chain;

int nero_mlp_backward(const nero_bwd_chain* chain /*host*/, int n_rows, void* stream);
/* Workgroup organisation of the F16X3 chain passes: bit 0 / 1 / 2 of `mask` = the forward / tangent / reverse pass runs with TWO 256-thread
 * workgroups per CU (mlp_f16p.hip: one's epilogue, barriers and HBM waits under the other's MFMAs) instead of one 512-thread workgroup
 * (mlp_f16x3.hip) on launches of more than 4 tiles of 64 rows per CU; bit 3 = on launches of every size.  Same packed operands, same
 * descriptors, the same results bit for bit (tests/test_paired_engine.py).  mask < 0 only queries; returns the previous selection.
 * Initial value: NERO_F16_PAIRED in the environment, default 3.  Process-wide, not thread-safe against concurrent launches.
 * (No reference counterpart: an execution detail of network/field.py's nn.Linear stacks.) */
int nero_f16_paired(int mask);
/* Forward chains that save nothing (no saved activations, no ReLU masks: the sampler's SDF evaluations, forward-only inference) on the
 * row-owner kernel (round 6, mlp_f16r.hip): a wave owns 32 rows and all features, activation planes in registers, weights through an LDS-DMA
 * ring.  Bit-identical results.  mask bit 0 = on for launches of >= 128 rows per CU, bit 1 = whatever the size; < 0 queries; returns the
 * previous selection (default: NERO_F16_ROWOWNER or 0). */
int nero_f16_rowowner(int mask);

/* ---- weight-gradient GEMM ------------------------------------------------------------------------------------
 * dW[n][k] (+)= sum_r D0[r][n] * B0[r][k] (+ sum_r D1[r][n] * B1[r][k]),  db[n] (+)= sum_r D0[r][n]
 * split over row slices; `partials` holds n_slices * (n_out_pad*k_pad + n_out_pad) floats of workspace and is reduced
 * by the same call.  Output dW is row-major [n_out, ldw] written at column offset col0 (so skip-layer parts land in
 * place), scaled by `scale`.  Replaces autograd's mm(grad.t(), input) per Linear. */
typedef struct {
    const float* d0; const float* b0;   /* [rows_pad, ldd0], [rows_pad, ldb0] */
    const float* d1; const float* b1;   /* optional second pair (NULL)        */
    int ldd0, ldb0, ldd1, ldb1;
    int n_out, k_cols;                  /* logical sizes (<= 256 each)        */
    float* dW; int ldw, col0;
    float* db;                          /* [n_out] or NULL                    */
    float scale;
    int accumulate;                     /* 0: overwrite, 1: add into dW/db    */
    int gemm_mode, pad_;                /* NERO_GEMM_* (operands are plain fp32 matrices in every mode).  F16X3: three fp16
                                           plane products, each 16-row chunk block-scaled to the top of fp16's range, one fp32
                                           accumulator with a running unit (mlp_f16dw.hip); BF16X6: six bf16 products */
} nero_dw_job;

int nero_dw_workspace_floats(int n_rows);
int nero_dw_gemm(const nero_dw_job* job /*host*/, int n_rows, float* partials, void* stream);
/* The weight gradients of SEVERAL jobs over the same n_rows rows (all layers of one MLP chain).  Same results as n_jobs calls of
 * nero_dw_gemm up to the order of the row-slice sums: below 131072 rows the fp16 engine runs the jobs as ONE launch per kernel kind
 * (grid.y = job, ~1024 row slices over all jobs instead of 256 per job) plus one batched reduction -- a chain of the reference's own
 * train_ray_num = 512 batch, or the 2 P rows of the Stage-II material MLPs, otherwise pays 2 launches and 256 partial matrices per
 * layer for ~40 k rows.  At or above that row count, and for the other engines, it IS the per-job loop.  partials: as nero_dw_gemm. */
int nero_dw_gemm_batch(const nero_dw_job* jobs /*host*/, int n_jobs, int n_rows, float* partials, void* stream);

/* Head weight gradient: dWh[j][k] = sum_r dy[r][j] a[r][k] (+ extra[r][k] for j == 0 if extra != NULL), dbh[j] = sum_r dy[r][j]. */
int nero_head_dw(const float* dy /*[rows,4]*/, const float* a /*[rows,256]*/, const float* extra, int n_head, int n_rows,
                 float* dWh /*[n_head,256]*/, float* dbh, float* partials, int accumulate, void* stream);
/* the same with an explicit destination shape: dWh [n_head rows of k_cols columns, row pitch ld_dwh] (nero_head_dw: 256 / 256) */
int nero_head_dw_ld(const float* dy, const float* a, const float* extra, int n_head, int n_rows, float* dWh, int ld_dwh, int k_cols, float* dbh,
                    float* partials, int accumulate, void* stream);

/* ---- trainer-loop fusion (replaces the per-Linear nn.utils.weight_norm reparametrisation, network/field.py:
 *      118-119, 323-331, and torch.optim.Adam over ~125 tensors, train/trainer.py:105-170, ~750 tiny kernels per step) -------------
 * One job per weight-normed matrix (dim = 0: one norm per output row).  forward: w_eff = g v / ||v||_row, inv_norm = 1 / ||v||_row.
 * adam: dg = <dW, v> inv_norm, dv = g inv_norm (dW - <dW, v> inv_norm^2 v), then Adam on g and v in place (m_*, v_* = first / second
 * moment buffers).  Plain jobs: Adam on p given grad.  Hyper-parameters and operation order of torch.optim.Adam (fused). */
#define NERO_MAX_WN_JOBS 40
#define NERO_MAX_ADAM_JOBS 96
typedef struct {
    const float* v; const float* g;      /* forward inputs: weight_v [rows, cols], weight_g [rows]                                */
    float* w_eff; float* inv_norm;       /* forward outputs / backward inputs                                                     */
    float* v_rw; float* g_rw;            /* the same parameters, writable (updated in place by the Adam step)                     */
    const float* dW;                     /* dL/dW_eff [rows, cols]                                                                */
    float* m_v; float* v_v; float* m_g; float* v_g;
    int rows, cols;
} nero_wn_job;
/* step: 0 = this job uses the call's `step` argument; > 0 = the job's own 1-based Adam step (bias correction); < 0 = the tensor has no
 * gradient this step: skipped entirely -- torch.optim.Adam skips parameters whose .grad is None, their step counter and moments do not
 * advance (deviation_network.variance while step < freeze_inv_s_step, network/renderer.py:494-495). */
typedef struct { float* p; const float* grad; float* m; float* v; int n; int step; } nero_adam_job;
int nero_wn_forward_batch(const nero_wn_job* jobs /*host*/, int n_jobs, void* stream);
/* The weight-norm backward ALONE, for hosts that keep their own optimiser (the drop-in renderers under torch.optim.Adam: round 6): given
 * dL/dW_eff of every weight-normed matrix, dg = <dW, v> inv_norm and dv = g inv_norm (dW - <dW, v> inv_norm^2 v) in ONE launch -- what
 * autograd does per Linear through torch._weight_norm_interface_backward (nn.utils.weight_norm, network/field.py:118-119, 323-331).
 * inv_norm: the row norms nero_wn_forward_batch left behind. */
typedef struct {
    const float* v; const float* g; const float* inv_norm; const float* dW;
    float* dv; float* dg;                /* out: [rows, cols], [rows]                                                                 */
    int rows, cols;
} nero_wn_grad_job;
int nero_wn_backward_batch(const nero_wn_grad_job* jobs /*host*/, int n_jobs, void* stream);
int nero_wn_adam_batch(const nero_wn_job* wn /*host*/, int n_wn, const nero_adam_job* plain /*host*/, int n_plain, float lr, float beta1,
                       float beta2, float eps, int step /*1-based*/, void* stream);

/* ---- encodings --------------------------------------------------------------------------------------------------
 * Positional encoding rows [x, sin(2^k x), cos(2^k x)]_{k<n_freq}, zero padded to ldo, rows >= n zero
 * (Embedder, network/field.py:14-58). */
int nero_encode_pe(const float* x, int ldx, int dim, int n_freq, int n, float* out, int ldo, void* stream);
/* out[r,0:3] = J_e(x_r)^T (e0[r] + e1[r])  -- input gradient of the SDF network through its PE-6 (the NORMAL,
 * SDFNetwork.gradient, network/field.py:155-167).  e1 may be NULL. */
int nero_pe_vjp(const float* x, int ldx, const float* e0, int ld0, const float* e1, int ld1, int n_freq, int n,
                float* out, int ldo, void* stream);
/* out[r] = J_e(x_r) t_r  (tangent of the PE; seeds the second-order pass), zero padded like nero_encode_pe. */
int nero_pe_jvp(const float* x, int ldx, const float* t, int ldt, int n_freq, int n, float* out, int ldo, void* stream);

/* ---- hierarchical sampling (NeROShapeRenderer.sample_ray / upsample / cat_z_vals, network/renderer.py:355-443;
 *      sample_pdf, network/field.py:399-429).  Per-ray tables: z [R, ldz], sdf [R, lds]. ------------------------------- */
int nero_coarse_z(const float* near, const float* far, const float* rand1 /*[R] or NULL*/, int R, int n, float* z, int ldz, void* stream);
int nero_background_z(const float* far, const float* rand_bg /*[R,n_bg] or NULL*/, int R, int n_bg, float* z, int ldz, int col0, void* stream);
/* PE-6 rows (ld 40) of the points o + d*z[r, col0+j], row = r*ncols + j */
int nero_ray_points_pe(const float* o, const float* d, const float* z, int ldz, int col0, int ncols, int R, float* pe, void* stream);
/* one up-sampling round: m new z per ray from the NeuS section weights of (z, sdf)[0..n).  inv_s = min(exp(10*variance), cap),
 * or cap when variance == NULL.  w_out [R,n-1] / inds_out int32 [R,m] optional (tests). */
int nero_upsample(const float* o, const float* d, const float* z, int ldz, const float* sdf, int lds, int n,
                  const float* variance, float inv_s_cap, int m, int R, float* z_new, float* w_out, int* inds_out, void* stream);
/* deterministic inverse-CDF sampling from given bins/weights; inds = searchsorted(cdf, u, right=True) (bit-exact contract) */
int nero_sample_pdf(const float* bins, int ldb, const float* w, int ldw, int n, int m, int R, float* out, int* inds_out, void* stream);
/* stable in-place merge of sorted z[r,0..n) with sorted z_new[r,0..m); sdf permuted alike (sdf/sdf_new may be NULL);
 * sdf_new is read with stride ldsn; index_out int32 [R,n+m] optional = position in the concatenation [z, z_new] */
int nero_merge_sorted(float* z, int ldz, int n, float* sdf, int lds, const float* z_new, int m, const float* sdf_new, int ldsn,
                      int R, int* index_out, void* stream);
int nero_scatter_sdf(const float* src, int ld_src, int R, int n, float* sdf, int lds, void* stream);

/* ---- occlusion-loss march along the reflected rays (compute_occ_loss, network/renderer.py:522-548; get_intersection /
 *      get_weights / get_sphere_intersection, network/field.py:390-396, 432-484) ----------------------------------------- */
int nero_occ_candidates(const float* x4, const float* sdf4, const float* grad, const int* idx, const float* d, int T, float thresh,
                        int n, unsigned char* flag, void* stream);
int nero_occ_z(const float* o, const float* d, int P, int n, float* z /*[P,n]*/, void* stream);
/* sdf is read as sdf[(p*n+i)*lds] (column 0 of a head output [rows,4] -> lds = 4); w_out [P,n-1] and/or wsum [P] */
int nero_section_weights(const float* z, const float* sdf, int lds, int n, const float* variance, int P, float* w_out, float* wsum,
                         void* stream);

/* ---- render preparation (render_core, network/renderer.py:550-565): mid points, section lengths, inner/outer split ---- */
/* pts4 [R*T,4] = (x,y,z,dist); ray_counts/ray_off int32 [R]; counts int32 [2] = (#inner, #outer) */
int nero_render_prep(const float* o, const float* d, const float* z, int R, int T, float* pts4, int* ray_counts, int* ray_off,
                     int* counts, void* stream);
int nero_compact(const float* pts4, const int* ray_off, int R, int T, int* inner_idx, int* outer_idx, void* stream);
int nero_gather_inner(const float* pts4, const int* idx, int n, float* x4 /*[rows,4]*/, float* pe /*[rows,40]*/, void* stream);
int nero_gather_outer(const float* pts4, const float* d, const int* idx, int T, int n, float* pe88, float* pev32, float* dist, void* stream);

/* ---- inner samples: NeuS alpha + shading frame + eikonal term (compute_sdf_alpha, network/renderer.py:484-512, 574) ---- */
/* geo [rows,8] = { nhat(3), NoV, refl(3), |grad| };  variance: device pointer to deviation_network.variance */
int nero_sdf_alpha_fwd(const float* sdf4, const float* grad, const float* x4, const int* idx, const float* d, int T,
                       const float* variance, float anneal, int n, float* alpha, float* geo, float* gerr, void* stream);
int nero_sdf_alpha_bwd(const float* sdf4, const float* grad, const float* x4, const int* idx, const float* d, int T,
                       const float* variance, float anneal, int n, const float* d_alpha, const float* d_gerr, const float* d_geo,
                       float* d_sdf4, float* d_grad, float* dinv /*[rows_pad] per-sample d inv_s*/, void* stream);

/* ---- split-sum shader algebra (AppShadingNetwork.forward, network/field.py:591-651; IDE utils/ref_utils.py:53-117;
 *      dr.texture field.py:612; linear_to_srgb utils/raw_utils.py:4-10) ------------------------------------------------ */
/* sphere_direction != 0 (shader_config.sphere_direction, field.py:558-562, 582-586): Xd / Xs rows are 144 wide,
 * [IDE(v, .) | IDE(normalised sphere exit point of the ray (offset_points_to_sphere(p), v), .)] for v = normal / reflection */
int nero_shade_encode(const float* x4, const float* geo, const float* m_raw, const float* r_raw, const float* a_raw, int n,
                      float* mat /*[rows,8]*/, float* Xd /*[rows,72|144]*/, float* Xs /*[rows,72|144]*/, float* Xi /*[rows,128]*/,
                      float* Xo /*[rows,96]*/, int sphere_direction, void* stream);
/* Lh [rows,4] raw human-light head + hmask [rows] (from nero_human_encode), or NULL/NULL when shader_config.human_light is off */
int nero_shade_combine_fwd(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                           const float* lut /*[256,256,2]*/, float exp_max, int n, float* color /*[n,3]*/, float* occ_prob,
                           const float* Lh, const float* hmask, void* stream);
/* dmat[k] = { d_metallic, d_rough (LUT part), d_albedo(3), d_NoV (LUT part), 0, 0 }: whole rows.  d_geo is NOT written here (round 5: the
 * argument is kept for the ABI); nero_shade_encode_bwd writes every row of d_geo once: { d_nhat(3), d_NoV = dmat[k][5], d_refl(3), 0 } */
int nero_shade_combine_bwd(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                           const float* lut, float exp_max, int n, const float* d_color, const float* d_occ /*or NULL*/, float* dLd,
                           float* dLs, float* dLi, float* dLo, float* dmat /*[rows,8]*/, float* d_geo /*[rows,8]*/,
                           const float* Lh, const float* hmask, float* dLh, void* stream);
/* validation-only shader intermediates (inter_results=True, network/field.py:630-649): rec [n,32], layout in shade.hip */
int nero_shade_inter_results(const float* geo, const float* mat, const float* Ld, const float* Ls, const float* Li, const float* Lo,
                             const float* lut, float exp_max, int n, const float* Lh, const float* hmask, float* rec, void* stream);
/* extra [rows,4] = { d_refl(3), d_rough } from nero_human_encode_bwd, or NULL */
int nero_shade_encode_bwd(const float* geo, const float* mat, const float* dXd, const float* dXs, const float* dXi, const float* dmat,
                          int n, float* d_geo, float* dm_raw, float* dr_raw, float* da_raw, const float* extra,
                          const float* x4 /*sample positions [rows,4]; needed when sphere_direction*/, int sphere_direction, void* stream);
/* human ("photo capturer") light input (predict_human_light, network/field.py:536-552; get_camera_plane_intersection :348-367;
 * IPE :369-378): Xh [rows,24], hmask [rows]; poses [R,3,4] human-frame poses per RAY, sample k belongs to ray idx[k]/T */
int nero_human_encode(const float* x4, const float* geo, const float* mat, const int* idx, int T, const float* poses, int n,
                      float* Xh, float* hmask, void* stream);
int nero_human_encode_bwd(const float* x4, const float* geo, const float* mat, const int* idx, int T, const float* poses, int n,
                          const float* dXh, float* extra, void* stream);

/* ---- NeRF++ head (compute_density_alpha, network/renderer.py:514-520) and compositing (renderer.py:578-579) ----------- */
int nero_nerf_head_fwd(const float* sig4, const float* rgb4, const float* dist, int n, float* alpha, float* color, void* stream);
int nero_nerf_head_bwd(const float* sig4, const float* rgb4, const float* dist, int n, const float* d_alpha, const float* d_color,
                       float* d_sig4, float* d_rgb4, void* stream);
int nero_scatter_samples(const float* a, const float* c, const int* idx, int n, float* alphaRT, float* colorRT, void* stream);
int nero_composite_fwd(const float* alphaRT, const float* colorRT, int R, int T, float* weights, float* rgb, void* stream);
int nero_composite_bwd(const float* alphaRT, const float* colorRT, const float* weights, const float* d_rgb, int R, int T,
                       float* d_alphaRT, float* d_colorRT, void* stream);
int nero_gather_sample_grads(const float* d_alphaRT, const float* d_colorRT, const int* idx, int n, float* d_a, float* d_c, void* stream);

/* ---- mesh ray tracer (Stage II).  Replaces the third-party CUDA extension `_raytracing` behind raytracing/raytracer.py:
 *      create_raytracer(vertices, triangles) (:19) and impl.trace(rays_o, rays_d, positions, face_normals, depth) (:49).
 *      verts [nV,3] float32 / tris [nT,3] int32 are HOST arrays; rays and outputs are device arrays.  Closest hit with t > 0;
 *      face normal = normalize(cross(v1-v0, v2-v0)); a miss reports depth = 10, position = o + 10 d, normal = 0
 *      (NeROMaterialRenderer.trace treats depth >= 10 as a miss, network/renderer.py:727). */
int nero_bvh_create(const float* verts, int nV, const int* tris, int nT, void** handle);
int nero_bvh_trace(void* handle, const float* rays_o, const float* rays_d, int n, float* positions, float* face_normals, float* depth,
                   void* stream);
/* nero_bvh_trace with a LAUNCH-ORDER hint (round 5): the rays come in groups of `group` (Stage II: the D = Dd + Ds secondary rays of a
 * surface point, field.py:856-880) whose entries [heavy_from, group) are the expensive ones (the GGX specular directions: the only rays
 * that can point below the surface and cross the inside of the mesh); their 64-ray chunks are started first, so that the launch ends
 * with short rays.  Outputs identical to nero_bvh_trace's.  group, heavy_from multiples of 64, 0 < heavy_from < group, n % group == 0;
 * otherwise the natural order. */
int nero_bvh_trace_grouped(void* handle, const float* rays_o, const float* rays_d, int n, float* positions, float* face_normals, float* depth,
                           int group, int heavy_from, void* stream);
/* nero_bvh_trace for a caller that knows some rays' results will not be used (round 6): skip [n] bytes, non-zero = do not traverse; such a
 * ray is reported as a miss (depth 10, zero normal).  Every other ray's outputs are nero_bvh_trace's bit for bit; skip = NULL: the same call.
 * Stage II passes the flags of nero_mc_dead_rays: GGX-sampled directions below the shading horizon, whose estimator weight is exactly zero
 * (field.py:892-903, 979-987 with geometry_type 'schlick') and which are the longest rays of the launch (they cross the inside of the mesh). */
int nero_bvh_trace_masked(void* handle, const float* rays_o, const float* rays_d, int n, const unsigned char* skip, float* positions,
                          float* face_normals, float* depth, void* stream);
/* nero_bvh_trace_masked (skip may be NULL) with an explicit LAUNCH ORDER: rays in groups of n_order * 64; phase p of the launch holds chunk
 * order[p] of every group (a permutation of 0 .. n_order-1, n_order <= 32, n % (64 n_order) == 0; anything else falls back to the natural order).
 * Outputs identical to nero_bvh_trace_masked's.  Stage II starts the chunks in DESCENDING order: the later entries of both direction tables
 * (field.py:741-749) are the grazing directions, the long traversals; with them first the launch no longer ends on a handful of waves. */
int nero_bvh_trace_ordered(void* handle, const float* rays_o, const float* rays_d, int n, const unsigned char* skip, const int* order, int n_order,
                           float* positions, float* face_normals, float* depth, void* stream);
int nero_bvh_destroy(void* handle);
/* which kernel nero_bvh_trace launches: 1 = memory requests of a traversal step overlapped, stack in LDS (default when the tree is no
 * deeper than the 24-entry LDS stack), 0 = private stack, one request after the other.  Same visit order and arithmetic per ray:
 * bit-identical outputs. */
int nero_bvh_set_traversal(void* handle, int mode);

/* ---- Stage-II Monte-Carlo shading glue (MCShadingNetwork.shade_mixed and helpers, network/field.py:756-1012).
 *      Row r = p*D + j, D = Dd + Ds (j < Dd cosine-weighted diffuse samples, then GGX specular samples).
 *      pt [P,32] per-point record (layout in mc_shade.hip); mat5 [P,5] = metallic, roughness, albedo(3);
 *      tab_d [Dd,2] / tab_s [Ds,2] = fixed (azimuth, elevation) tables in [0,1] (field.py:741-749);
 *      slot[r] >= 0 -> miss row index, < 0 -> hit row index -(slot)-1. ---------------------------------------------------- */
int nero_mc_point_setup(const float* pts, const float* view, const float* normals, const float* mat5, const float* rand_d,
                        const float* rand_s, int P, float* pt, void* stream);
/* hit / miss split of the n = P*D secondary rays by `depth < 10` (NeROMaterialRenderer.trace, network/renderer.py:727; get_lights,
 * network/field.py:861-877): miss_idx / hit_idx = the ray ids in ascending order (what torch.nonzero yields), slot as above, counts int32 [2]
 * = (n_miss, n_hit) on the device; miss_idx / hit_idx need room for n entries each; tmp: nero_mc_split_tmp_ints(n) int32 of scratch. */
int nero_mc_split_tmp_ints(int n);
int nero_mc_split(const float* depth, int n, int* slot, int* miss_idx, int* hit_idx, int* counts, int* tmp, void* stream);
/* DEAD rays (round 6).  With geometry_type 'schlick' (the reference's default, field.py:702; the only one its configurations use) a direction
 * below the shading horizon has NoL = saturate(n.w) = 0, hence G = 0 and an estimator weight D G / (4 NoV p + 1e-5) of EXACTLY zero, with
 * zero derivatives (the clamp passes no gradient for n.w < 0): the ray contributes nothing to any output or gradient of shade_mixed
 * (field.py:950-1012), whatever it hits.  nero_mc_dead_rays: dead [P*D] bytes = 1 for the GGX-sampled rows with n.w < -1e-6 (0 everywhere for
 * geometry_type 1, whose weight at NoL = 0 is small but not zero).  nero_mc_split_dead: nero_mc_split with a third class -- flagged rays get
 * slot = INT_MIN and enter neither list (counts[0] + counts[1] = n - #dead); the estimator kernels read L = 0 for them; dead = NULL: nero_mc_split. */
int nero_mc_dead_rays(const float* pt, const float* dirs, int P, int Dd, int Ds, int geometry_type, unsigned char* dead, void* stream);
int nero_mc_split_dead(const float* depth, const unsigned char* dead, int n, int* slot, int* miss_idx, int* hit_idx, int* counts, int* tmp,
                       void* stream);

/* ---- Stage-II training glue (nero_amd/csrc/mat_loss.hip): what NeROMaterialRenderer.train_step does between the MLP / shading kernels,
 * as single launches.
 * nero_mat_reg_points: out [2P,3] = [pts ; pts + (cos(a) x + sin(a) y) eps], a = 2 pi ang01[p], (x, y) the tangent frame of
 *   get_orthogonal_directions (network/field.py:756-766, 1066-1076); eps [P] (change_type 'gaussian') or NULL -> eps_const.
 * nero_mat_head_fwd / _bwd: raw [n,5] -> (metallic, roughness in [0.04^2, 1], albedo[3]) = sigmoid heads of predict_materials
 *   (network/field.py:915-922) and their backward (d_raw from d_mat; raw is re-read, not the output).
 * nero_mat_loss_fwd: loss[0] = mean(loss_rgb) + mean(loss_mat_reg) + mean(loss_diffuse_light) (train/trainer.py:134-137 over
 *   network/renderer.py:837-844); loss[1..3] = the three means; rgb_pr [P,3] = linear_to_srgb(rgb_lin) (may be NULL).
 *   mat: [2P,5] when has_reg (rows P.. = the materials at the perturbed points), else [P,5]; partials: nero_mat_loss_partials(P) floats.
 *   hinge_weight: 0 = the reg_min_max hinge is off (step >= 2000), otherwise the weight of the hinge SUM (1 for one process; the
 *   data-parallel step passes world size).
 * nero_mat_loss_bwd: gradients of loss[0] * grad_out[0] (grad_out: device scalar, NULL = 1) -> d_mat (same shape as mat), d_rgb_lin, d_dl. */
typedef struct {
    int rgb_l1;                 /* 0: Charbonnier sqrt(sum d^2 + 1e-3) (default), 1: L1 */
    int reg_mat, reg_change;
    float reg_lambda1;
    float hinge_weight;
    int reg_diffuse;
    float reg_diffuse_lambda;
} nero_mat_loss_cfg;
int nero_mat_reg_points(int P, const float* pts, const float* normals, const float* ang01, const float* eps, float eps_const, float* out, void* stream);
int nero_mat_head_fwd(int n, const float* raw, float* mat, void* stream);
int nero_mat_head_bwd(int n, const float* raw, const float* d_mat, float* d_raw, void* stream);
int nero_mat_loss_partials(int P);
int nero_mat_loss_fwd(const nero_mat_loss_cfg* cfg, int P, int has_reg, const float* mat, const float* rgb_lin, const float* dl, const float* gt,
                      float* rgb_pr, float* partials, float* loss, void* stream);
int nero_mat_loss_bwd(const nero_mat_loss_cfg* cfg, int P, int has_reg, const float* mat, const float* rgb_lin, const float* dl, const float* gt,
                      const float* grad_out, float* d_mat, float* d_rgb_lin, float* d_dl, void* stream);
/* ---- Stage-I training glue (nero_amd/csrc/step_glue.hip): what NeROShapeRenderer.train_step / the trainer do BETWEEN the render calls,
 *      as a handful of launches and without a second host read-back.  Replaces, per step: near_far_from_sphere
 *      (network/renderer.py:231-238), the candidate subset of compute_occ_loss (network/renderer.py:528-541: mask -> nonzero ->
 *      random subset of occ_loss_max_pn), compute_rgb_loss + the means of loss_rgb / loss_eikonal / loss_occ and their sum
 *      (network/renderer.py:332-343, network/loss.py:8-55, train/trainer.py:127-137) and autograd's seeds for the render backward.
 * nero_occ_select: flag [n] (nero_occ_candidates) -> counts = (kept = min(total, cap), total); cand [cap] = the sample indices of the
 *   kept candidates in ascending order, -1 in unused slots.  Above the cap the kept ones are the `cap` smallest keys[ordinal]
 *   (ordinal = rank of the candidate among the candidates; ties to the lower ordinal): argsort(keys[:total], stable)[:cap], sorted.
 *   Nothing is read back: every launch covers the worst case n.  ws: nero_occ_select_workspace(n) bytes.
 * nero_occ_gather: pts / dirs [cap,3] = x4[cand, 0:3] / geo[cand, 4:7] (the reflected direction); unused slots: origin, +z.
 * nero_occ_l1: loss[0] = sum over the used slots k of |occ_prob[cand[k]] - gt[k]| / max(counts[0], 1) -- F.l1_loss(occ_prob[cand], gt) of
 *   network/renderer.py:546-547 on the fixed-capacity candidate list (one block, fixed summation tree); nero_occ_l1_backward: d_occ [n_in] =
 *   d_loss[0] * sign(occ_prob[cand[k]] - gt[k]) / max(counts[0], 1) at the kept candidates, zero elsewhere (sign(0) = 0 as in ATen).  For
 *   callers that assemble the loss themselves under autograd (the drop-in renderer); nero_shape_loss below holds the same term for the fused step.
 * nero_shape_loss: losses [4] = (total, mean loss_rgb, eik_weight * mean(gerr) * w[0], L1(occ_prob[cand], gt_occ) * w[1]);
 *   d_rgb [R,3], d_gerr [n_in], d_occ [n_in] (zero except at the kept candidates) = d total / d (ray_rgb, gradient_error, occ_prob).
 *   cand = NULL: no occlusion term (d_occ may be NULL).  weights: device [2] (data-parallel count weights) or NULL = (1, 1).
 *   partials: nero_shape_loss_partials(R, n_in) floats.
 * nero_var_grad: grad[0] = dsum[0] * 10 * inv_s * [1e-6 <= inv_s <= 1e6], inv_s = exp(10 variance[0])  (std_act 'exp'). */
enum { NERO_RGB_L2 = 0, NERO_RGB_L1 = 1, NERO_RGB_SMOOTH_L1 = 2, NERO_RGB_CHARBONIER = 3 };
int nero_near_far_sphere(const float* o, const float* d, int R, float* near, float* far, void* stream);
size_t nero_occ_select_workspace(int n);
int nero_occ_select(const unsigned char* flag, int n, const float* keys, int cap, int* cand, int* counts, void* ws, size_t ws_bytes,
                    void* stream);
int nero_occ_gather(const float* x4, const float* geo, const int* cand, int cap, float* pts, float* dirs, void* stream);
int nero_occ_l1(const float* occ_prob, const int* cand, const int* counts, const float* gt, int cap, float* loss, void* stream);
int nero_occ_l1_backward(const float* d_loss, const float* occ_prob, const int* cand, const int* counts, const float* gt, int cap, int n_in,
                         float* d_occ, void* stream);
int nero_shape_loss_partials(int R, int n_in);
int nero_shape_loss(int R, int rgb_kind, const float* rgb, const float* gt, int n_in, const float* gerr, float eik_weight,
                    const float* occ_prob, const int* cand, const int* counts, const float* gt_occ, const float* weights, float* losses,
                    float* d_rgb, float* d_gerr, float* d_occ, float* partials, void* stream);
int nero_var_grad(const float* dsum, const float* variance, float* grad, void* stream);
int nero_mc_dirs(const float* pt, const float* tab_d, const float* tab_s, int P, int Dd, int Ds, float* dirs, float* origins, void* stream);
/* sphere != 0 ('sphere_direction'): X [rows,144] = [IDE(w,0) | IDE(unit-sphere exit point,0)], else X [rows,72] */
int nero_mc_encode_miss(const float* dirs, const int* idx, const float* pt, int D, int sphere, int n, float* X, void* stream);
/* human-light input of the miss rows (get_human_light, network/field.py:820-834): Xh [rows,24], hmask [rows]; poses [P,3,4] */
int nero_mc_human_encode(const float* dirs, const int* idx, const float* pt, int D, const float* poses, int n, float* Xh, float* hmask,
                         void* stream);
int nero_mc_encode_hit(const float* dirs, const float* pos, const float* face_normals, const int* idx, int n, float* X /*[rows,128]*/,
                       void* stream);
/* human_raw [miss rows,4] / hmask [miss rows] may be NULL (shader_cfg.human_lights false).
 * geometry_type: 0 = 'schlick' (geometry_schlick, network/field.py:892-903), 1 = 'ggx_smith' (geometry_ggx_smith_correlated, :905-913) */
int nero_mc_combine_fwd(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                        const float* inner_raw, const float* human_raw, const float* hmask, float exp_max, float inner_exp_max, int P,
                        int Dd, int Ds, int geometry_type, float* rgb_lin, float* dl_mean, float* sl_mean, float* spec_lin /*or NULL*/,
                        void* stream);
int nero_mc_combine_bwd(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                        const float* inner_raw, const float* human_raw, const float* hmask, float exp_max, float inner_exp_max, int P,
                        int Dd, int Ds, int geometry_type, const float* d_rgb, const float* d_dl, float* d_outer_raw, float* d_inner_raw,
                        float* d_human_raw, float* d_mat5, float* d_wspec, void* stream);
int nero_mc_dir_bwd(const float* pt, const float* dirs, const float* face_normals, const int* slot, const float* tab_s,
                    const float* dX_miss, const float* dX_hit, const float* d_wspec, int P, int Dd, int Ds, float* d_mat5, int sphere,
                    const float* dXh /*or NULL*/, const float* poses /*[P,3,4] or NULL*/, void* stream);
/* Round 6: the photographer's light exists only where a ray reaches his region of the camera plane -- get_human_light multiplies the MLP's
 * output by that mask (field.py:819-829), so every other row of the human-light MLP is exactly dead.  nero_mc_human_flags: hum [P*D] bytes =
 * the mask per ray (hits & |0.3 inter_xy| < 1.5 & dist > 0).  nero_mc_split_classes: nero_mc_split_dead (dead may be NULL) with the miss
 * list PARTITIONED by `hum`: the misses with hum != 0 first (ray order), then the others (ray order); counts3 int32 [3] = (n_miss, n_hit, n_hum).
 * The human-light MLP then runs on the miss rows [0, n_hum); the _h entry points take n_hum and read human_raw / hmask / dXh / write d_human_raw
 * for those rows only (the plain entry points: for every miss row). */
int nero_mc_human_flags(const float* pt, const float* dirs, const float* poses, int P, int D, unsigned char* hum, void* stream);
int nero_mc_split_classes(const float* depth, const unsigned char* dead, const unsigned char* hum, int n, int* slot, int* miss_idx, int* hit_idx,
                          int* counts3, int* tmp, void* stream);
int nero_mc_combine_fwd_h(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                          const float* inner_raw, const float* human_raw, const float* hmask, int n_hum, float exp_max, float inner_exp_max, int P,
                          int Dd, int Ds, int geometry_type, float* rgb_lin, float* dl_mean, float* sl_mean, float* spec_lin /*or NULL*/,
                          void* stream);
int nero_mc_combine_bwd_h(const float* pt, const float* dirs, const float* depth, const int* slot, const float* outer_raw,
                          const float* inner_raw, const float* human_raw, const float* hmask, int n_hum, float exp_max, float inner_exp_max, int P,
                          int Dd, int Ds, int geometry_type, const float* d_rgb, const float* d_dl, float* d_outer_raw, float* d_inner_raw,
                          float* d_human_raw, float* d_mat5, float* d_wspec, void* stream);
int nero_mc_dir_bwd_h(const float* pt, const float* dirs, const float* face_normals, const int* slot, const float* tab_s,
                      const float* dX_miss, const float* dX_hit, const float* d_wspec, int P, int Dd, int Ds, float* d_mat5, int sphere,
                      const float* dXh /*or NULL*/, const float* poses /*[P,3,4] or NULL*/, int n_hum, void* stream);

/* ---- C-level driver of the Stage-I render step (nero_stage1_render_fwd / _bwd, nero_workspace_bytes) -------------------
 * One call each for sample_ray (network/renderer.py:403-443), render / render_core (:445-463, 550-606: the drop-in boundary) and
 * their backward incl. the second-order SDF term (autograd through SDFNetwork.gradient, network/field.py:155-167).  The calls
 * sequence the entry points above exactly as nero_amd/shape_step.py does (tests/test_stage1_driver.py: bit-for-bit equal), so a host in
 * any language runs the step without Python.  ALL memory is the caller's: one workspace (nero_stage1_workspace_bytes) from which every
 * intermediate is carved, one buffer for the packed operand images (nero_stage1_pack_bytes).  The only host synchronisation is the
 * read-back of the inner / outer sample counts inside nero_stage1_render_fwd (they size every later launch).
 * The fp16 two-plane engines only (gemm_* = NERO_GEMM_F16X3); NERO_ERR_UNSUPPORTED otherwise. */
#define NERO_S1_LINEARS 49
/* index of a Linear in nero_stage1_weights / _grads: sdf_network.lin0..8 = 0..8, outer_nerf.pts_linears.0..7 = 9..16, views_linears.0 = 17,
 * feature_linear = 18, alpha_linear = 19, rgb_linear = 20, then the predictors' four Linears each: metallic 21.., roughness 25.., albedo
 * 29.., outer_light 33.., inner_light 37.., inner_weight 41.., human_light_predictor 45.. (shader_config.human_light only). */
typedef struct { const float* W; const float* b; } nero_linear;        /* EFFECTIVE weight [n_out, k] row-major contiguous, bias [n_out] */
typedef struct { float* dW; float* db; } nero_linear_grad;              /* destinations shaped like W / b (overwritten)                   */
typedef struct { nero_linear lin[NERO_S1_LINEARS]; } nero_stage1_weights;
typedef struct { nero_linear_grad lin[NERO_S1_LINEARS]; } nero_stage1_grads;
typedef struct {
    int n_samples, n_importance, n_bg_samples, up_sample_steps, clip_sample_variance;   /* NeROShapeRenderer.default_cfg, renderer.py:84-95 */
    int human_light, sphere_direction;                                                  /* shader_config, network/field.py:487-495          */
    float light_exp_max;
    int gemm_fwd, gemm_tan, gemm_bwd, gemm_dw;                                           /* NERO_GEMM_*                                      */
} nero_stage1_cfg;
/* device pointers into the workspace, valid from nero_stage1_render_fwd until the next one (what the losses and the validation path read:
 * compute_occ_loss renderer.py:522-548 needs x4 / sdf4 / normal / geo / inner_idx, compute_validation_info :465-482 needs weights) */
typedef struct {
    int R, T, n_in, n_out;
    float* pts4;            /* [R*T,4] mid points + section lengths                       */
    int* ray_counts; int* ray_off; int* counts;
    int* inner_idx; int* outer_idx;
    float* x4;              /* [rows_pad(n_in),4] inner sample positions                  */
    float* sdf4;            /* [rows_pad(n_in),4] column 0 = sdf                          */
    float* feat;            /* [rows_pad(n_in),256]                                       */
    float* normal;          /* [n_in,3] SDF gradient                                      */
    float* geo;             /* [rows_pad(n_in),8] nhat, NoV, reflection, |grad|           */
    float* weights;         /* [R,T] compositing weights                                  */
} nero_stage1_state;
typedef struct nero_stage1 nero_stage1;

int nero_stage1_create(const nero_stage1_cfg* cfg, nero_stage1** out);
void nero_stage1_destroy(nero_stage1* h);
size_t nero_stage1_pack_bytes(nero_stage1* h);
/* (re)build the packed operand images of all ten networks from the effective weights: once per optimisation step */
int nero_stage1_pack(nero_stage1* h, const nero_stage1_weights* w, void* pack_buf, void* stream);
/* worst case over the data-dependent inner / outer split for R rays (sampler + forward + backward) */
size_t nero_stage1_workspace_bytes(nero_stage1* h, int R);
size_t nero_stage1_workspace_bytes_for(nero_stage1* h, int R, int n_in, int n_out, int with_sampler);
/* sampler + render forward only (inference chunks: no nero_stage1_render_bwd on this workspace), worst case over the split */
size_t nero_stage1_workspace_bytes_fwd(nero_stage1* h, int R);
/* debug: (device pointer, bytes) of nine intermediates of the last nero_stage1_render_bwd -- d_geo, d_feat, d_sdf4, d_grad, dinv, ehat, adot,
 * d_alpha_inner, d_metallic_raw -- for run-to-run comparisons (scripts/r05/dbg_streams.py) */
int nero_stage1_debug_buffers(nero_stage1* h, const void** ptrs, size_t* bytes);
/* z_vals [R, n_samples + n_importance + n_bg_samples]; rand1 [R] / rand_bg [R, n_bg] uniform draws or NULL (no perturbation) */
int nero_stage1_sample(nero_stage1* h, int R, const float* o, const float* d, const float* near, const float* far, const float* variance,
                       const float* rand1, const float* rand_bg, float* z_vals, void* ws, size_t ws_bytes, void* stream);
/* rgb [R,3]; gerr / occ_prob: capacity R*T floats, the first n_in entries are written (eikonal term, unclamped occ_prob);
 * n_in_out / n_out_out: host ints.  poses [R,3,4] human frames or NULL.  Keeps its state in `ws` for nero_stage1_render_bwd. */
int nero_stage1_render_fwd(nero_stage1* h, int R, const float* o, const float* d, const float* z_vals, const float* variance,
                           const float* lut, const float* poses, float anneal, float* rgb, float* gerr, float* occ_prob,
                           int* n_in_out, int* n_out_out, void* ws, size_t ws_bytes, void* stream);
/* d_rgb [R,3], d_gerr [n_in] or NULL, d_occ [n_in] or NULL -> every dW / db of `grads` (NULL entries are skipped);
 * d_inv_s_sum: device float receiving sum_k d L / d inv_s (the caller applies d inv_s / d variance), or NULL */
int nero_stage1_render_bwd(nero_stage1* h, const float* d_rgb, const float* d_gerr, const float* d_occ, const nero_stage1_grads* grads,
                           float* d_inv_s_sum, void* stream);
int nero_stage1_get_state(nero_stage1* h, nero_stage1_state* out);
/* no-grad SDF values of PE-6 rows [rows_pad(n),40] -> out4 [rows_pad(n),4], column 0 = sdf (sampler / occlusion march / mesh grid) */
int nero_stage1_sdf_from_pe(nero_stage1* h, const float* pe, int n, float* out4, void* ws, size_t ws_bytes, void* stream);

/* ---- C-level driver of the Stage-II (material) shading step (nero_mc_shade_fwd / _bwd) -----------------------------------
 * predict_materials (network/field.py:915-922) and MCShadingNetwork.shade_mixed / get_lights (:856-880, 950-1012) with their backward, in
 * the call order of NeROMaterialRenderer.shade (network/renderer.py:810-813):
 *     nero_stage2_predict_fwd   raw material heads of the points (and of the regulariser's perturbed copies)
 *     nero_stage2_rays          tangent frames, cosine-weighted + GGX directions, secondary-ray origins p + 1e-5 w
 *     <the caller traces `origins, dirs` -- nero_bvh_trace, where the reference calls raytracing/raytracer.py:49 from field.py:860>
 *     nero_stage2_shade_fwd     hit / miss split, light MLPs on the compacted rows, microfacet estimator
 *     nero_stage2_shade_bwd, nero_stage2_predict_bwd   (autograd order: the shading first)
 * One caller-owned workspace for the whole step (opened by predict_fwd), one buffer for the packed operand images; the only host
 * synchronisation is the read-back of the (miss, hit) counts in shade_fwd.  fp16 two-plane engines only. */
#define NERO_S2_LINEARS 32
/* Linear index: feats_network.module0 / module1 Linears = 0..7, then the predictors' four each: metallic 8.., roughness 12.., albedo 16..,
 * outer_light 20.., inner_light 24.., human_light 28.. (shader_cfg.human_lights only) */
typedef struct { nero_linear lin[NERO_S2_LINEARS]; } nero_stage2_weights;
typedef struct { nero_linear_grad lin[NERO_S2_LINEARS]; } nero_stage2_grads;
typedef struct {
    int diffuse_sample_num, specular_sample_num, human_lights, sphere_direction;    /* MCShadingNetwork.default_cfg, network/field.py:695-712 */
    int geometry_type;                                                              /* 0 'schlick', 1 'ggx_smith'                             */
    float light_exp_max, inner_light_exp_max;
    int gemm_fwd, gemm_bwd, gemm_dw;
} nero_stage2_cfg;
typedef struct nero_stage2 nero_stage2;
int nero_stage2_create(const nero_stage2_cfg* cfg, nero_stage2** out);
void nero_stage2_destroy(nero_stage2* h);
size_t nero_stage2_pack_bytes(nero_stage2* h);
int nero_stage2_pack(nero_stage2* h, const nero_stage2_weights* w, void* pack_buf, void* stream);
/* n_pred rows through predict_materials (P points + their P perturbed copies when the smoothness regulariser is on), P shaded points:
 * worst case over the hit / miss split of the P * (Dd + Ds) light rays */
size_t nero_stage2_workspace_bytes(nero_stage2* h, int n_pred, int P);
/* x [n,3] -> raw5 [n,5] = pre-sigmoid (metallic, roughness, albedo rgb) */
int nero_stage2_predict_fwd(nero_stage2* h, const float* x, int n, float* raw5, void* ws, size_t ws_bytes, void* stream);
/* mat5 [P,5] = (metallic, roughness, albedo) after their activations; rand_d / rand_s [P] azimuth offsets or NULL; tab_d [Dd,2] / tab_s [Ds,2]
 * the fixed (azimuth, elevation) tables (field.py:741-749) -> origins, dirs [P*(Dd+Ds),3] (caller-owned, kept alive until shade_bwd) */
int nero_stage2_rays(nero_stage2* h, int P, const float* pts, const float* view, const float* normals, const float* mat5, const float* rand_d,
                     const float* rand_s, const float* tab_d, const float* tab_s, float* origins, float* dirs, void* stream);
/* the flags of the rays of the last nero_stage2_rays call whose estimator weight is exactly zero (nero_mc_dead_rays; device pointer, [P*D]
 * bytes, valid until the next nero_stage2_rays), for the caller's tracer (nero_bvh_trace_masked); NULL when nothing is skipped (geometry_type 1,
 * or NERO_MC_SKIP_DEAD=0 at nero_stage2_create).  nero_stage2_shade_fwd leaves the flagged rays out of both light MLPs whatever the tracer reported. */
const unsigned char* nero_stage2_dead_rays(nero_stage2* h);
/* rows of the last nero_stage2_shade_fwd: miss rows (outer light), hit rows (inner light), and how many of the miss rows own a row of the
 * human-light MLP (round 6: the rays that reach the photographer's region of the camera plane; 0 without human_lights).  Rays that are none of
 * these are the zero-weight rays nobody shades. */
int nero_stage2_counts(nero_stage2* h, int* n_miss, int* n_hit, int* n_hum);
/* pos / face_normals [P*D,3], depth [P*D] of the traced rays (depth >= 10 = miss; kept alive until shade_bwd); poses [P,3,4] or NULL
 * -> rgb (linear), mean diffuse light, mean weighted specular light, specular part: [P,3] each */
int nero_stage2_shade_fwd(nero_stage2* h, const float* pos, const float* face_normals, const float* depth, const float* poses, float* rgb,
                          float* dl, float* sl, float* sp, int* n_miss_out, int* n_hit_out, void* stream);
/* -> dW / db of the light MLPs (entries 20.. of `grads`), d_mat5 [P,5] */
int nero_stage2_shade_bwd(nero_stage2* h, const float* d_rgb, const float* d_dl, const nero_stage2_grads* grads, float* d_mat5, void* stream);
/* d_raw5 [n,5] -> dW / db of the feats network and the material predictors (entries 0..19) */
int nero_stage2_predict_bwd(nero_stage2* h, const float* d_raw5, const nero_stage2_grads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif
