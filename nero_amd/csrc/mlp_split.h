// mlp_split.h -- internal entry points of the split-bf16 engine (mlp_split.hip), dispatched from the C-ABI functions in
// mlp_engine.hip when a chain / job asks for gemm_mode NERO_GEMM_BF16X6
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/nero_hip.h"

int nero_split_pack(const float* W, int nrows, int ld, int col0, int ncols, int transpose, float scale, int kpad, int nt_count,
                    void* out, hipStream_t stream);
int nero_split_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream);
int nero_split_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream);
int nero_split_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream);
int nero_split_dw(const nero_dw_job* job, int n_rows, int rows_per_slice, int slices, float* partials, int n_pad, int k_pad,
                  hipStream_t stream);
int nero_split_pack_batch(const nero_pack_job* jobs, int n_jobs, hipStream_t stream);
// fp16 two-plane engine (mlp_f16x3.hip)
int nero_f16_pack_batch(const nero_pack_job* jobs, int n_jobs, hipStream_t stream);
int nero_f16_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream);
int nero_f16_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream);
int nero_f16_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream);
// the same passes with two 256-thread workgroups per CU (mlp_f16p.hip): same operands, same results bit for bit; chosen per pass by
// NERO_F16_PAIRED (bit 0 forward, 1 tangent, 2 reverse) inside nero_f16_forward / _tangent / _backward
int nero_f16p_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream);
int nero_f16p_tangent(const nero_tan_chain* ch, int n_rows, hipStream_t stream);
int nero_f16p_backward(const nero_bwd_chain* ch, int n_rows, hipStream_t stream);
bool nero_f16p_masks_only(const nero_bwd_chain* ch);     // ReLU / identity chain with sign words: the spill-free paired reverse kernel applies
// forward chains that save nothing, with the chain walk re-cut so that a wave owns rows (mlp_f16r.hip; NERO_F16_ROWOWNER inside nero_f16_forward)
bool nero_f16r_covers(const nero_fwd_chain* ch);
int nero_f16r_forward(const nero_fwd_chain* ch, int n_rows, hipStream_t stream);
int nero_f16_dw(const nero_dw_job* job, int n_rows, int rows_per_slice, int slices, float* partials, int n_pad, int k_pad,
                hipStream_t stream);                                                    // mlp_f16dw.hip
// several weight-gradient jobs over the SAME rows in one launch (grid.y = job): kernel argument of dw_f16_batch_kernel / dw_reduce_batch_kernel
#define NERO_DW_BATCH_MAX 12
struct nero_dw_batch {
    nero_dw_job j[NERO_DW_BATCH_MAX];
    unsigned long long poff[NERO_DW_BATCH_MAX];      // float offset of the job's partial matrices inside `partials`
    short n_pad[NERO_DW_BATCH_MAX], k_pad[NERO_DW_BATCH_MAX];
};
int nero_f16_dw_batch(const nero_dw_batch* B, int n_jobs, int narrow, int n_rows, int rows_per_slice, int slices, float* partials, hipStream_t stream);
