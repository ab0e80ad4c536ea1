// mlp_f16dw.hip -- the weight-gradient GEMM on the fp16 matrix pipe with THREE plane products per fp32 multiply-add
// (nero_dw_job.gemm_mode NERO_GEMM_F16X3; the bf16x6 kernel of mlp_split.hip needs six).
//
//   C[n][k] = sum_r D[r][n] B[r][k]        (contraction over the batch rows, split over row slices; dw_reduce_kernel adds the slices)
//
// Same staging as dw_split_kernel: a chunk of 16 batch rows per step, thread (col, rh) loads rows 8rh..8rh+7 of its column of D
// and of B, converts them and writes one 16-byte LDS entry per plane in fragment order; chunks double buffered.
//
// Arithmetic.  The contraction runs over the rows, so the block scale of an operand must be uniform over the 16 rows of a chunk:
// every chunk of D and of B is scaled by an exact power of two that puts its largest magnitude into [2^14, 2^15) -- the top of
// fp16's range -- and split as  xs = h + l,  h = fp16(xs),  l = fp16(xs - h)  (the TRUE remainder: at this scale it stays a normal
// fp16 number for every element within 2^-2 of the chunk maximum and has an absolute error <= 2^-25 below that, i.e. the pair
// represents xs to 2^-39 of the chunk maximum or better).  hD hB + hD lB + lD hB go into ONE fp32 accumulator (dropped:
// lD lB <= 2^-24 of the product).  Chunks have different scales; the accumulator keeps a running unit 2^(E-30), E = the largest
// eD + eB seen so far: a chunk whose eD + eB is smaller by s is shifted down by s binary places, split between the two operands
// (each absorbs 15 places without leaving fp32 grade, 29 before its maximum stops being representable; beyond s = 58 the chunk
// is below 2^-58 of the largest one and contributes nothing to an fp32 sum); a chunk that raises E rescales the accumulators
// (exact, rare: the running maximum settles after the first chunks).  Chunk maxima are exchanged through LDS one pipeline stage
// ahead, so the exchange adds no barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nero_hip.h"
#include "common.h"
#include "mlp_split.h"

namespace {

#ifdef DW_PHASE_TIMING
__device__ unsigned long long g_dw_phase[8];
#define DPH_DECL long long ph_t = clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DPH(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); ph_acc[k] += t_ - ph_t; ph_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define DPH_END do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_dw_phase[k_], (unsigned long long)ph_acc[k_]); } while (0)
#else
#define DPH_DECL
#define DPH(k)
#define DPH_END
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DWH_PLANE = 2 * 256 * 16;              // one plane of one matrix: [2][256] x 16 B = 8 KB
constexpr int DWH_MAT = 2 * DWH_PLANE;               // h + l = 16 KB
constexpr int DWH_STAGE = 2 * DWH_MAT;               // D + B = 32 KB
constexpr int DWH_SMAX = 2 * DWH_STAGE;              // float smax[2 slots][8 waves]
constexpr int DWH_LDS = DWH_SMAX + 2 * 8 * 4;

__device__ __forceinline__ int exp_of(float m) {     // e with m * 2^-e in [0.5, 1) (0 for m == 0 / denormal), clamped
    const int eb = (__float_as_uint(m) >> 23) & 0xff;
    int e = eb ? eb - 126 : -60;
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}
__device__ __forceinline__ float p2(int e) {         // 2^e, e in [-126, 127]
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split_true(float a, float b, unsigned& h, unsigned& l) {
    h = pk_f16(a, b);
    const f16x2 hh = __builtin_bit_cast(f16x2, h);
    l = pk_f16(a - (float)hh[0], b - (float)hh[1]);
}

// ---- staging --------------------------------------------------------------------------------------------------------------
// Waves 0-3 stage D, waves 4-7 stage B.  Inside a group thread (cq = tid & 63, rq = wave & 3) owns columns 4cq..4cq+3 of the
// chunk rows 4rq..4rq+3: FOUR buffer_load_dwordx4 per chunk (a wave instruction = one full 1 KB row; the dword-per-lane form
// needs 4x the wave instructions and the texture-address path, ~38 cycles of issue per instruction beside 8 waves, was the
// pole of the first version).  The buffer resource is rebased to the chunk's first row (SALU); rows past the slice end and
// column quads past the matrix width fall outside num_records and read as 0; a ragged last quad is masked by multiplies.
#ifndef DW_LOAD_AUX
#define DW_LOAD_AUX 2                                // nt: both operand streams (0.6 GB per job) are read once; -4.5 % on the dW kernels
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dwh_fetch(f32x4 (&v)[4], const float* __restrict__ src, int ld, int r0, int r1, int off) {
    const int left = r1 - r0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)r0 * ld), 0, (left > 0 ? left : 0) * ld * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * ld * 4, 0, DW_LOAD_AUX));
}
// maximum of a non-negative value over the wave on the DPP network (no LDS traffic): xor 1, xor 2, half-row mirror, row mirror,
// then lane 15 -> next row, lane 31 -> upper half; lane 63 holds the result
__device__ __forceinline__ float wave_max(float m) {
#define NERO_DPPMAX(CTRL, ROWS) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), CTRL, ROWS, 0xf, true)))
    NERO_DPPMAX(0xB1, 0xf);
    NERO_DPPMAX(0x4E, 0xf);
    NERO_DPPMAX(0x141, 0xf);
    NERO_DPPMAX(0x140, 0xf);
    NERO_DPPMAX(0x142, 0xa);
    NERO_DPPMAX(0x143, 0xc);
#undef NERO_DPPMAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
}
// LDS image of one plane: [rh = row octet][entry e(col)] x 16 B, e(col) = 64 (col & 3) + ((col >> 2) + 4 (col & 3)) mod 64.
// The four columns of a thread land 64 entries apart, so that for each of them the 64 lanes of a wave write CONSECUTIVE entries
// (ds_write_b64: 2-way at worst), and the rotation by 4 (col & 3) keeps the fragment reads (lane i <-> column 32 t + i,
// ds_read_b128, lane groups {0-3,12-15,20-27} ...) on 16 distinct 16-byte slots.
__device__ __forceinline__ int dwh_entry(int col) { return 64 * (col & 3) + (((col >> 2) + 4 * (col & 3)) & 63); }

struct Fr2 { uint4 h, l; };
__device__ __forceinline__ Fr2 dwh_frag(const char* mat, int byte_off) {
    Fr2 f;
    f.h = *reinterpret_cast<const uint4*>(mat + byte_off);
    f.l = *reinterpret_cast<const uint4*>(mat + byte_off + DWH_PLANE);
    return f;
}
#define NERO_MFH(ACC, A, B) \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)
__device__ __forceinline__ void mf3(f32x16& acc, const Fr2& a, const Fr2& b) {       // (back-to-back accumulation forwards at full rate)
    NERO_MFH(acc, a.l, b.h);
    NERO_MFH(acc, a.h, b.l);
    NERO_MFH(acc, a.h, b.h);
}

// operand shifts of a chunk that lies s >= 0 binary places below the accumulator unit
__device__ __forceinline__ void split_shift(int s, int& a, int& b) {
    a = (s + 1) >> 1;
    a = a > 29 ? 29 : a;
    b = s - a;
    b = b > 29 ? 29 : b;
}

#ifndef DW_SGB_V
#define DW_SGB_V 4
#endif
// (Round 4, measured and dropped: staging through LDS-DMA.  global_load_lds_dwordx4 into a raw fp32 stage of 2 x 32 KB beside the planes, a
// chunk requested TWO iterations before its maxima are taken -- no registers while in flight, 233 instead of 248 VGPRs -- on the theory that
// the loop waits for loads it requested only one iteration earlier.  It does not: bit-identical results, 0.177 -> 0.194 ms per 298 k-row
// job, 7.0 -> 7.7 ms of weight gradients per step.  Phase clocks of wave 0 per chunk (scripts/bench_dw.py on -DDW_PHASE_TIMING builds):
// buffer loads  fetch 367, fragments 156, convert 797, MFMA 744, wait + publish 1033, barrier 911 = 4.0 k cycles;  LDS-DMA  248 / 180 /
// 808 / 744 / 1549 / 831 = 4.4 k -- the read-out of the raw stage and four DMA requests with their 64-bit addresses cost more issue
// slots than the wait they remove.  What the loop pays is ISSUE: ~200 VALU instructions per wave and chunk (800 cycles) beside 768 cycles
// of its own MFMAs, two waves per SIMD, and the two do not co-issue while both waves run the same phase between two barriers.
// Also tried: PHASE-SHIFTING the two waves of a SIMD (the D-stagers multiply chunk q while the B-stagers convert chunk q + 1, a barrier,
// roles swapped) so that one feeds the matrix pipe while the other feeds the VALU.  Not measurable: with the product block behind a
// wave-uniform branch hipcc spills the accumulators (158 VGPRs with two copies of the block, 286 with one copy in a half-step loop).)
// (Round 3, measured and dropped: TWO resident workgroups per CU for the narrow jobs -- 512 slices, __launch_bounds__(512, 4): 128 VGPRs with
// 12 spilled -- to double the waves pulling the delta stream, which only the four D-staging waves of a workgroup do: every narrow job
// got slower, 0.109 -> 0.132 ms (27-column job), 0.165 -> 0.226 ms (the SDF's 39-column jobs).)
// The kernel body; `slice` = the row slice of this workgroup (blockIdx.x of both entry points below).
template <bool NARROW>
__device__ __forceinline__ void dw_f16_body(const nero_dw_job& job, int n_rows, int rows_per_slice, float* __restrict__ partials,
                                            int n_pad, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* smax = reinterpret_cast<float*>(smem + DWH_SMAX);         // [2 slots][8 waves]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const bool isB = wave >= 4;                          // staging role of this wave
    const int cq = lane, rq = wave & 3;
    const int r_begin = blockIdx.x * rows_per_slice;
    int r_end = r_begin + rows_per_slice;
    r_end = r_end < n_rows ? r_end : n_rows;
    const int n_tiles = n_pad >> 5, k_tiles = k_pad >> 5;
    constexpr int NA = NARROW ? 1 : 2;                   // n-tiles per wave
    const int nt0 = NARROW ? wave : 2 * (wave >> 1), kt0 = NARROW ? 0 : 4 * (wave & 1);
    f32x16 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                   // bias gradient partials of this thread's 4 columns (D group)
    const int nch = r_end > r_begin ? (r_end - r_begin + 15) / 16 : 0;
    const int total = nch * (job.d1 ? 2 : 1);
    constexpr int OOB = 0x40000000;                      // byte offset beyond any num_records: the load returns 0
    const int cols = isB ? job.k_cols : job.n_out;
    const int ld0 = isB ? job.ldb0 : job.ldd0, ld1 = isB ? job.ldb1 : job.ldd1;
    const float* src0 = isB ? job.b0 : job.d0;
    const float* src1 = isB ? job.b1 : job.d1;
    const int off0 = 4 * cq < cols ? (4 * rq * ld0 + 4 * cq) * 4 : OOB, off1 = 4 * cq < cols ? (4 * rq * ld1 + 4 * cq) * 4 : OOB;
    f32x4 cmask;                                         // ragged last quad: columns >= cols hold foreign data
#pragma unroll
    for (int c = 0; c < 4; ++c) cmask[c] = 4 * cq + c < cols ? 1.f : 0.f;
    // byte offsets inside a plane: this thread's four 8-byte store slots, and the fragment entries of this wave's tiles
    int st_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) st_off[c] = (rq >> 1) * 4096 + dwh_entry(4 * cq + c) * 16 + (rq & 1) * 8;
    int fa_off[NA], fb_off[4];
#pragma unroll
    for (int a = 0; a < NA; ++a) fa_off[a] = h * 4096 + dwh_entry(32 * (nt0 + a) + i) * 16;
#pragma unroll
    for (int b = 0; b < 4; ++b) fb_off[b] = h * 4096 + dwh_entry(32 * (kt0 + b) + i) * 16;
    auto fetch = [&](f32x4 (&v)[4], int q) {
        const bool second = q >= nch;
#ifdef DW_NOSTREAM                                       // (timing experiment: every chunk re-reads the slice's first rows, L2 hits)
        const int r0 = r_begin;
#else
        const int r0 = r_begin + (second ? q - nch : q) * 16;
#endif
        dwh_fetch(v, second ? src1 : src0, second ? ld1 : ld0, r0, r_end, second ? off1 : off0);
    };
    // mask the ragged quad, publish this wave's maximum of the chunk part held in v -> smax[slot][wave]
    auto publish = [&](f32x4 (&v)[4], int slot) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[j][c] *= cmask[c];   // (component-wise: no v_pk_mul_f32 -- see the note on packed fp32 in common.h)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j][0]), fabsf(v[j][1])), fmaxf(fabsf(v[j][2]), fabsf(v[j][3]))));
        }
        m = wave_max(m);
        if (lane == 0) smax[slot * 8 + wave] = m;
    };
    int E = -1000;                                       // accumulator unit = 2^(E - 30); -1000 = nothing accumulated yet
    float resc = 1.f;                                    // factor the accumulators take before the next chunk's products are added
    // convert the chunk part in v (maxima of the whole chunk in smax[slot]) into stage `st`; updates E / resc (selects only:
    // `live` = false makes the call a harmless store into a stage nobody reads any more)
    auto put = [&](const f32x4 (&v)[4], int slot, char* st, bool live) {
        const float4 ma = *reinterpret_cast<const float4*>(smax + slot * 8), mb = *reinterpret_cast<const float4*>(smax + slot * 8 + 4);
        const int eD = exp_of(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w))), eB = exp_of(fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
        const int ec = eD + eB;
        const bool raise = live && ec > E;
        const float r_up = E <= -1000 ? 1.f : p2(E - ec);
        resc = raise ? r_up : 1.f;
        E = raise ? ec : E;
        const int s = E - ec;                            // >= 0 for live chunks
        int sa, sb;
        split_shift(s < 0 ? 0 : s, sa, sb);
        const float dead = s > 58 ? 0.f : 1.f;           // (below 2^-58 of the running maximum: nothing reaches an fp32 sum)
        const float sc = isB ? p2(15 - eB - sb) : p2(15 - eD - sa) * dead;
        char* mat = st + (isB ? DWH_MAT : 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned h01, l01, h23, l23;
            split_true(v[0][c] * sc, v[1][c] * sc, h01, l01);
            split_true(v[2][c] * sc, v[3][c] * sc, h23, l23);
            *reinterpret_cast<uint2*>(mat + st_off[c]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(mat + DWH_PLANE + st_off[c]) = make_uint2(l01, l23);
        }
    };
    f32x4 av[4], cv[4];                                  // chunk q+1 (landed) and chunk q+2 (in flight)
    float resc_next = 1.f;
    if (total > 0) {
        fetch(av, 0);
        publish(av, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) bsum[c] += av[j][c];
        if (total > 1) fetch(cv, 1);
    }
    __syncthreads();                                     // maxima of chunk 0 visible
    if (total > 0) {
        put(av, 0, smem, true);
        if (total > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = cv[j];
            publish(av, 1);
        }
        resc = 1.f;                                      // (chunk 0 defines the first unit: the accumulators are still zero)
    }
    __syncthreads();                                     // stage 0 and the maxima of chunk 1 visible
    DPH_DECL;
    for (int q = 0; q < total; ++q) {
        const char* sD = smem + (q & 1) * DWH_STAGE;
        const char* sB = sD + DWH_MAT;
        // (indices past the end are clamped: the extra fetch / store is harmless -- nobody reads that stage any more)
        fetch(cv, q + 2 < total ? q + 2 : total - 1);
        __builtin_amdgcn_sched_barrier(0);               // (the loads of chunk q+2 stay at the top: they are consumed at the bottom;
                                                         //  issuing them one step earlier, before the barrier, measured 5-10 % slower)
        DPH(0);
        // Program order inside the ONE basic block of the loop body: operand fragments of chunk q first (LDS reads of the current
        // stage -- issued before the stores into the other stage, which the compiler cannot prove disjoint), then the conversion
        // of chunk q+1 (VALU + LDS stores), then the products; the schedule-group pattern spreads the MFMAs over the VALU stream.
        auto convert_next = [&]() {
            const float keep = (q + 1 < nch) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) bsum[c] = fmaf(av[j][c], keep, bsum[c]);
            // chunk q+1 -> the other stage, in the unit it leaves E at
            put(av, (q + 1) & 1, smem + ((q + 1) & 1) * DWH_STAGE, q + 1 < total);
            resc_next = resc;
        };
        if (NARROW) {
            convert_next();
            if (nt0 < n_tiles) {
                const Fr2 fa = dwh_frag(sD, fa_off[0]);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b < k_tiles) mf3(acc[0][b], fa, dwh_frag(sB, fb_off[b]));
            }
        } else {
            Fr2 fa[NA], fb[4];
#pragma unroll
            for (int a = 0; a < NA; ++a) fa[a] = dwh_frag(sD, fa_off[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = dwh_frag(sB, fb_off[b]);
            DPH(1);
            convert_next();
            DPH(2);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < NA; ++a) mf3(acc[a][b], fa[a], fb[b]);
            DPH(3);
#if DW_SGB_V > 0 && !defined(DW_PHASE_TIMING)
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, DW_SGB_V, 0);   // DW_SGB_V VALU
            }
#endif
        }
        // the products of chunk q are in; bring the accumulators to the unit of chunk q+1 (rare)
        if (resc_next != 1.f) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[a][b][v] *= resc_next;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = cv[j];
        publish(av, q & 1);                              // chunk q+2's maxima -> the slot chunk q's just left
        DPH(4);
        __syncthreads();
        DPH(5);
    }
    DPH_END;
    // this slice's partial C (row-major [n_pad][k_pad]) and bias partial, in the layout dw_reduce_kernel expects
    const float unit = E <= -1000 ? 0.f : p2(E - 30);
    float* __restrict__ P = partials + (size_t)blockIdx.x * ((size_t)n_pad * k_pad + n_pad);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int nt = nt0 + a, kt = kt0 + b;
            if (nt < n_tiles && kt < k_tiles) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = 32 * nt + (v & 3) + 8 * (v >> 2) + 4 * h;
                    P[(size_t)row * k_pad + 32 * kt + i] = acc[a][b][v] * unit;
                }
            }
        }
    // bias: the four row quads of a column quad live in waves 0..3
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (!isB) red[rq * 64 + cq] = bsum;
    __syncthreads();
    if (tid < n_pad) {
        const float* r = reinterpret_cast<const float*>(smem);
        P[(size_t)n_pad * k_pad + tid] = (r[tid] + r[256 + tid]) + (r[512 + tid] + r[768 + tid]);
    }
}

template <bool NARROW>
__global__ __launch_bounds__(512, 1) void dw_f16_kernel(nero_dw_job job, int n_rows, int rows_per_slice, float* __restrict__ partials,
                                                        int n_pad, int k_pad) {
    dw_f16_body<NARROW>(job, n_rows, rows_per_slice, partials, n_pad, k_pad);
}
// Several jobs over the same rows in ONE launch: blockIdx.y = job.  For the launches of a few ten thousand rows (the reference's own
// train_ray_num = 512, the 2 P rows of the Stage-II material MLPs) a job alone neither fills the chip nor amortises its 256 partial
// matrices; batched, the host picks ~1024 slices over ALL jobs of a chain (nero_dw_gemm_batch).
template <bool NARROW>
__global__ __launch_bounds__(512, 1) void dw_f16_batch_kernel(nero_dw_batch B, int n_rows, int rows_per_slice, float* __restrict__ partials) {
    const int q = blockIdx.y;
    dw_f16_body<NARROW>(B.j[q], n_rows, rows_per_slice, partials + B.poff[q], B.n_pad[q], B.k_pad[q]);
}

}  // namespace

#ifdef DW_PHASE_TIMING
extern "C" int nero_debug_phases_dw(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dw_phase), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dw_phase), z, sizeof(z)); }
    return 0;
}
#endif
int nero_f16_dw(const nero_dw_job* job, int n_rows, int rows_per_slice, int slices, float* partials, int n_pad, int k_pad, hipStream_t stream) {
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_f16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DWH_LDS));
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_f16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DWH_LDS));
    if (k_pad <= 128)
        hipLaunchKernelGGL(dw_f16_kernel<true>, dim3(slices), dim3(512), DWH_LDS, stream, *job, n_rows, rows_per_slice, partials, n_pad, k_pad);
    else
        hipLaunchKernelGGL(dw_f16_kernel<false>, dim3(slices), dim3(512), DWH_LDS, stream, *job, n_rows, rows_per_slice, partials, n_pad, k_pad);
    return NERO_OK;
}
int nero_f16_dw_batch(const nero_dw_batch* B, int n_jobs, int narrow, int n_rows, int rows_per_slice, int slices, float* partials, hipStream_t stream) {
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_f16_batch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DWH_LDS));
    NERO_ONCE(hipFuncSetAttribute((const void*)dw_f16_batch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DWH_LDS));
    if (narrow)
        hipLaunchKernelGGL(dw_f16_batch_kernel<true>, dim3(slices, n_jobs), dim3(512), DWH_LDS, stream, *B, n_rows, rows_per_slice, partials);
    else
        hipLaunchKernelGGL(dw_f16_batch_kernel<false>, dim3(slices, n_jobs), dim3(512), DWH_LDS, stream, *B, n_rows, rows_per_slice, partials);
    return NERO_OK;
}
