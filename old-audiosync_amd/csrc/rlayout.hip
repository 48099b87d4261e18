// csrc/rlayout.hip -- the "real-column" decomposition of the cross-correlation path (production lengths).
//
// Same plan (F = 2N real samples, M = F/2 = M1*M2, the same twiddle tables and LDS engine) as xcorr_kernels.hip,
// different assignment of the real-input symmetry:
//
//   xcorr_kernels.hip  packs ADJACENT SAMPLES as complex (z[j] = x[2j] + i x[2j+1]); the k <-> M-k partners of the
//                      real-FFT untangling then sit in two different spectrum rows, so k_rows holds four rows per
//                      block and spends a phase (and its LDS passes) on the combine.
//   here               the real sequence is the matrix x[j1][j2], j = j1*M2 + j2, with 2*M1 rows of M2 samples.  The
//                      column transforms (over j1, length 2*M1, REAL input) are r2c transforms done inside the tile:
//                      rows 2m and 2m+1 are packed as one complex row, an M1-point complex transform runs in LDS and
//                      the untangling k1 <-> M1-k1 happens between slots of the SAME tile (k_fwd_cols_r).  What
//                      reaches HBM is C[k1][j2] for k1 = 0 .. M1 (M1+1 rows; the other half is its mirror image).
//                      Every row k1 is then an independent complex problem:
//                          X[k1 + 2*M1*k2] = DFT_M2( C[k1][.] * w_F^(k1 j2) )[k2]
//                      so k_rows_r takes ONE row of each spectrum per block (19.2 KB of LDS instead of 38.4: eight
//                      blocks of two waves per CU instead of four of four), multiplies X conj(Y)
//                      (src/cross_correlation.c:232-233) in the registers of the last forward stage -- no combine
//                      phase, no partner rows, no index tables -- and runs the inverse row transform on the single
//                      product row.  k_inv_cols_r undoes the column step: c2r of length 2*M1 from the M1+1 rows, with
//                      the tangling inside the tile, and searches the peak as k_inv_cols does.
//
// Algebra (checked in tests/model_fourstep.py::rlayout_*): with j = j1*M2 + j2 and k = k1 + 2*M1*k2,
//   forward  C[k1][j2]  = sum_j1 x[j1][j2] w_{2M1}^(j1 k1)                         (k_fwd_cols_r, k1 = 0..M1)
//            X[k]       = sum_j2 C[k1][j2] w_F^(k1 j2) w_M2^(j2 k2)                (k_rows_r)
//   inverse  Q[k1][j2]  = conj(w_F^(k1 j2)) sum_k2 P[k1 + 2M1 k2] conj(w_M2^(j2 k2)) (k_rows_r)
//            r[j1][j2]  = sum_{k1 < 2M1} Q[k1][j2] conj(w_{2M1}^(j1 k1)),  Q[2M1-k1] = conj Q[k1]   (k_inv_cols_r)
// r is F times the plain sum of products, like the other path (asx_api.hip: bound_scale).
//
// Replaces FFTW's r2c / c2r work of src/cross_correlation.c:34-39,237-239 and the scan of :52-67, :242.
#include "asx_internal.h"
#include "lds_fft.h"
#include "xcorr_dev.h"

#include <initializer_list>
#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) float2 asx_lds_r[];

// Diagnostic phase clocks (never in a shipped build; -DASX_STAMPS, tools/dbg/stamps_r.py): lane 0 of a block records the
// clock at phase boundaries into a buffer nothing else reads (kernel 0 = rows, 1 = forward columns, 2 = inverse columns).
#ifdef ASX_STAMPS
#define RSTAMP(kernel, block, slot)                                                                  \
    do {                                                                                             \
        if (P.stamps && P.stamp_kernel == (kernel) && threadIdx.x == 0)                              \
            P.stamps[(size_t)(block) * 8 + (slot)] = clock64();                                      \
    } while (0)
#else
#define RSTAMP(kernel, block, slot) do {} while (0)
#endif

// What these kernels need from the plan, BY VALUE in the kernel arguments.  Read through the plan pointer (as the
// packed-sample kernels do) a block starts with a chain of dependent latencies -- plan struct (scalar loads) -> table
// pointers -> twiddle lookups, and in the inverse column kernel a bound load + branch before the tile loads were even
// issued: 4-5 k cycles, a fifth of a block's life (tools/dbg/stamps_r.py, round 4).  No arrays in here: nothing the
// compiler would index dynamically and send to scratch.
struct RArgs {
    const float2 *tw1, *tw2, *tw_lo, *tw_hi;
    uint32_t N, nout;
    int M1, M2, ntiles;
    float bound_scale;
    unsigned long long *stamps;
    int stamp_kernel;
};
static RArgs rargs_of(const AsxDev &P)
{
    RArgs a;
    a.tw1 = P.tw1; a.tw2 = P.tw2; a.tw_lo = P.tw_lo; a.tw_hi = P.tw_hi;
    a.N = P.N; a.nout = P.nout; a.M1 = P.M1; a.M2 = P.M2; a.ntiles = P.ntiles; a.bound_scale = P.bound_scale;
    a.stamps = P.stamps; a.stamp_kernel = P.stamp_kernel;
    return a;
}
// w_F^p from the two-level table (xcorr_dev.h's tw_F on the argument block)
__device__ __forceinline__ float2 tw_F(const RArgs &P, uint32_t p)
{
    const float2 lo = P.tw_lo[p & (ASX_TW_LO - 1u)];
    const float2 hi = P.tw_hi[p >> ASX_TW_LOG];
    return cmul(lo, hi);
}

// single-member values live in the first 8 bytes of a 16-byte slot: the wave-local stages rewrite exactly the
// slots they read, so no other wave's data is ever touched
__device__ __forceinline__ Cx1 lds_get1(const float4 *p)
{
    const float2 x = *reinterpret_cast<const float2 *>(p);
    return Cx1{ x.x, x.y };
}
__device__ __forceinline__ void lds_put1(float4 *p, Cx1 v) { *reinterpret_cast<float2 *>(p) = make_float2(v.re, v.im); }

__device__ __forceinline__ void wave_lds_sync()
{
    // same wave: the next stage reads what this one wrote (LDS executes a wave's operations in order; the
    // explicit wait does not lean on that)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Diagnostic builds only (tools/mkr.sh; never defined in the shipped library):
//   -DASX_EXP_PAIRMOD=<m>   every pair of a launch uses the C / Q workspaces of pair (pair % m): wrong results, the traffic
//                           of a launch whose intermediates never leave the 256 MiB Infinity Cache (upper bound of keeping
//                           them on die, EXPERIMENTS.md)
//   -DASX_ROWS2_SCHED=a,b,c the radix schedule of the 1200-point sub-rows of the two-half row kernel
// sum over the four lanes of a quad (every lane gets it): two DPP steps
__device__ __forceinline__ float quad_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
    return v;
}
// p[0] + ... + p[N-1] as a balanced tree: ceil(log2 N) roundings on every path
template <int N> __device__ __forceinline__ float tree_sum(const float (&p)[N])
{
    if constexpr (N == 1) return p[0];
    else {
        float q[(N + 1) / 2];
        static_for<0, N / 2>([&](auto I) __attribute__((always_inline)) { q[I] = p[2 * decltype(I)::value] + p[2 * decltype(I)::value + 1]; });
        if constexpr (N & 1) q[N / 2] = p[N - 1];
        return tree_sum<(N + 1) / 2>(q);
    }
}
// row pairs a lane group of k_fwd_cols_r<Sched<m1, ...>, t, nt> loads = half the rows of a band (asx_rlayout_band_rows)
__host__ __device__ constexpr int rcol_rows_per_group(int m1, int nt, int t) { return (m1 + nt / (t / 4) - 1) / (nt / (t / 4)); }

#ifdef ASX_EXP_PAIRMOD
#define RWS_PAIR(pair) ((pair) % (ASX_EXP_PAIRMOD))
#else
#define RWS_PAIR(pair) (pair)
#endif
// Non-temporal accesses for everything these kernels touch exactly once between its producer and its consumer: C written by
// k_fwd_cols_r (2) and read by k_rows_r (4), Q written by k_rows_r (8) and read by k_inv_cols_r (1).  All four together,
// same box, six alternating rounds: forward columns 1.019 -> 0.992 ms, rows 0.944 -> 0.926, inverse columns 0.325 -> 0.306,
// 51.6 -> 53.1 k/s (profiles/r5_experiments/10_*; the row LOADS alone slow the inverse kernel by 10 %: all or none).  The inputs stay
// temporal: the two tiles of a 128-byte input line meet in one XCD's L2 (rcol_tile_of_block); non-temporal, 0.987 -> 1.000 ms.
#ifndef ASX_RNT
#define ASX_RNT 15
#endif
typedef float asx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_f4(const void *p, bool nt)
{
    if (nt) { const asx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const asx_f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void st_f4(void *p, float4 v, bool nt)
{
    if (nt) { asx_f4v w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<asx_f4v *>(p)); }
    else *reinterpret_cast<float4 *>(p) = v;
}
typedef float asx_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_f2(float2 *p, float2 v, bool nt)
{
    if (nt) { asx_f2v w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<asx_f2v *>(p)); }
    else *p = v;
}
#ifndef ASX_ROWS2_SCHED
#define ASX_ROWS2_SCHED 12, 10, 10
#endif

// k_rows_r, which table look-ups are issued ahead of the loads they would otherwise queue behind (vmcnt completes in order):
// bit 0 = the block's tw_step / leg entries BEFORE the row loads (the barrier in front of the load phase's arithmetic then waits for an
// L2 round trip that overlaps the HBM loads instead of following them); bit 1 = the twiddles of the two-half form's store phase at
// the top of the kernel instead of behind its last barrier
// Measured (profiles/r5_experiments/19_*, alternating rounds): bit 1 rows 0.931 -> 0.918 ms (the store phase started with an L2 round
// trip on the two waves that run it); bit 0 nothing (0.930); both 0.921.  Bit 2 (the load steps' w_M2 values in front of the rows, whole
// steps without a condition): 0.946 -> 0.912 ms on top of bit 1, 0.939 -> 0.904 at N = 960 000.  Bit 3 (EVERY table look-up of the thread in
// front of its row loads): 480-point rows 0.771 -> 0.742 ms per 1024 pairs of N = 144 000, 0.928 -> 0.899 at 288 000; two-half form 0.917 -> 0.911.
// Default: bits 1, 2 and 3.
#ifndef ASX_ROWS_EARLY
#define ASX_ROWS_EARLY 14
#endif
#ifndef ASX_ROWSR_WAVES
#define ASX_ROWSR_WAVES 4 // waves per SIMD the register allocation must allow: 8 blocks of 2 waves per CU
#endif

// ---------------------------------------------------------------------------
// k_rows_r: grid ((M1 + 1) * npairs).  One block = row k1 of both spectra of one pair.
//   S = Sched<n, R0, R1, R2>: three DIF stages.  Stage 0 spans the (sub-)row (block-wide, barrier after it); stages 1
//   and 2 stay inside the R0 sub-blocks of length n/R0 ("units"), and so do the first two stages of the inverse:
//   a wave that owns whole units runs  forward 1 -> forward 2 -> product -> inverse 2 -> inverse 1  on them with no
//   block barrier in between.  Inverse stage 0 spans the (sub-)row again.
//   TWO = false: M2 = n, NT threads; the outputs of inverse stage 0 leave from registers.
//   TWO = true:  M2 = 2n, 2*NT threads.  The first radix-2 DIF stage of the 2n-point row transforms runs in the
//                registers of the load phase -- a[j] = c[j] + c[j+n], b[j] = (c[j] - c[j+n]) w_M2^j -- after which the
//                even bins (from a) and the odd bins (from b) are two independent n-point problems: threads
//                [0, NT) take a, threads [NT, 2NT) take b, each half in its own LDS region; they meet again in the
//                store phase, Q[j] = A[j] + conj(w_M2^j) B[j], Q[j+n] = A[j] - conj(w_M2^j) B[j].  This is how the two
//                longest reference lengths get 600- and 400-row column tiles of sixteen real columns (64-byte input
//                pieces, whole 128-byte lines of C and Q) instead of 1200 / 800 rows of eight.
// ---------------------------------------------------------------------------
template <class S, int NT, bool TWO>
__global__ __launch_bounds__(TWO ? 2 * NT : NT, ASX_ROWSR_WAVES) void k_rows_r(const RArgs P, const float2 *__restrict__ cx,
                                                                                const float2 *__restrict__ cy, float2 *__restrict__ qo,
                                                                                int nrows, size_t pair_pitch, AsxPeakWs W)
{
    static_assert(S::nstages == 3, "three-stage row schedules only");
    constexpr int NS = S::n;                 // length of a (sub-)row transform
    constexpr int M2 = TWO ? 2 * NS : NS;    // row length
    constexpr int TWS = TWO ? 2 : 1;         // w_NS^q = tw2[TWS * q] (tw2 is the table of M2)
    constexpr int NTB = TWO ? 2 * NT : NT;   // block size
    constexpr StageK K0 = S::stage(0), K1 = S::stage(1), K2 = S::stage(2);
    constexpr int R0 = K0.R, R1 = K1.R, R2 = K2.R;
    constexpr int Q0 = K0.q;  // butterflies of stage 0 = length of a unit
    constexpr int UN = K1.ns; // = Q0
    static_assert(UN == R1 * R2 && Q0 == UN && K2.q == 1, "unit = R1 x R2");
    static_assert((NS & 1) == 0, "rows move as 16 bytes per lane");
    constexpr int HALF = NS / 2, WSTEPS = (HALF + NTB - 1) / NTB;
    constexpr int LPU = R1 > R2 ? R1 : R2, UPW = 64 / LPU, NW = NT / 64;
    static_assert(NT % 64 == 0 && UPW >= 1, "whole waves");

#ifdef ASX_STAMPS
    const long long t_entry = clock64(), w_entry = wall_clock64();
#endif
    const int task = blockIdx.x, tid = threadIdx.x;
    const int half = TWO ? tid / NT : 0, lt = tid - half * NT; // which sub-row, thread index inside its half
    float4 *A4 = reinterpret_cast<float4 *>(asx_lds_r) + half * NS;
    __shared__ float2 tw_step[WSTEPS]; // (1/2) w_F^(k1 * 2*NTB*i): the four-step twiddle from load step to load step
    __shared__ float2 leg[R0];         // w_F^(k1 * Q0*t): ... and from leg to leg of the last inverse stage

    const int pair = task / nrows;
    const uint32_t k1 = (uint32_t)(task - pair * nrows);
    const size_t row = (size_t)RWS_PAIR(pair) * pair_pitch + (size_t)k1 * M2;
    if (k1 == 0 && tid < 64) {
        // Row 0 of a pair also prepares the pair's peak search (k_inv_cols_r runs after this kernel): the float32
        // error bound from the norms k_fwd_cols_r left, the running maximum and the candidate count back to zero.
        const float *np = W.nrm_part + (size_t)pair * 2 * P.ntiles;
        float sx = 0.f, sy = 0.f;
        for (int t = tid; t < P.ntiles; t += 64) { sx += np[t]; sy += np[P.ntiles + t]; }
        sx = wave_sum_f32(sx); sy = wave_sum_f32(sy);
        if (tid == 0) {
            W.bound2[pair] = P.bound_scale * sqrtf(sx) * sqrtf(sy);
            W.pairmax[pair] = 0;
            W.cand_n[pair] = 0;
        }
    }
    // bit 3 of ASX_ROWS_EARLY (see the row loads below): -3 ... -4 % for 480-point rows, -0.7 % for the two-half form, +0.6 % for 1200-point
    // rows in one piece, which keep the old order (profiles/r5_experiments/19_*)
    constexpr bool LOOKUPS_FIRST = (ASX_ROWS_EARLY & 8) != 0 && (TWO || NS != 1200);
    const bool is_step = tid < WSTEPS, is_leg = tid >= NTB - R0;
    static_assert(WSTEPS <= NTB - R0, "the threads that fill tw_step and leg are different threads");
    float2 pre_lo = make_float2(1.f, 0.f), pre_hi = pre_lo;
    if ((LOOKUPS_FIRST || (ASX_ROWS_EARLY & 1)) && (is_step || is_leg)) {
        const uint32_t p = is_step ? k1 * (uint32_t)(2 * NTB * tid) : k1 * (uint32_t)(Q0 * (tid - (NTB - R0)));
        pre_lo = P.tw_lo[p & (ASX_TW_LO - 1u)];
        pre_hi = P.tw_hi[p >> ASX_TW_LOG];
    }
    // the look-ups of the two-half form's store phase (threads [0, Q0): butterfly j = tid)
    float2 tail_w2 = make_float2(1.f, 0.f), tail_lo = tail_w2, tail_hi = tail_w2;
    if (TWO && (ASX_ROWS_EARLY & 2) && tid < Q0) {
        const uint32_t p = k1 * (uint32_t)tid;
        tail_w2 = P.tw2[tid];
        tail_lo = P.tw_lo[p & (ASX_TW_LO - 1u)];
        tail_hi = P.tw_hi[p >> ASX_TW_LOG];
    }
    // (bit 2 of ASX_ROWS_EARLY: the two-half form's w_M2^(2q), w_M2^(2q+1) of every load step asked for HERE, in front of the rows, instead
    // of inside the step behind the barrier -- a table load and the wait for it between the arrival of the rows and their first use)
    float4 w2pre[TWO ? WSTEPS : 1];
    if constexpr (TWO && (ASX_ROWS_EARLY & 4) != 0) {
        static_for<0, WSTEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int q = tid + i * NTB;
            w2pre[I] = make_float4(1.f, 0.f, 1.f, 0.f);
            if ((i + 1) * NTB <= HALF || q < HALF) w2pre[I] = *reinterpret_cast<const float4 *>(P.tw2 + 2 * q);
        });
    }
    // The row loads.  Bit 3 of ASX_ROWS_EARLY: issued BEHIND every table look-up of the thread (they used to come first, "the look-ups
    // overlap them" -- but vmcnt completes in order: a look-up issued behind the rows is not usable before the rows have arrived, and the
    // tw_step / leg threads then started an L2 round trip of their own in front of the block's barrier)
    float4 lx[WSTEPS], ly[WSTEPS], lx2[TWO ? WSTEPS : 1], ly2[TWO ? WSTEPS : 1];
    auto issue_row_loads = [&]() __attribute__((always_inline)) {
        const float2 *gx = cx + row, *gy = cy + row;
        static_for<0, WSTEPS>([&](auto I) __attribute__((always_inline)) {
            const int q = tid + decltype(I)::value * NTB;
            if (((ASX_ROWS_EARLY & 4) && (decltype(I)::value + 1) * NTB <= HALF) || q < HALF) { // (a whole step inside the row: no condition)
                lx[I] = ld_f4(gx + 2 * q, ASX_RNT & 4);
                ly[I] = ld_f4(gy + 2 * q, ASX_RNT & 4);
                if constexpr (TWO) {
                    lx2[I] = ld_f4(gx + NS + 2 * q, ASX_RNT & 4);
                    ly2[I] = ld_f4(gy + NS + 2 * q, ASX_RNT & 4);
                }
            }
        });
    };
    if constexpr (!LOOKUPS_FIRST) issue_row_loads();
    auto publish_step_and_leg = [&]() __attribute__((always_inline)) {
        if (is_step) {
            // ... times 1/2: k_fwd_cols_r stores 2 C (its untangling without the halving)
            const float2 t = (LOOKUPS_FIRST || (ASX_ROWS_EARLY & 1)) ? cmul(pre_lo, pre_hi) : tw_F(P, k1 * (uint32_t)(2 * NTB * tid));
            tw_step[tid] = make_float2(0.5f * t.x, 0.5f * t.y);
        }
        if (is_leg) { const int t = tid - (NTB - R0); leg[t] = (LOOKUPS_FIRST || (ASX_ROWS_EARLY & 1)) ? cmul(pre_lo, pre_hi) : tw_F(P, k1 * (uint32_t)(Q0 * t)); }
    };
    if constexpr (!LOOKUPS_FIRST) publish_step_and_leg();
    const float2 twa = tw_F(P, k1 * (uint32_t)(2 * tid < NS ? 2 * tid : 0)); // w_F^(k1 * 2 tid)
    const float2 wk1 = tw_F(P, k1);
    const float2 wh = TWO ? tw_F(P, k1 * (uint32_t)NS) : make_float2(1.f, 0.f); // w_F^(k1 n): from c[j] to c[j + n]
    // stage twiddle seeds (they depend on the thread only): stage 0 / inverse stage 0, and the wave-local stage 1
    const int j0 = lt < Q0 ? lt : 0;
    const float2 s0w1 = P.tw2[TWS * j0], s0w4 = P.tw2[TWS * 4 * j0];
    const float2 tw0base = TWO ? make_float2(1.f, 0.f) : tw_F(P, k1 * (uint32_t)j0); // four-step factor of the outputs of inverse stage 0 (!TWO)
    const int lane = tid & 63, wave = lt >> 6;
    const int ul = lane / LPU, jl = lane - ul * LPU;
    const int j1c = jl < R2 ? jl : 0;
    const float2 s1w1 = P.tw2[TWS * K1.twmul * j1c], s1w4 = P.tw2[TWS * (R1 > 4 ? 4 : 1) * K1.twmul * j1c];
    if constexpr (LOOKUPS_FIRST) { issue_row_loads(); publish_step_and_leg(); }
    RSTAMP(0, task, 0);
#ifdef ASX_STAMPS
    if (P.stamps && P.stamp_kernel == 0 && threadIdx.x == 0) { P.stamps[(size_t)task * 8 + 6] = t_entry; P.stamps[(size_t)task * 8 + 7] = w_entry; }
#endif
    __syncthreads(); // tw_step, leg
    static_for<0, WSTEPS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int q = tid + i * NTB;
        if (((ASX_ROWS_EARLY & 4) && (i + 1) * NTB <= HALF) || q < HALF) {
            const float2 wa0 = cmul(twa, tw_step[i]), wa1 = cmul(wa0, wk1);
            const Cx2 c0 = mulw(Cx2{ v2f{ lx[I].x, ly[I].x }, v2f{ lx[I].y, ly[I].y } }, wa0);
            const Cx2 c1 = mulw(Cx2{ v2f{ lx[I].z, ly[I].z }, v2f{ lx[I].w, ly[I].w } }, wa1);
            if constexpr (!TWO) {
                lds_put(A4 + 2 * q, c0);
                lds_put(A4 + 2 * q + 1, c1);
            } else {
                float4 *base = reinterpret_cast<float4 *>(asx_lds_r);
                const float2 wb0 = cmul(wa0, wh), wb1 = cmul(wa1, wh);
                const Cx2 d0 = mulw(Cx2{ v2f{ lx2[I].x, ly2[I].x }, v2f{ lx2[I].y, ly2[I].y } }, wb0);
                const Cx2 d1 = mulw(Cx2{ v2f{ lx2[I].z, ly2[I].z }, v2f{ lx2[I].w, ly2[I].w } }, wb1);
                const float4 w2 = (ASX_ROWS_EARLY & 4) ? w2pre[I] : *reinterpret_cast<const float4 *>(P.tw2 + 2 * q); // w_M2^(2q), w_M2^(2q+1)
                lds_put(base + 2 * q, c0 + d0);
                lds_put(base + 2 * q + 1, c1 + d1);
                lds_put(base + NS + 2 * q, mulw(c0 - d0, make_float2(w2.x, w2.y)));
                lds_put(base + NS + 2 * q + 1, mulw(c1 - d1, make_float2(w2.z, w2.w)));
            }
        }
    });
    __syncthreads();
    RSTAMP(0, task, 1);

    // ---- forward stage 0: butterflies j < Q0, legs Q0 apart ------------------------------------------------
    for (int j = lt; j < Q0; j += NT) {
        float4 *p = A4 + j;
        Cx2 v[R0];
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + decltype(T)::value * Q0); });
        float2 w1 = s0w1, w4 = s0w4;
        if (j != lt) { w1 = P.tw2[TWS * j]; w4 = P.tw2[TWS * 4 * j]; }
        float2 tww[R0];
        stage_twiddles_from<R0>(w1, w4, tww);
        Bfly<R0, false>::run(v);
        static_for<1, R0>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tww[U]); });
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) { lds_put(p + decltype(T)::value * Q0, v[T]); });
    }
    __syncthreads();
    RSTAMP(0, task, 2);

    // ---- wave-local: forward 1, forward 2 + X conj(Y) + inverse 2, inverse 1 -------------------------------
    for (int u0 = wave * UPW; u0 < R0; u0 += NW * UPW) { // wave-uniform trip count
        const int unit = u0 + ul;
        const bool on = (ul < UPW) && (unit < R0);
        float4 *base = A4 + (on ? unit : 0) * UN;
        float2 tw1[R1];
        stage_twiddles_from<R1>(s1w1, s1w4, tw1);
        if (on && jl < R2) { // stage 1: butterfly jl of the unit, legs R2 apart, outputs times w_UN^(jl u)
            float4 *p = base + jl;
            Cx2 v[R1];
            static_for<0, R1>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + decltype(T)::value * R2); });
            Bfly<R1, false>::run(v);
            static_for<1, R1>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tw1[U]); });
            static_for<0, R1>([&](auto T) __attribute__((always_inline)) { lds_put(p + decltype(T)::value * R2, v[T]); });
        }
        wave_lds_sync();
        if (on && jl < R1) { // stage 2: R2 consecutive slots; the bins of X and Y meet in registers
            float4 *p = base + jl * R2;
            Cx2 v[R2];
            static_for<0, R2>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + decltype(T)::value); });
            Bfly<R2, false>::run(v);
            Cx1 pr[R2];
            static_for<0, R2>([&](auto T) __attribute__((always_inline)) {
                // src/cross_correlation.c:232-233: X * conj(Y); member x = source, member y = sample
                pr[T] = Cx1{ v[T].re.x * v[T].re.y + v[T].im.x * v[T].im.y, v[T].im.x * v[T].re.y - v[T].re.x * v[T].im.y };
            });
            Bfly<R2, true>::run(pr);
            static_for<0, R2>([&](auto T) __attribute__((always_inline)) { lds_put1(p + decltype(T)::value, pr[T]); });
        }
        wave_lds_sync();
        if (on && jl < R2) { // inverse stage 1
            float4 *p = base + jl;
            Cx1 v[R1];
            static_for<0, R1>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get1(p + decltype(T)::value * R2); });
            static_for<1, R1>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tw1[U]); });
            Bfly<R1, true>::run(v);
            static_for<0, R1>([&](auto T) __attribute__((always_inline)) { lds_put1(p + decltype(T)::value * R2, v[T]); });
        }
    }
    __syncthreads();
    RSTAMP(0, task, 3);

    // ---- inverse stage 0 from LDS, the outputs leave from registers -------------------------------------------------
    float2 *go = qo + row;
    if constexpr (!TWO) {
        for (int j = lt; j < Q0; j += NT) {
            const float4 *p = A4 + j;
            Cx1 v[R0];
            static_for<0, R0>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get1(p + decltype(T)::value * Q0); });
            float2 w1 = s0w1, w4 = s0w4, fb = tw0base;
            if (j != lt) { w1 = P.tw2[j]; w4 = P.tw2[4 * j]; fb = tw_F(P, k1 * (uint32_t)j); }
            float2 tww[R0];
            stage_twiddles_from<R0>(w1, w4, tww);
            static_for<1, R0>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tww[U]); });
            Bfly<R0, true>::run(v);
            // conjugate four-step twiddle, straight to HBM
            static_for<0, R0>([&](auto T) __attribute__((always_inline)) {
                constexpr int t = decltype(T)::value;
                const Cx1 y = mulwc(v[t], t == 0 ? fb : cmul(fb, leg[t]));
                st_f2(go + j + t * Q0, make_float2(y.re, y.im), ASX_RNT & 8);
            });
        }
    } else {
        // Both halves at once, as ONE pair of sequences (A, B) per thread: the butterfly j of inverse stage 0 of the
        // even-bin and of the odd-bin problem share their twiddles, and their outputs A[j + Q0 t], B[j + Q0 t] are exactly
        // what the last radix-2 stage combines:  Q[i] = A[i] + conj(w_M2^i) B[i],  Q[i + n] = A[i] - conj(w_M2^i) B[i]
        // (w_M2^(j + Q0 t) = w_M2^j times the compile-time root w_{2 R0}^t), then the conjugate four-step twiddles.
        // Only Q0 threads work here; the others are done.
        const float4 *base = reinterpret_cast<const float4 *>(asx_lds_r);
        static_assert(Q0 <= NT, "one butterfly per thread of the first half");
        for (int j = tid; j < Q0; j += NTB) {
            Cx2 v[R0];
            static_for<0, R0>([&](auto T) __attribute__((always_inline)) {
                const Cx1 a = lds_get1(base + j + decltype(T)::value * Q0), b = lds_get1(base + NS + j + decltype(T)::value * Q0);
                v[T] = Cx2{ v2f{ a.re, b.re }, v2f{ a.im, b.im } };
            });
            float2 tww[R0];
            if constexpr (ASX_ROWS_EARLY & 2) stage_twiddles_from<R0>(s0w1, s0w4, tww); // j = tid = lt here: the seeds of forward stage 0
            else stage_twiddles_from<R0>(P.tw2[TWS * j], P.tw2[TWS * 4 * j], tww);
            static_for<1, R0>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tww[U]); });
            Bfly<R0, true>::run(v);
            const float2 w2 = (ASX_ROWS_EARLY & 2) ? tail_w2 : P.tw2[j];                                  // w_M2^j
            const float2 fa = (ASX_ROWS_EARLY & 2) ? cmul(tail_lo, tail_hi) : tw_F(P, k1 * (uint32_t)j);  // w_F^(k1 j)
            const float2 fbh = cmul(fa, wh);                  // w_F^(k1 (j + n))
            static_for<0, R0>([&](auto T) __attribute__((always_inline)) {
                constexpr int t = decltype(T)::value;
                const Cx1 A = Cx1{ v[t].re.x, v[t].im.x };
                const Cx1 Bw = mul_root<2 * R0, t, true>(mulwc(Cx1{ v[t].re.y, v[t].im.y }, w2)); // conj(w_M2^(j + Q0 t)) B
                const float2 lg = leg[t];
                const Cx1 y = mulwc(A + Bw, t == 0 ? fa : cmul(fa, lg));
                const Cx1 z = mulwc(A - Bw, t == 0 ? fbh : cmul(fbh, lg));
                st_f2(go + j + t * Q0, make_float2(y.re, y.im), ASX_RNT & 8);
                st_f2(go + NS + j + t * Q0, make_float2(z.re, z.im), ASX_RNT & 8);
            });
        }
    }
    RSTAMP(0, task, 4);
#ifdef ASX_STAMPS
    if (P.stamps && P.stamp_kernel == 0 && threadIdx.x == 0) P.stamps[(size_t)task * 8 + 5] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------
// Column tiles of the real-column kernels: T REAL columns j2 = c0 .. c0+T-1 of the [2 M1][M2] sample matrix, held
// in LDS as [M1][T/2] float4 slots: slot (m, g) = { x[2m][c0+2g], x[2m+1][c0+2g], x[2m][c0+2g+1], x[2m+1][c0+2g+1] },
// i.e. the complex sequences z[m] = x[2m] + i x[2m+1] of two adjacent columns side by side (the pair-planar form
// of lds_fft.h).  A tile row is 4T bytes of the input (32 / T tiles share a 128-byte line: blocks b, b+8, ... of
// one XCD, as in col_tile_of_block) and 8T bytes of C / Q.
// ---------------------------------------------------------------------------
__device__ __host__ constexpr int rcol_line_log(int logT) { return logT >= 5 ? 0 : 5 - logT; } // log2(tiles per input line)
__device__ __forceinline__ int rcol_tile_of_block(int b, int logT)
{
    const int ll = rcol_line_log(logT);
    const int grp = 8 << ll;
    return (b & ~(grp - 1)) + ((b & 7) << ll) + ((b >> 3) & ((1 << ll) - 1));
}
static inline unsigned rcol_grid_x(int ntiles, int logT)
{
    const unsigned grp = 8u << rcol_line_log(logT);
    return ((unsigned)ntiles + grp - 1) / grp * grp;
}

// The work items v = 0 of the column kernels' register stage (the butterflies that pair with themselves: H of (MB / 2) H items, a branch
// of their own) sit in the LAST H threads of the block when that leaves them a wave of their own (600- and 400-row tiles: 392 / 312
// regular items, 512 threads): with them in threads [0, H), wave 0 ran both branches one after the other -- in k_fwd_cols_r at the very
// end of the block's life, in k_inv_cols_r with a row load of its own behind the others.  Forward / inverse columns 0.989 -> 0.978 /
// 0.314 -> 0.310 ms at 600 rows, 2.766 -> 2.746 / 1.049 -> 1.019 at 400; 300-row tiles (256 threads, no idle wave): +1.5 % / 0, left alone
// (profiles/r5_experiments/21_*; -DASX_SPECIAL_LAST=0: the old order everywhere).
#ifndef ASX_SPECIAL_LAST
#define ASX_SPECIAL_LAST 1
#endif
// item of thread t: regular items e = H .. NITEMS-1 in threads [0, NITEMS - H), the H special ones in threads [NT - H, NT); -1 = none
template <int NITEMS, int H, int NT> __device__ __forceinline__ int rcol_item_of_thread(int t)
{
    constexpr bool own_wave = NITEMS - H <= NT - 64;
    if (!ASX_SPECIAL_LAST || !own_wave) return t < NITEMS ? t : -1;
    return t < NITEMS - H ? t + H : t >= NT - H ? t - (NT - H) : -1;
}
#ifndef ASX_INV_WU
#define ASX_INV_WU 1 // k_inv_cols_r: see WU_FRONT
#endif
#ifndef ASX_RCOL_LOADS
#define ASX_RCOL_LOADS 5 // row-pair pieces (two 16-byte loads each) a thread keeps in flight
#endif

// k_fwd_cols_r: grid (tiles, {source, sample}, npairs).  r2c column transforms of length 2 M1 (the zero half of the
// sample -- rows j1 >= M1 -- is never loaded, src/cross_correlation.c:159-166): M1-point complex transform of the
// packed rows, untangling between the slots of u and M1 - u, rows u and M1 - u of C (twice its value: the factor
// is taken back by k_rows_r) stored in natural row order.
template <class S1, int TC, int NT>
__global__ __launch_bounds__(NT, 4) void k_fwd_cols_r(const RArgs P, const float *__restrict__ src,
                                                       const float *__restrict__ smp, float2 *__restrict__ cx,
                                                       float2 *__restrict__ cy, float *__restrict__ nrm_part, size_t pair_pitch,
                                                       float2 *__restrict__ band)
{
    constexpr int M1 = S1::n, T = TC, logT = asx_ilog2(TC), H = T / 2, logH = logT - 1, Q4 = T / 4, logQ4 = logT - 2;
    static_assert(T >= 4 && (M1 & 1) == 0, "four real columns per 16-byte load, an even number of packed rows");
    __shared__ float nrm_red[NT / 64];
    const bool is_smp = blockIdx.y != 0;
    const size_t pair = blockIdx.z;
    const int tile = rcol_tile_of_block(blockIdx.x, logT);
    const unsigned which = blockIdx.y;
    if (tile >= P.ntiles) return; // grid.x is rounded up
    const int M2 = P.M2, c0 = tile * T;
    const float *in = is_smp ? smp + pair * (size_t)P.N : src + pair * (size_t)(2u * P.N);
    const int data_m = is_smp ? M1 / 2 : M1; // packed rows that are not zero padding
    float2 *out = (is_smp ? cy : cx) + RWS_PAIR(pair) * pair_pitch;
    float4 *lds4 = reinterpret_cast<float4 *>(asx_lds_r);
    const LdsLayout Lc = col_layout(T, logT, NT);
    const TwPre pre = tw_prefetch_first<S1, false, true>(Lc, P.tw1);

    // Work items (m, h): rows 2m and 2m+1, real columns c0 + 4h .. 4h+3 -> slots (m, 2h) and (m, 2h+1).  The Q4 lanes h of a lane
    // group take RPQ CONSECUTIVE row pairs m = RPQ * group + i (each a 64-byte piece of two rows: the addresses a wave asks for
    // are 16 pieces apart whichever way the pairs are dealt).  That makes a lane group the owner of one BAND of 2 RPQ rows x T
    // columns, whose sum and sum of squares -- what the spectral Pearson form (pearson_spectral.hip) builds its window sums
    // from -- are RPQ register additions and two DPP steps away from the loads this pass makes anyway.  (Dealing the pairs
    // round-robin, 16 lanes of a wave per band of eight rows: five 16-lane reductions per thread, k_fwd_cols_r +3 %,
    // profiles/r5_experiments/04_*.)
    constexpr int QN = NT / Q4, RPQ = rcol_rows_per_group(M1, NT, T);
    static_assert(Q4 == 4, "sixteen real columns: four lanes per row pair (the quad reductions below)");
    static_assert(M1 % RPQ == 0 && (M1 / 2) % RPQ == 0 && RPQ <= ASX_RCOL_LOADS && RPQ * QN >= M1, "whole bands in both tracks");
    const int grp = threadIdx.x >> logQ4, h = threadIdx.x & (Q4 - 1), m0 = grp * RPQ;
    float ss = 0.f, q1 = 0.f;
    {
        float4 a[RPQ], b[RPQ];
        float sq[RPQ]; // a row pair's eight squares: one chain of eight roundings; the pairs are then added as a tree (below)
        static_for<0, RPQ>([&](auto I) __attribute__((always_inline)) {
            const int m = m0 + decltype(I)::value;
            a[I] = b[I] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < data_m) { // also false past the end of the tile (m >= M1 >= data_m)
                const float *p = in + (size_t)(2 * m) * M2 + c0 + 4 * h;
                a[I] = *reinterpret_cast<const float4 *>(p);
                b[I] = *reinterpret_cast<const float4 *>(p + M2);
            }
        });
        static_for<0, RPQ>([&](auto I) __attribute__((always_inline)) {
            const int m = m0 + decltype(I)::value;
            if (m < M1) {
                float4 *o = lds4 + (m << logH) + 2 * h;
                o[0] = make_float4(a[I].x, b[I].x, a[I].y, b[I].y);
                o[1] = make_float4(a[I].z, b[I].z, a[I].w, b[I].w);
            }
            // (row pairs past the end of the tile or in the sample's zero half hold zeros)
            sq[I] = fmaf(a[I].x, a[I].x, fmaf(a[I].y, a[I].y, fmaf(a[I].z, a[I].z, a[I].w * a[I].w)));
            sq[I] = fmaf(b[I].x, b[I].x, fmaf(b[I].y, b[I].y, fmaf(b[I].z, b[I].z, fmaf(b[I].w, b[I].w, sq[I]))));
            if (band) q1 += ((a[I].x + a[I].y) + (a[I].z + a[I].w)) + ((b[I].x + b[I].y) + (b[I].z + b[I].w)); // kernel-uniform
        });
        // The band's sum of squares feeds the spectral Pearson form's error bound (asx_spec_pick, xcorr_dev.h: <= 16 u relative): eight
        // roundings per row pair, ceil(log2 RPQ) <= 3 down this tree, two in the quad sum = 13 u.  (Round 5 ran ONE chain through all
        // RPQ pairs: 8 RPQ + 2 = 42 roundings, which that bound did not cover -- ADVICE r5.)
        ss = tree_sum<RPQ>(sq);
    }
    if (band) {
        const float r1 = quad_sum(q1), r2 = quad_sum(ss);
        // tile-major: a block's band sums are one contiguous run (row-major -- adjacent tiles 8 bytes apart, sixteen blocks of
        // eight XCDs writing into every 128-byte line: k_fwd_cols_r +2 % at 600 x 2400, +7 % at 400 x 1200, profiles/r5_experiments/04_*)
        if (h == 0 && m0 < data_m)
            band[(((size_t)pair * 2 + which) * (size_t)P.ntiles + tile) * (size_t)(M1 / RPQ) + grp] = make_float2(r1, r2);
    }
    // |source|^2 and |sample|^2 (the scale of the float32 error bound of the peak search) from the pass that reads
    // the inputs anyway
    ss = wave_sum_f32(ss);
    if ((threadIdx.x & 63) == 0) nrm_red[threadIdx.x >> 6] = ss;
    __syncthreads();
    (void)lds_fft_static_head<S1, false, true>(lds4, Lc, P.tw1, pre); // every stage but the innermost one
    // A thread that has no work item in the store phase below (rcol_item_of_thread).  (Thread 0's wave waited for this store before its first butterfly
    // of that phase, at the end of the block's life where nothing hides it: 0.997 -> 0.979 ms, profiles/r5_experiments/20_*.)
    if (threadIdx.x == NT - H - 1) {
        float t = nrm_red[0];
        for (int w = 1; w < NT / 64; w++) t += nrm_red[w];
        nrm_part[(pair * 2 + which) * (size_t)P.ntiles + tile] = t;
    }
    // ---- the innermost stage (radix RL, RL consecutive slots per butterfly, no twiddles), untangling and the stores, all
    // from registers.  Butterfly b of a column pair yields the frequencies u_b + MB t (t < RL, MB = M1 / RL, u_b = digit
    // swap of b); their untangling partners M1 - u_b - MB t are element RL-1-t of butterfly b' (u_b' = MB - u_b).  A work
    // item runs BOTH butterflies and stores the 2 RL rows
    //     2 C[u] = S + w (-i D),  2 C[M1-u] = conj(S - w (-i D)),  S = Z[u] + conj Z[M1-u], D = Z[u] - conj Z[M1-u], w = w_{2 M1}^u
    // (w = w_{2 M1}^u_b times the compile-time root w_{2 RL}^t): no LDS pass for the result, no index table.  The two
    // butterflies that pair with themselves (u_b = 0, whose partners are its own elements RL - t and which also yields
    // row M1; u_b = MB/2) share the work item v = 0.
    static_assert(S1::nstages == 3, "three-stage column schedules");
    constexpr int RL = S1::stage(2).R, R0c = S1::stage(0).R, R1c = S1::stage(1).R, MB = M1 / RL;
    static_assert(S1::stage(2).q == 1 && MB == R0c * R1c && MB % 2 == 0, "innermost stage of consecutive slots");
    constexpr int NBPAIRS = (MB / 2) * H;
    static_assert(NBPAIRS <= NT - 1, "one work item per thread at most, none for thread NT - H - 1 (the block's norm)");
    if (const int e = rcol_item_of_thread<NBPAIRS, H, NT>(threadIdx.x); e >= 0) {
        const int g = e & (H - 1), v = e >> logH;  // v = u_b in [0, MB/2)
        const int ub = v, ubp = v == 0 ? MB / 2 : MB - v;
        const int d1 = ub / R0c, d0 = ub - d1 * R0c, e1 = ubp / R0c, e0 = ubp - e1 * R0c;
        const float4 *pa = lds4 + (((d0 * R1c + d1) * RL) << logH) + g, *pb = lds4 + (((e0 * R1c + e1) * RL) << logH) + g;
        Cx2 za[RL], zb[RL];
        static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
            za[TT] = lds_get(pa + (decltype(TT)::value << logH));
            zb[TT] = lds_get(pb + (decltype(TT)::value << logH));
        });
        Bfly<RL, false>::run(za);
        Bfly<RL, false>::run(zb);
        const int cg = 2 * g;
        auto row = [&](int u, Cx2 c, bool conj) __attribute__((always_inline)) {
            st_f4(out + (size_t)u * M2 + c0 + cg, conj ? make_float4(c.re.x, -c.im.x, c.re.y, -c.im.y)
                                                       : make_float4(c.re.x, c.im.x, c.re.y, c.im.y), ASX_RNT & 2);
        };
        if (v != 0) {
            const float2 wu = tw_F(P, (uint32_t)ub * (uint32_t)M2); // w_{2 M1}^u_b = w_F^(u_b M2)
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const Cx2 a = za[t], bq = zb[RL - 1 - t];
                const Cx2 S = Cx2{ a.re + bq.re, a.im - bq.im };
                const Cx2 Dm = Cx2{ a.im + bq.im, bq.re - a.re }; // -i D
                const Cx2 wd = mul_root<2 * RL, t, false>(mulw(Dm, wu));
                row(ub + MB * t, S + wd, false);
                row(ubp + MB * (RL - 1 - t), S - wd, true);
            });
        } else {
            // u_b = 0: Z[MB t] pairs with Z[MB (RL - t)] (t = 0 with itself: rows 0 and M1); w = w_{2 RL}^t
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const Cx2 a = za[t], bq = za[(RL - t) % RL];
                const Cx2 S = Cx2{ a.re + bq.re, a.im - bq.im };
                const Cx2 Dm = Cx2{ a.im + bq.im, bq.re - a.re };
                const Cx2 wd = mul_root<2 * RL, t, false>(Dm);
                row(MB * t, S + wd, false);
                if constexpr (t == 0) row(M1, S - wd, true);
            });
            // u_b = MB/2: Z[MB/2 + MB t] pairs with Z[MB/2 + MB (RL-1-t)]; w = w_{4 RL}^(2t + 1)
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const Cx2 a = zb[t], bq = zb[RL - 1 - t];
                const Cx2 S = Cx2{ a.re + bq.re, a.im - bq.im };
                const Cx2 Dm = Cx2{ a.im + bq.im, bq.re - a.re };
                const Cx2 wd = mul_root<4 * RL, 2 * t + 1, false>(Dm);
                row(MB / 2 + MB * t, S + wd, false);
            });
        }
    }
}

// ---------------------------------------------------------------------------
// k_inv_cols_r: grid (npairs, tiles).  c2r column transforms of length 2 M1 from the rows k1 = 0 .. M1 of Q: tangling
// between rows u and M1 - u into the slots of the M1-point inverse transform, whose outputs are the packed rows
// z[m] = r[2m] + i r[2m+1] of the tile's T real columns; the last stage is consumed from registers (r reaches
// neither HBM nor LDS): |.|-argmax with the reference's tie / sign / NaN rules (src/cross_correlation.c:52-67),
// lags inside the float32 error window appended to the pair's candidate list, as in k_inv_cols.
// Lag of component h of slot (m, g): h = 0 re0, 1 im0, 2 re1, 3 im1 -> (2m + (h & 1)) * M2 + c0 + 2g + (h >> 1).
// ---------------------------------------------------------------------------
template <class S1, int TC, int NT>
__global__ __launch_bounds__(NT, 4) void k_inv_cols_r(const RArgs P, const float2 *__restrict__ qi, size_t pair_pitch,
                                                       AsxPeakWs W, float *__restrict__ r_out, unsigned first_gen)
{
    constexpr int M1 = S1::n, T = TC, logT = asx_ilog2(TC), H = T / 2, logH = logT - 1;
    __shared__ asx_peak_t red[NT / 64];
    __shared__ asx_peak_t s_run0;
    __shared__ float s_b2;
    // TILE-major launch order, grid (npairs, tiles): the tiles of one pair are spread over the life of the launch, so the running
    // maximum a block fetches at its start (`run0`) already holds the maximum of the tiles before it, and a tile that cannot hold
    // the peak -- nearly all of them -- never enters the candidate path with its returning atomic.  Pair-major (the 150 tiles of a
    // pair resident together, run0 == 0 for all of them): 0.405 against 0.367 ms at 600 rows, 0.423 against 0.299 ms per 1024 pairs
    // at 300 rows (profiles/r5_experiments/02_*).
    const size_t pair = blockIdx.x;
    const int tile = rcol_tile_of_block(blockIdx.y, logT);
    if (tile * T >= P.M2) return; // the tile count is rounded up; the tile width is this kernel's own (it reads only)
    {
        // Stagger: the blocks that share a CU run the same program -- a load phase (the tile's rows), then compute phases of about
        // the same length -- and, started together, stay in step: both wait for memory, then both compute.  The blocks of the FIRST
        // generation start half a block's life apart, by the parity of their position in the launch order (the split that measured
        // best); later generations inherit the offset.  0.370 -> 0.349 ms at 600 rows, -3 .. -6 % at 400 (rows and
        // forward columns: nothing, profiles/r5_experiments/05_*).  Speed only.  (What it staggers is not the two blocks of a CU -- those are
        // 256 apart in launch order, same parity -- but HALF THE CHIP against the other half: a read-only kernel whose blocks all take the
        // same time otherwise loads in step and computes in step chip-wide, and the memory system idles between the bursts.  On the round's
        // final kernel: 0.373 ms without, 0.301 with; more phases or other lengths: the same, profiles/r5_experiments/25_*.)
        // The first generation = the blocks resident at once: two per CU for 600- and 400-row tiles, four for 300-row tiles.  (Round 5 had
        // left the 300-row instance alone -- with the FIRST 512 of its 1024 resident blocks treated that way it was 1.5 % slower; with the
        // whole first generation: 0.310 -> 0.278 ms per 1024 pairs of N = 144 000, profiles/r5_experiments/25_*.)
        // first_gen comes from the launcher (occupancy of this kernel x the device's CUs: nothing here assumes 256 CUs).
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        if (gridDim.x * gridDim.y > first_gen && lin < first_gen && (lin & 1u)) // (a launch of one generation has nobody to inherit the offset)
            for (unsigned i = 0; i < (unsigned)(M1 * 27 / 64 / 2); i += 16) __builtin_amdgcn_s_sleep(16); // half a block's life; 64 cycles per unit
    }
    const double shift = W.shift ? W.shift[pair] : 0.0; // non-zero only in the second look (asx_api.hip)
    const int M2 = P.M2, c0 = tile * T;
    const float2 *in = qi + RWS_PAIR(pair) * pair_pitch;
    float4 *lds4 = reinterpret_cast<float4 *>(asx_lds_r);
    const LdsLayout Lc = col_layout(T, logT, NT);
    // the pair's running maximum so far and the width of the near-maximum window: fetched now, used behind the barriers
    const float b2_early = W.bound2[pair];
    // (thread 0 asks for the running maximum here and leaves it in LDS BEHIND the tile loads: stored at once, its wave sat out a memory
    // round trip before it issued its share of the tile's loads, and the block's first barrier waited for that wave: 0.3195 -> 0.3145 ms
    // at 600 rows, -1 % at 400; every thread asking instead: +3 % at 600 rows, profiles/r5_experiments/20_*)
    asx_peak_t run0_early = 0;
    if (threadIdx.x == 0) run0_early = W.pairmax[pair];
    // ---- first stage to run (the innermost, radix RL = R_last, RL consecutive slots per butterfly), fed from HBM --------
    // Butterfly b of a column pair holds the frequencies u_b + MB t (t < RL, MB = M1 / RL, u_b = digit swap of b); their
    // tangling partners M1 - u_b - MB t are element RL-1-t of butterfly b' (u_b' = MB - u_b).  A work item takes BOTH
    // butterflies: 2 RL rows of 16 bytes straight from HBM, the tangling in registers
    //     Z'[u] = S + i conj(w) D,  Z'[M1-u] = conj(S - i conj(w) D),  S = Q[u] + conj Q[M1-u], D = Q[u] - conj Q[M1-u], w = w_{2 M1}^u
    // (w = w_{2 M1}^u_b times the compile-time root w_{2 RL}^t), the two inverse butterflies, 2 RL slots written: no fill
    // phase, no index table, no barrier before the first stage.  u_b = 0 (rows 0, MB, ..., M1: the extra row is its own
    // partner row set) and u_b = MB/2 pair with themselves: one butterfly.
    static_assert(S1::nstages == 3, "three-stage column schedules");
    constexpr StageK KL = S1::stage(2), KM = S1::stage(1);
    constexpr int RL = KL.R, R0c = S1::stage(0).R, R1c = KM.R, MB = M1 / RL;
    static_assert(KL.q == 1 && MB == R0c * R1c && MB % 2 == 0, "innermost stage of consecutive slots");
    constexpr int NITEMS = (MB / 2) * H; // v = 0 takes both butterflies that pair with themselves (u_b = 0 and MB/2)
    const size_t sblock = pair * (size_t)(P.M2 / T) + tile; (void)sblock;
    RSTAMP(2, sblock, 0);
    const TwPre pre_mid = tw_prefetch_exec<S1, 1, true, true, true>(Lc, P.tw1);
    static_assert(NITEMS <= NT, "one work item per thread at most");
    if (const int e = rcol_item_of_thread<NITEMS, H, NT>(threadIdx.x); e >= 0) {
        const int g = e & (H - 1), v = e >> logH;  // v = u_b in [0, MB/2)
        const int ub = v, ubp = v == 0 ? MB / 2 : MB - v; // first rows of the two butterflies' row sets
        const float2 *ca = in + (size_t)ub * M2 + c0 + 2 * g, *cb = in + (size_t)ubp * M2 + c0 + 2 * g;
        // The tangling twiddle w_{2 M1}^u_b = w_F^(u_b M2).  WU_FRONT: asked for IN FRONT of the rows (vmcnt completes in order) and used by
        // both branches below (u_b = 0: exactly 1), which keeps the compiler from sinking the look-up into the one branch that needs it,
        // behind the rows, where its second table value was only issued when the rows had arrived: 0.317 -> 0.3055 ms at 600 rows.  The
        // 400- and 300-row instances lose 1 - 1.6 % with it (profiles/r5_experiments/23_*) and keep the look-up where it was.
        constexpr bool WU_FRONT = ASX_INV_WU && M1 == 600;
        float2 wu_all = make_float2(1.f, 0.f);
        if constexpr (WU_FRONT) wu_all = tw_F(P, (uint32_t)ub * (uint32_t)M2);
        Cx2 A[RL], B[RL];
        static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
            constexpr int t = decltype(TT)::value;
            const float4 x = ld_f4(ca + (size_t)(t * MB) * M2, ASX_RNT & 1);
            const float4 y = ld_f4(cb + (size_t)(t * MB) * M2, ASX_RNT & 1);
            A[t] = Cx2{ v2f{ x.x, x.z }, v2f{ x.y, x.w } };
            B[t] = Cx2{ v2f{ y.x, y.z }, v2f{ y.y, y.w } };
        });
        Cx2 za[RL], zb[RL];
        if (v != 0) {
            const float2 wu = WU_FRONT ? wu_all : tw_F(P, (uint32_t)ub * (uint32_t)M2);
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const Cx2 b = B[RL - 1 - t];
                const Cx2 S = Cx2{ A[t].re + b.re, A[t].im - b.im };
                const Cx2 D = Cx2{ A[t].re - b.re, A[t].im + b.im };
                const Cx2 tt = mul_pos_i(mul_root<2 * RL, t, true>(mulwc(D, wu))); // i conj(w) D, w = wu * w_{2 RL}^t
                za[t] = S + tt;
                const Cx2 m = S - tt;
                zb[RL - 1 - t] = Cx2{ m.re, -m.im };
            });
        } else {
            // u_b = 0: Q[MB t] pairs with Q[MB (RL - t)], t = 0 with the extra row M1; u_b = MB/2: within the butterfly
            const float4 xm = ld_f4(in + (size_t)M1 * M2 + c0 + 2 * g, ASX_RNT & 1);
            const Cx2 QM = Cx2{ v2f{ xm.x, xm.z }, v2f{ xm.y, xm.w } };
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                Cx2 b = QM;
                if constexpr (t != 0) b = A[RL - t];
                const Cx2 S = Cx2{ A[t].re + b.re, A[t].im - b.im };
                const Cx2 D = Cx2{ A[t].re - b.re, A[t].im + b.im };
                if constexpr (WU_FRONT) za[t] = S + mul_pos_i(mul_root<2 * RL, t, true>(mulwc(D, wu_all))); // wu_all = w_F^0 = 1 here
                else za[t] = S + mul_pos_i(mul_root<2 * RL, t, true>(D));
            });
            static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const Cx2 b = B[RL - 1 - t];
                const Cx2 S = Cx2{ B[t].re + b.re, B[t].im - b.im };
                const Cx2 D = Cx2{ B[t].re - b.re, B[t].im + b.im };
                zb[t] = S + mul_pos_i(mul_root<4 * RL, 2 * t + 1, true>(D));
            });
        }
        // slots: butterfly b = d0 * R1 + d1 for u_b = d0 + R0 * d1
        const int d1 = ub / R0c, d0 = ub - d1 * R0c, e1 = ubp / R0c, e0 = ubp - e1 * R0c;
        float4 *pa = lds4 + (((d0 * R1c + d1) * RL) << logH) + g, *pb = lds4 + (((e0 * R1c + e1) * RL) << logH) + g;
        Bfly<RL, true>::run(za);
        Bfly<RL, true>::run(zb);
        static_for<0, RL>([&](auto TT) __attribute__((always_inline)) {
            lds_put(pa + (decltype(TT)::value << logH), za[TT]);
            lds_put(pb + (decltype(TT)::value << logH), zb[TT]);
        });
    }
    if (threadIdx.x == 0) {
        s_run0 = run0_early;
        s_b2 = b2_early;
    }
    RSTAMP(2, sblock, 1);
    // A digitally silent track (zero norm: r is exactly zero everywhere): the running maximum stays zero, which k_finalize
    // reads as index 0, the reference's answer (see k_inv_cols).  Checked HERE, behind the tile loads: in front of them the
    // block waited a memory latency for this one float before it issued anything.  Block-uniform.
    if (b2_early == 0.f && r_out == nullptr) return;
    __syncthreads();
    const TwPre pre_last = lds_fft_static_steps<S1, true, true, S1::nstages - 1, true, 1>(lds4, Lc, P.tw1, pre_mid);
    RSTAMP(2, sblock, 2);
    const asx_peak_t run0 = s_run0;
    const float b2 = s_b2;
    auto last_stage = [&](auto &&sink) __attribute__((always_inline)) {
        lds_last_stage_static<S1, true, true>(lds4, Lc, P.tw1, pre_last, sink);
    };
    const uint32_t uM2 = (uint32_t)M2;
    const bool fast = (tile != 0) && (r_out == nullptr) && shift == 0.0;
    auto examine_again = [&](float thr) __attribute__((always_inline)) {
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const float val[4] = { v[t].re.x, v[t].im.x, v[t].re.y, v[t].im.y };
                const uint32_t i0 = (uint32_t)(2 * (pos0 + t * q)) * uM2 + (uint32_t)(c0 + 2 * g);
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t idx = i0 + (uint32_t)(h & 1) * uM2 + (uint32_t)(h >> 1);
                    const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                    if (key >= thr) cand_append(W, pair, idx, key);
                }
            });
        });
    };
    float thr_again = 0.f;
    bool again = false;
    if (fast) {
        // The scan keeps ONE number per thread: the largest |r| of its 4 R lags (NaNs drop out of fmaxf).  Which lag it was is
        // looked up again only by a tile that can matter (below): with the tile-major launch order nearly every tile finds its
        // maximum under the window of the running maximum it fetched at its start and is done behind one barrier -- no index,
        // no second maximum, no atomic.  (Tracking index, slot and runner-up in the scan: 17 instead of 3 instructions per slot,
        // k_inv_cols_r 0.353 against 0.332 ms at 600 rows, 1.15 against 1.08 ms per 1024 pairs at 400, equal at 600 x 480,
        // 0.31 against 0.32 at 300 rows; profiles/r5_experiments/06_*.)
        float best_m = -INFINITY;
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                best_m = fmaxf(fmaxf(best_m, fmaxf(fabsf(v[t].re.x), fabsf(v[t].im.x))), fmaxf(fabsf(v[t].re.y), fabsf(v[t].im.y)));
            });
        });
        RSTAMP(2, sblock, 4);
        const float wmax = wave_max_nonneg(fmaxf(best_m, 0.f));
        float *redf = reinterpret_cast<float *>(red);
        if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = wmax;
        __syncthreads();
        float bm = redf[0];
        for (int w = 1; w < NT / 64; w++) bm = fmaxf(bm, redf[w]);
        // the final maximum is >= run0: below run0's window nothing of this tile can be the peak or near it (run0 == 0, nothing
        // seen yet, gives a NaN threshold: no exit)
        if (bm < near_max_threshold(peak_key(run0), b2)) return; // block-uniform
        RSTAMP(2, sblock, 5);
        // ---- a tile that can matter (the first generation of blocks, record setters, the peak's tile): the last stage again for
        // the threads that hold its maximum or a near-maximum -- smallest lag and signed value of the maximum, candidates.
        // (The holders publishing for themselves, without the fold and its two barriers: the same time, and 8 bytes of scratch.)
        const float thr = near_max_threshold(fmaxf(bm, peak_key(run0)), b2); // key(run0) is NaN when nothing was seen: fmaxf drops it
        uint32_t my_idx = 0xFFFFFFFFu;
        float my_val = 0.f;
        if (best_m >= thr || best_m == bm) {
            last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
                static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                    constexpr int t = decltype(TT)::value;
                    const float val[4] = { v[t].re.x, v[t].re.y, v[t].im.x, v[t].im.y }; // in lag order: i0, i0 + 1, i0 + M2, i0 + M2 + 1
                    const uint32_t i0 = (uint32_t)(2 * (pos0 + t * q)) * uM2 + (uint32_t)(c0 + 2 * g);
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const uint32_t idx = i0 + (uint32_t)(h >> 1) * uM2 + (uint32_t)(h & 1);
                        const float a = fabsf(val[h]);
                        if (a == bm && idx < my_idx) { my_idx = idx; my_val = val[h]; }
                        if (a >= thr) cand_append(W, pair, idx, a);
                    }
                });
            });
        }
        __syncthreads(); // redf is read by every thread above
        const asx_peak_t mine = my_idx == 0xFFFFFFFFu ? 0 : peak_pack_key(bm, my_idx);
        const asx_peak_t tb = block_peak_max(mine, red);
        if (threadIdx.x == 0) { atomicMax(&W.pairmax[pair], tb); red[0] = tb; }
        __syncthreads();
        // the one thread that holds the tile's best lag: its SIGNED value (the key is |r|) for the spectral Pearson form
        if (W.tile_peak && mine != 0 && mine == red[0]) W.tile_peak[pair * (size_t)(P.M2 / T) + tile] = my_val;
    } else {
        // general form: first tile (lag 0 competes signed), r dumped for tests, shifted keys of the second look
        float best_key = -INFINITY, best_val = 0.f;
        uint32_t best_idx = 0xFFFFFFFFu;
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const float val[4] = { v[t].re.x, v[t].im.x, v[t].re.y, v[t].im.y };
                const uint32_t i0 = (uint32_t)(2 * (pos0 + t * q)) * uM2 + (uint32_t)(c0 + 2 * g);
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t idx = i0 + (uint32_t)(h & 1) * uM2 + (uint32_t)(h >> 1);
                    const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                    if (key > best_key || (key == best_key && idx < best_idx) || best_idx == 0xFFFFFFFFu) { best_key = key; best_idx = idx; best_val = val[h]; }
                    if (r_out) r_out[pair * (size_t)P.nout + idx] = val[h];
                }
            });
        });
        const asx_peak_t mine = best_idx == 0xFFFFFFFFu ? 0 : peak_pack_key(best_key, best_idx);
        asx_peak_t best = block_peak_max(mine, red);
        if (threadIdx.x == 0) { atomicMax(&W.pairmax[pair], best); red[0] = best; }
        __syncthreads();
        if (W.tile_peak && mine != 0 && mine == red[0]) W.tile_peak[pair * (size_t)(P.M2 / T) + tile] = best_val; // one thread: indices are unique
        const float thr = near_max_threshold(peak_key(peak_max(red[0], run0)), b2);
        again = best_key >= thr;
        thr_again = thr;
    }
    if (again) examine_again(thr_again);
    RSTAMP(2, sblock, 3);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static bool schedule_is_r(const AsxStages &st, int n, std::initializer_list<int> radices)
{
    if (st.n != n || st.nstages != (int)radices.size()) return false;
    int i = 0;
    for (int r : radices)
        if (st.radix[i++] != r) return false;
    return true;
}

bool asx_launch_rows_r(const AsxDev &P, const float2 *cx, const float2 *cy, float2 *q, const AsxPeakWs &W, int npairs,
                       hipStream_t s)
{
    const int nrows = P.M1 + 1;
    const size_t pitch = (size_t)nrows * (size_t)P.M2;
    const size_t lds = (size_t)P.M2 * sizeof(float4);
    // chosen by the row length alone: these kernels carry their own schedule and only read the plan's w_M2 table
#define ASX_ROWSR_CASE(nt, two, n, ...)                                                                                         \
    if (P.M2 == ((two) ? 2 * n : n)) {                                                                                          \
        hipLaunchKernelGGL((k_rows_r<Sched<n, __VA_ARGS__>, nt, two>), dim3((unsigned)nrows * (unsigned)npairs),                \
                           dim3((two) ? 2 * nt : nt), lds, s, rargs_of(P), cx, cy, q, nrows, pitch, W);                          \
        return true;                                                                                                            \
    }
    ASX_ROWSR_CASE(128, false, 1200, 12, 10, 10)
    ASX_ROWSR_CASE(128, true, 1200, ASX_ROWS2_SCHED)
    // 480-point rows: ONE wave per block -- a block is 11.5 KB of traffic and a chain of five short phases, so what counts is
    // how many are in flight: sixteen single-wave blocks per CU against eight of two waves (rows 0.93 -> 0.83 ms per 1024
    // pairs of N = 144 000, same box)
    ASX_ROWSR_CASE(64, false, 480, 10, 8, 6)
#undef ASX_ROWSR_CASE
    return false;
}

// Column schedules of the production sample lengths (plan_math.cpp's tuned table):  X(M1, tile width in real columns,
// block size, radices...).  (1200- and 800-row tiles hold only eight real columns: 32-byte input pieces, measured 25 % slower
// in k_fwd_cols_r, and a fed first stage of radix 10 needs 20 rows in flight per thread: the two longest lengths use 600 / 400
// rows with 2400-point rows instead.)  Block sizes are measured (profiles/r4_experiments/10_*, 11_*): 400-row tiles 512
// threads (320, the packed kernels' choice: 12 % slower at N = 480 000), 300-row tiles 256 (320 / 384 / 512: 20-35 % slower).
#define ASX_RCOLS(X) \
    X(600, 16, 512, 10, 10, 6) X(400, 16, 512, 10, 8, 5) X(300, 16, 256, 10, 6, 5)

static void allow_big_lds_r(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Blocks of `fn` the current device holds at once (k_inv_cols_r's first generation, the one its stagger delays half of): occupancy
// x CUs, asked once per kernel and device.  ADVICE r5: the kernel used to hard-code 512 / 1024 = this part's 256 CUs.
static unsigned resident_blocks(const void *fn, int nthreads, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, unsigned> known;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find({ fn, dev });
    if (it != known.end()) return it->second;
    int per_cu = 0, cus = 0;
    unsigned n = 512;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nthreads, lds) == hipSuccess && per_cu > 0 && cus > 0)
        n = (unsigned)per_cu * (unsigned)cus;
    known[{ fn, dev }] = n;
    return n;
}

bool asx_launch_fwd_cols_r(const AsxDev &P, const float *src, const float *smp, float2 *cx, float2 *cy, const AsxPeakWs &W,
                           int npairs, hipStream_t s)
{
    if (!P.col_pairs) return false;
    const size_t pitch = ((size_t)P.M1 + 1) * (size_t)P.M2, lds = (size_t)P.M1 * P.T * sizeof(float2);
    const dim3 grid(rcol_grid_x(P.ntiles, P.logT), 2, npairs);
#define ASX_TRY(m1, t, nt, ...)                                                                                             \
    if (P.T == (t) && schedule_is_r(P.st1, m1, { __VA_ARGS__ })) {                                    \
        allow_big_lds_r((const void *)k_fwd_cols_r<Sched<m1, __VA_ARGS__>, t, nt>, lds);                                    \
        hipLaunchKernelGGL((k_fwd_cols_r<Sched<m1, __VA_ARGS__>, t, nt>), grid, dim3(nt), lds, s, rargs_of(P), src, smp, cx, cy, \
                           W.nrm_part, pitch, W.band);                                                                      \
        return true;                                                                                                        \
    }
    ASX_RCOLS(ASX_TRY)
#undef ASX_TRY
    return false;
}

bool asx_launch_inv_cols_r(const AsxDev &P, const float2 *q, const AsxPeakWs &W, float *r_out, int npairs, hipStream_t s)
{
    if (!P.col_pairs) return false;
    const size_t pitch = ((size_t)P.M1 + 1) * (size_t)P.M2;
#define ASX_TRY(m1, t, nt, ...)                                                                                             \
    if (P.T == (t) && schedule_is_r(P.st1, m1, { __VA_ARGS__ })) {                               \
        const size_t lds = (size_t)(m1) * (t) * sizeof(float2);                                                             \
        const dim3 grid(npairs, rcol_grid_x(P.M2 / (t), asx_ilog2(t)));                                                     \
        const void *fn = (const void *)k_inv_cols_r<Sched<m1, __VA_ARGS__>, t, nt>;                                         \
        allow_big_lds_r(fn, lds);                                                                                           \
        hipLaunchKernelGGL((k_inv_cols_r<Sched<m1, __VA_ARGS__>, t, nt>), grid, dim3(nt), lds, s, rargs_of(P), q, pitch, W, r_out, \
                           resident_blocks(fn, nt, lds));                                                                   \
        return true;                                                                                                        \
    }
    ASX_RCOLS(ASX_TRY)
#undef ASX_TRY
    return false;
}

// rows of the sample matrix per band of the spectral Pearson form = what one lane group of this plan's k_fwd_cols_r loads; 0 = no kernel
int asx_rlayout_band_rows(const AsxDev &P)
{
#define ASX_TRY(m1, t, nt, ...) if (P.T == (t) && schedule_is_r(P.st1, m1, { __VA_ARGS__ })) return 2 * rcol_rows_per_group(m1, nt, t);
    ASX_RCOLS(ASX_TRY)
#undef ASX_TRY
    return 0;
}

bool asx_rlayout_available(const AsxDev &P)
{
    if (!P.col_pairs || P.nout != P.F) return false;
    bool cols = false, rows = false;
#define ASX_TRY(m1, t, nt, ...) if (P.T == (t) && schedule_is_r(P.st1, m1, { __VA_ARGS__ })) cols = true;
    ASX_RCOLS(ASX_TRY)
#undef ASX_TRY
    rows = P.M2 == 1200 || P.M2 == 480 || P.M2 == 2400;
    return cols && rows;
}
