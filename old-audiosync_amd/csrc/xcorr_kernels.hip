// csrc/xcorr_kernels.hip — the gfx950 kernels of the cross-correlation path.
//
// Pipeline for one pair (reference: src/cross_correlation.c:133-307):
//
//   k_fwd_cols   packs source / zero-padded sample as complex (z[j] = x[2j] + i x[2j+1]),
//                column transforms of length M1 in LDS, four-step twiddle      (:159-166, :34-39)
//   k_rows       row transforms of length M2 of BOTH spectra, real-FFT untangling,
//                X * conj(Y), inverse real-FFT tangling, inverse row transforms,
//                inverse four-step twiddle -- all inside LDS                    (:232-233, :237-239)
//   k_inv_cols   inverse column transforms, |.|-argmax with the reference's exact
//                tie/sign/NaN rules, r never written to HBM                     (:52-67, :242)
//   k_finalize   per-pair reduction of the tile partials, lag wrap, segments   (:256-271)
//   k_pearson_*  five float64 sums over the compared segments, coefficient,
//                NaN gate                                                       (:74-116, :272-276)
//
// The transforms are memory-bound (about 6 flop/byte); nothing here uses MFMA.
#include "asx_internal.h"
#include "lds_fft.h"
#include "xcorr_dev.h"

#include <algorithm>
#include <initializer_list>
#include <math.h>
#include <type_traits>
#include <stdlib.h>

// The file is compiled in parts so that a build uses all the cores (Makefile: one object per part):
//   -DASX_PART=<bit mask>; no ASX_PART = everything in one translation unit.
//   1 k_fwd_cols, compile-time schedules    2 k_fwd_cols, run-time schedules
//   4 k_rows, compile-time schedules        8 k_rows, run-time schedules
//   16 k_inv_cols, compile-time schedules   32 k_inv_cols, run-time schedules
//   64 everything else (peak finalisation, refinement, Pearson, helpers, the launch dispatchers)
#ifndef ASX_PART
#define ASX_PART 127
#endif
#define ASX_HAS_PART(bit) ((ASX_PART & (bit)) != 0)

// Diagnostic phase clocks (never in a shipped build; -DASX_STAMPS): lane 0 of every k_rows block
// records s_memtime at phase boundaries into a buffer nothing else reads.
#ifdef ASX_STAMPS
#define ASX_STAMP_AT(kernel, block, slot)                                                            \
    do {                                                                                             \
        if (P.stamps && P.stamp_kernel == (kernel) && threadIdx.x == 0)                              \
            P.stamps[(size_t)(block) * 8 + (slot)] = clock64();                                      \
    } while (0)
#else
#define ASX_STAMP_AT(kernel, block, slot) do {} while (0)
#endif
#define ASX_STAMP(slot) ASX_STAMP_AT(0, task, slot)

// ASX_NT bits (non-temporal accesses): 1 row loads of k_rows, 2 its row stores, 4 loads of the Pearson pass,
// 8 tile stores of k_fwd_cols, 16 tile loads of k_inv_cols
#ifndef ASX_NT
#define ASX_NT 4 // Pearson reads its inputs once: non-temporal loads, 0.213 -> 0.201 ms; the row kernel got slower with them
#endif
typedef float asx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 asx_ld16(const float2 *p, bool nt)
{
    if (nt) { const asx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const asx_f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void asx_st16(float2 *p, float4 v, bool nt)
{
    if (nt) { asx_f4v w = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(w, reinterpret_cast<asx_f4v *>(p)); }
    else *reinterpret_cast<float4 *>(p) = v;
}
#ifndef ASX_ROWS_MIN_WAVES
#define ASX_ROWS_MIN_WAVES 4   // k_rows fits 128 VGPRs (92-119 by variant): four blocks per CU, which is also what its LDS allows
#endif

extern __shared__ __attribute__((aligned(16))) float2 asx_lds[];

// ---------------------------------------------------------------------------
// Column tiles.  A tile is T columns (T even, a power of two) of the [M1][M2] matrix,
// held in LDS as [M1][T/2] float4 slots; a slot is two adjacent columns as they lie in HBM,
// {re0, im0, re1, im1} (lds_fft.h), so one thread transforms two columns with shared
// twiddles and b128 LDS accesses, and tiles move in and out without a shuffle.
// ---------------------------------------------------------------------------
// k_fwd_cols: grid (ntiles, 2, npairs).  blockIdx.y: 0 = source, 1 = sample.
// Packs real samples as complex (z[j] = x[2j] + i x[2j+1]); zero padding and the periodic
// extension of the source (embedded lengths) happen in the loads, never in HBM.
// S1 = void: column schedule, tile width and block size from the plan at run time (any length);
// S1 = Sched<M1, radices...>, TC = tile width, NT = block size: compiled in (production lengths,
// see the launchers).
template <int MAXR, class S1 = void, int TC = 0, int NT = 0>
__global__ __launch_bounds__(ASX_FFT_THREADS_MAX, 4) void k_fwd_cols(const AsxDev *__restrict__ Pp,
                                                                      const float *__restrict__ src,
                                                                      const float *__restrict__ smp,
                                                                      float2 *__restrict__ zxa,
                                                                      float2 *__restrict__ zya,
                                                                      float *__restrict__ nrm_part)
{
    const AsxDev &PD = *Pp; // the plan lives in device memory: uniform scalar loads, taken once
    const AsxKP P = asx_kp(PD);
    __shared__ float nrm_red[ASX_FFT_THREADS_MAX / 64];
    const bool is_smp = blockIdx.y != 0;
    const size_t pair = blockIdx.z;
    constexpr bool STATIC = !std::is_void<S1>::value;
    int T = P.T, logT = P.logT, M1 = P.M1;
    int nthreads = blockDim.x;
    if constexpr (STATIC) { T = TC; logT = asx_ilog2(TC); M1 = S1::n; nthreads = NT; }
    const int tile = col_tile_of_block(blockIdx.x, logT);
    if (tile >= P.ntiles) return; // grid.x is rounded up (col_grid_x)
    const int logH = logT - 1, H = T >> 1, M2 = P.M2;
    const int c0 = tile * T;

    const float *in = is_smp ? smp + pair * (size_t)P.N : src + pair * (size_t)P.src_period;
    const uint32_t valid = is_smp ? P.N : P.src_valid;      // real samples that are not zero padding
    const uint32_t period = is_smp ? P.N : P.src_period;    // source may be periodically extended
    float2 *out = (is_smp ? zya : zxa) + pair * (size_t)P.M;
    const bool even = (M2 & 1) == 0;
    const bool vec_in = even && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    float4 *lds4 = reinterpret_cast<float4 *>(asx_lds);

    const int nelem4 = M1 << logH;
    const LdsLayout Lc = col_layout(T, logT, nthreads);
    const size_t stamp_block = (pair * 2 + blockIdx.y) * P.ntiles + tile;
    (void)stamp_block;
    ASX_STAMP_AT(1, stamp_block, 0);
    TwPre pre;
    if constexpr (STATIC) pre = tw_prefetch_first<S1, false, true>(Lc, P.tw1);
    else pre = tw_prefetch<true>(PD.st1, 0, Lc, P.tw1);
    // all of a thread's tile loads are issued before the first is consumed (ASX_COL_LOADS per
    // round): a rolled loop would pay the HBM latency once per iteration
    // Fast path (block-uniform): full tile, 16-byte aligned rows, no periodic extension, and the
    // zero padding starts on a row boundary -> a row is either all data or all zeros.
    const uint32_t row_reals = 2u * (uint32_t)M2;                 // real samples per matrix row
    const bool fast = vec_in && (c0 + T <= M2) && (valid <= period) && (valid % row_reals == 0u);
    const int data_rows = fast ? (int)(valid / row_reals) : 0;     // rows below this are all data
    // sum of squares of everything this block loads: |source|^2 and |sample|^2 (the scale of the
    // float32 error bound of the peak search) come out of the pass that reads the inputs anyway
    float ss = 0.f;
    for (int e0 = threadIdx.x; e0 < nelem4; e0 += ASX_COL_LOADS * nthreads) {
        float4 v[ASX_COL_LOADS];
        if (fast) {
            static_for<0, ASX_COL_LOADS>([&](auto I) __attribute__((always_inline)) {
                const int e = e0 + decltype(I)::value * nthreads;
                const int cg = e & (H - 1), j1 = e >> logH;
                v[I] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j1 < data_rows) // also false for e >= nelem4 (j1 >= M1 >= data_rows)
                    v[I] = *reinterpret_cast<const float4 *>(in + 2 * ((size_t)j1 * M2 + c0 + 2 * cg));
            });
        } else {
        static_for<0, ASX_COL_LOADS>([&](auto I) __attribute__((always_inline)) {
            const int e = e0 + decltype(I)::value * nthreads;
            const int cg = e & (H - 1), j1 = e >> logH;
            const int j2 = c0 + 2 * cg;
            v[I] = make_float4(0.f, 0.f, 0.f, 0.f); // (re0, im0, re1, im1) as it lies in memory
            if (e < nelem4 && j2 < M2) {
                const uint32_t i0 = 2u * ((uint32_t)j1 * (uint32_t)M2 + (uint32_t)j2);
                if (vec_in && i0 + 3u < valid && i0 + 3u < period) {
                    v[I] = *reinterpret_cast<const float4 *>(in + i0);
                } else {
                    float r[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const uint32_t idx = i0 + h;
                        if (j2 + (h >> 1) < M2 && idx < valid) r[h] = in[idx < period ? idx : idx - period];
                    }
                    v[I] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        });
        }
        static_for<0, ASX_COL_LOADS>([&](auto I) __attribute__((always_inline)) {
            const int e = e0 + decltype(I)::value * nthreads;
            if (e < nelem4) {
                lds4[e] = v[I];
                ss = fmaf(v[I].x, v[I].x, fmaf(v[I].y, v[I].y, fmaf(v[I].z, v[I].z, fmaf(v[I].w, v[I].w, ss))));
            }
        });
    }
    ss = wave_sum_f32(ss);
    if ((threadIdx.x & 63) == 0) nrm_red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ASX_STAMP_AT(1, stamp_block, 1);
    // (the first stage fed straight from HBM, as in k_inv_cols, was measured 2 % slower here: this kernel is bound by its
    // HBM access pattern, not by its LDS passes -- DESIGN.md 5.1, tools/experiments/kernel_switches.patch)
    if constexpr (STATIC) lds_fft_static<S1, false, true>(lds4, Lc, P.tw1, pre);
    else lds_fft<MAXR, false, true>(lds4, PD.st1, Lc, P.tw1, pre);
    ASX_STAMP_AT(1, stamp_block, 2);
    if (threadIdx.x == 0) {
        float t = nrm_red[0];
        for (int w = 1; w < (nthreads + 63) >> 6; w++) t += nrm_red[w];
        nrm_part[(pair * 2 + blockIdx.y) * (size_t)P.ntiles + tile] = t;
    }

    // slot p1 holds frequency k1 = k1_of_pos1[p1] and stays in that slot in HBM; the
    // four-step twiddle w_M^(k1*j2) is applied by k_rows, where k1 is block-uniform.
    const bool full = even && (c0 + T <= M2);
    for (int e = threadIdx.x; e < nelem4; e += nthreads) {
        const int cg = e & (H - 1), p1 = e >> logH;
        const int j2 = c0 + 2 * cg;
        if (full) {
            const float4 v = lds4[e];
            asx_st16(out + (size_t)p1 * M2 + j2, v, ASX_NT & 8);
        } else if (j2 < M2) {
            const float4 v = lds4[e];
            float2 *o = out + (size_t)p1 * M2 + j2;
            if (even) {
                *reinterpret_cast<float4 *>(o) = v;
            } else {
                o[0] = make_float2(v.x, v.y);
                if (j2 + 1 < M2) o[1] = make_float2(v.z, v.w);
            }
        }
    }
    ASX_STAMP_AT(1, stamp_block, 3);
}

// ---------------------------------------------------------------------------
// k_rows: grid (M1/2 + 1, npairs).  Block k1 handles spectrum rows k1 and m1 = M1 - k1
// (the rows that hold each other's k <-> M-k partners).  LDS: A[M2], B[M2] float4 slots,
// pairing the two SPECTRA: A[e] = {X[e], Y[e]} of row k1, B of row m1.
// After the spectral combine the same storage holds C[e] = {Ga[e], Gb[e]},
// the two rows of G, which the inverse transforms as one pair.  Self-paired rows
// (k1 = 0, M1/2) use A only and carry zeros in the second member of C.
// ---------------------------------------------------------------------------
// Row loads of one task, all issued before the first is consumed.  (Issuing them one task
// AHEAD in a persistent block was tried: +50 live VGPRs pushed k_rows to 221 registers = two
// waves per SIMD, and the kernel got 1.6x slower; see DESIGN.md "what did not work".)
struct RowRegs {
    float2 xa[ASX_ROW_STEPS], ya[ASX_ROW_STEPS], xb[ASX_ROW_STEPS], yb[ASX_ROW_STEPS];
};
#define ASX_ROW_WSTEPS ((ASX_ROW_STEPS + 1) / 2) // steps of the 16-byte form: two elements per lane and step
struct RowRegsWide {
    float4 xa[ASX_ROW_WSTEPS], ya[ASX_ROW_WSTEPS], xb[ASX_ROW_WSTEPS], yb[ASX_ROW_WSTEPS];
};

// S2 = void: schedule of the row transforms read from the plan at run time (any length);
// S2 = Sched<M2, radices...>: compiled in (the production lengths, see asx_launch_rows).
template <int MAXR, class S2 = void, int NT = 0>
__global__ __launch_bounds__(ASX_FFT_THREADS_MAX, ASX_ROWS_MIN_WAVES) void k_rows(const AsxDev *__restrict__ Pp,
                                                                  const float2 *__restrict__ zxa,
                                                                  const float2 *__restrict__ zya,
                                                                  float2 *__restrict__ ga,
                                                                  const int4 *__restrict__ row_tasks,
                                                                  int M1, int M2_arg, uint32_t M, AsxPeakWs W)
{
    constexpr bool STATIC = !std::is_void<S2>::value;
    int M2 = M2_arg, nthreads = blockDim.x;
    if constexpr (STATIC) { M2 = S2::n; nthreads = NT; }
    // The start of a block is a chain of dependent memory accesses (plan struct -> index table ->
    // rows); M1/M2/M and the task table come as kernel arguments so that ONE 16-byte load
    // (slots and row numbers) separates the block from its row loads.
    const AsxDev &PD = *Pp; // the plan lives in device memory: uniform scalar loads, taken once
    const AsxKP P = asx_kp(PD);
    const int nrows = M1 / 2 + 1;
    float4 *A4 = reinterpret_cast<float4 *>(asx_lds), *B4 = A4 + M2;
    __shared__ float2 tw_step[2][ASX_ROW_STEPS];

    // one task per block.  (Resident blocks that pull tasks from a ticket counter were measured: no gain, block start-up
    // is not what this kernel waits for -- DESIGN.md 5.1, tools/experiments/kernel_switches.patch.)
    const int task = blockIdx.x;
    {
        const int pair = task / nrows;
        const int4 rt = row_tasks[task - pair * nrows];
        const int pa = rt.x, pb = rt.y;
        const int k1 = rt.z, m1 = rt.w;
        const bool self = (k1 == m1);
        ASX_STAMP(0);
        if (k1 == 0 && threadIdx.x < 64) {
            // Row 0 of a pair also prepares the pair's peak search (k_inv_cols runs after this kernel):
            // the float32 error bound from the norms k_fwd_cols left, and the running maximum and the
            // candidate count back to zero.
            const float *np = W.nrm_part + (size_t)pair * 2 * P.ntiles;
            float sx = 0.f, sy = 0.f;
            for (int t = threadIdx.x; t < P.ntiles; t += 64) { sx += np[t]; sy += np[P.ntiles + t]; }
            sx = wave_sum_f32(sx); sy = wave_sum_f32(sy);
            if (threadIdx.x == 0) {
                W.bound2[pair] = PD.bound_scale * sqrtf(sx) * sqrtf(sy);
                W.pairmax[pair] = 0;
                W.cand_n[pair] = 0;
            }
        }
        LdsLayout Lf;
        Lf.ngroups = self ? 1 : 2; Lf.log_ngroups = 0;
        const int tid = threadIdx.x;
        Lf.elem_stride = 1; Lf.group_stride = M2; Lf.nthreads = nthreads; Lf.tid = tid;
        LdsLayout Li;
        Li.ngroups = 1; Li.log_ngroups = 0;
        Li.elem_stride = 1; Li.group_stride = 0; Li.nthreads = nthreads; Li.tid = tid;
        // Every row load of the thread is issued first; the twiddle lookups below overlap them.
        // Rows of even length move as 16 bytes per lane (two complex values): a block's burst of
        // 8-byte loads takes 2-3x as long to come back (tools/micro/rowload_latency.hip: 21 k cycles
        // against 8-12 k for the same four rows).
        const bool wide = (M2 & 1) == 0;
        const int half = M2 >> 1; // wide: a thread owns the element pairs q = t + nthreads*i, j2 = 2q, 2q+1
        RowRegs L;
        RowRegsWide LW;
        {
            const float2 *gx = zxa + (size_t)pair * M, *gy = zya + (size_t)pair * M;
            if (wide) {
                static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                    const int q = tid + decltype(I)::value * nthreads;
                    // (no zero fill of the registers a thread does not load: every use below is under the same
                    //  conditions, and the fill was 100 v_mov per wave, 5 % of the kernel's VALU instructions)
                    if (q < half) {
                        LW.xa[I] = asx_ld16(gx + (size_t)pa * M2 + 2 * q, ASX_NT & 1);
                        LW.ya[I] = asx_ld16(gy + (size_t)pa * M2 + 2 * q, ASX_NT & 1);
                        if (!self) {
                            LW.xb[I] = asx_ld16(gx + (size_t)pb * M2 + 2 * q, ASX_NT & 1);
                            LW.yb[I] = asx_ld16(gy + (size_t)pb * M2 + 2 * q, ASX_NT & 1);
                        }
                    }
                });
            } else {
                static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
                    const int j2 = tid + decltype(I)::value * nthreads;
                    L.xa[I] = L.ya[I] = L.xb[I] = L.yb[I] = make_float2(0.f, 0.f);
                    if (j2 < M2) {
                        L.xa[I] = gx[(size_t)pa * M2 + j2];
                        L.ya[I] = gy[(size_t)pa * M2 + j2];
                        if (!self) {
                            L.xb[I] = gx[(size_t)pb * M2 + j2];
                            L.yb[I] = gy[(size_t)pb * M2 + j2];
                        }
                    }
                });
            }
        }
        TwPre pre_f;
        if constexpr (STATIC) pre_f = tw_prefetch_first<S2, false, false>(Lf, P.tw2);
        else pre_f = tw_prefetch<false>(PD.st2, 0, Lf, P.tw2);

        // Four-step twiddle of row k1: w_M^(k1*j2).  With j2 = c*(t + nthreads*i) + h (c = 2, h = 0/1
        // for the wide form; c = 1, h = 0 otherwise) it factors into w_M^(k1*c*t) (one two-level
        // lookup per thread), w_M^(k1*c*nthreads*i) (a handful per block) and w_M^(k1*h) (block-uniform).
        const int cw = wide ? 2 : 1;
        const int nsteps = wide ? (half + nthreads - 1) / nthreads : (M2 + nthreads - 1) / nthreads; // <= ASX_ROW_STEPS (launcher)
        if (tid < 2 * nsteps) {
            const int which = tid >= nsteps;
            const int i = tid - which * nsteps;
            const uint32_t row = which ? (uint32_t)m1 : (uint32_t)k1;
            tw_step[which][i] = tw_F(P, 2u * row * (uint32_t)(cw * i * nthreads));
        }
        const uint32_t tcol = (uint32_t)cw * tid < (unsigned)M2 ? (uint32_t)cw * tid : 0u;
        const float2 twa = tw_F(P, 2u * (uint32_t)k1 * tcol);
        const float2 twb = tw_F(P, 2u * (uint32_t)m1 * tcol);
        const float2 wk1 = tw_F(P, 2u * (uint32_t)k1), wm1 = tw_F(P, 2u * (uint32_t)m1); // w_M^k1, w_M^m1
        // (The first row stage fed straight from HBM, as in k_inv_cols, made no difference here: 1.18-1.20 ms with and
        // without, +8 VGPRs -- DESIGN.md 5.1, tools/experiments/kernel_switches.patch.)
        {
            __syncthreads();
            if (wide) {
                static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    const int q = tid + i * nthreads;
                    if (q < half) {
                        const float2 wa0 = cmul(twa, tw_step[0][i]), wa1 = cmul(wa0, wk1);
                        lds_put(A4 + 2 * q, mulw(Cx2{ v2f{ LW.xa[I].x, LW.ya[I].x }, v2f{ LW.xa[I].y, LW.ya[I].y } }, wa0));
                        lds_put(A4 + 2 * q + 1, mulw(Cx2{ v2f{ LW.xa[I].z, LW.ya[I].z }, v2f{ LW.xa[I].w, LW.ya[I].w } }, wa1));
                        if (!self) {
                            const float2 wb0 = cmul(twb, tw_step[1][i]), wb1 = cmul(wb0, wm1);
                            lds_put(B4 + 2 * q, mulw(Cx2{ v2f{ LW.xb[I].x, LW.yb[I].x }, v2f{ LW.xb[I].y, LW.yb[I].y } }, wb0));
                            lds_put(B4 + 2 * q + 1, mulw(Cx2{ v2f{ LW.xb[I].z, LW.yb[I].z }, v2f{ LW.xb[I].w, LW.yb[I].w } }, wb1));
                        }
                    }
                });
            } else {
                static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    const int j2 = tid + i * nthreads;
                    if (j2 < M2) {
                        lds_put(A4 + j2, mulw(Cx2{ v2f{ L.xa[I].x, L.ya[I].x }, v2f{ L.xa[I].y, L.ya[I].y } },
                                              cmul(twa, tw_step[0][i])));
                        if (!self)
                            lds_put(B4 + j2, mulw(Cx2{ v2f{ L.xb[I].x, L.yb[I].x }, v2f{ L.xb[I].y, L.yb[I].y } },
                                                  cmul(twb, tw_step[1][i])));
                    }
                });
            }
        __syncthreads();
        ASX_STAMP(1);
        if constexpr (STATIC) lds_fft_static<S2, false, false>(A4, Lf, P.tw2, pre_f);
        else lds_fft<MAXR, false, false>(A4, PD.st2, Lf, P.tw2, pre_f);
        }
        ASX_STAMP(2);

        // ---- spectral combine: the same storage then holds C[e] = {Ga[e], Gb[e]}, the two rows of G.
        const float2 wA = wk1; // w_M^k1, block-uniform; w_M^(k1 + M1*k2) = wA * w_M2^k2
        TwPre pre_i;
        if (!self) {
            // The common case.  A thread walks SLOT PAIRS (s, s' = M2-1-s), s = t + i*nthreads < M2/2, not bins:
            // digit reversal complements every digit, so the partner bin M2-1-k2 of the bin at slot s sits at
            // slot s'.  Bin(s) of row k1 pairs with bin(s') of row m1 and bin(s') of row k1 with bin(s) of row
            // m1: the thread reads A[s], B[s'], A[s'], B[s] and writes C[s] = {G_k1[s], G_m1[s]} and C[s'] --
            // nobody else touches these four slots, so there is NO barrier between the reads and the writes,
            // the writes are whole 16-byte slots, consecutive lanes walk consecutive slots (forwards in s,
            // backwards in s'): no bank conflicts, no index table (the twiddle table is kept in slot order).
            // Sweeps: table reads of all steps in flight, then the LDS reads, then the arithmetic.
            const int npairs2 = (M2 + 1) >> 1;
            float2 w2a[ASX_ROW_WSTEPS], w2b[ASX_ROW_WSTEPS];
            int sl[ASX_ROW_WSTEPS];
            static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int sx = tid + i * nthreads;
                sl[i] = sx < npairs2 ? sx : 0; // clamped: the loads are unconditional
                w2a[i] = P.tw2s[sl[i]];
                w2b[i] = P.tw2s[M2 - 1 - sl[i]];
            });
            if constexpr (STATIC) pre_i = tw_prefetch_first<S2, true, false>(Li, P.tw2);
            else pre_i = tw_prefetch<false>(PD.st2, PD.st2.nstages - 1, Li, P.tw2);
            Cx2 za[ASX_ROW_WSTEPS], zb[ASX_ROW_WSTEPS], zc[ASX_ROW_WSTEPS], zd[ASX_ROW_WSTEPS];
            static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                za[i] = lds_get(A4 + sl[i]);
                zb[i] = lds_get(B4 + (M2 - 1 - sl[i]));
                zc[i] = lds_get(A4 + (M2 - 1 - sl[i]));
                zd[i] = lds_get(B4 + sl[i]);
            });
            static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int sx = tid + i * nthreads;
                float2 gk0, gm0, gk1, gm1;
                combine_pair(za[i], zb[i], cmul(wA, w2a[i]), gk0, gm0); // bin(s) of k1 with bin(s') of m1
                combine_pair(zc[i], zd[i], cmul(wA, w2b[i]), gk1, gm1); // bin(s') of k1 with bin(s) of m1
                if (sx < npairs2) {
                    const int sp = M2 - 1 - sx;
                    if (sp != sx) {
                        A4[sx] = make_float4(gk0.x, gk0.y, gm1.x, gm1.y);
                        A4[sp] = make_float4(gk1.x, gk1.y, gm0.x, gm0.y);
                    } else { // odd M2: the middle slot is its own partner
                        A4[sx] = make_float4(gk0.x, gk0.y, gm0.x, gm0.y);
                    }
                }
            });
        } else {
        // Self-paired rows (k1 = 0, M1/2; two blocks per pair): bins pair up inside the row through the
        // index table.  Every thread first computes its G values into registers (it reads slots other
        // threads will overwrite), then, after a barrier, writes them.
        float2 gk[ASX_ROW_STEPS], gm[ASX_ROW_STEPS];
        int sa[ASX_ROW_STEPS], sb[ASX_ROW_STEPS]; // -1: nothing to write
        static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int k2 = tid + i * nthreads;
            sa[i] = -1; sb[i] = -1;
            gk[i] = make_float2(0.f, 0.f); gm[i] = make_float2(0.f, 0.f);
            if (k1 == 0) {
                if (k2 == 0) {
                    // DC and Nyquist bins are real: X[0] = Re Z0 + Im Z0, X[M] = Re Z0 - Im Z0
                    const Cx2 z = lds_get(A4);
                    const float P0 = (z.re.x + z.im.x) * (z.re.y + z.im.y);
                    const float PM = (z.re.x - z.im.x) * (z.re.y - z.im.y);
                    sa[i] = 0;
                    gk[i] = make_float2(P0 + PM, P0 - PM);
                } else if (k2 <= M2 / 2) {
                    const int m2 = M2 - k2;
                    sa[i] = P.pos2_of_k2[k2];
                    const int s2 = P.pos2_of_k2[m2];
                    combine_pair(lds_get(A4 + sa[i]), lds_get(A4 + s2), P.tw2[k2], gk[i], gm[i]);
                    if (m2 != k2) sb[i] = s2;
                }
            } else { // k1 == M1/2, M1 even
                if (k2 < (M2 + 1) / 2) {
                    const int m2 = M2 - 1 - k2;
                    sa[i] = P.pos2_of_k2[k2];
                    const int s2 = P.pos2_of_k2[m2];
                    combine_pair(lds_get(A4 + sa[i]), lds_get(A4 + s2), cmul(wA, P.tw2[k2]), gk[i], gm[i]);
                    if (m2 != k2) sb[i] = s2;
                }
            }
        });
        if constexpr (STATIC) pre_i = tw_prefetch_first<S2, true, false>(Li, P.tw2);
        else pre_i = tw_prefetch<false>(PD.st2, PD.st2.nstages - 1, Li, P.tw2);
        __syncthreads();
        static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if (sa[i] >= 0) A4[sa[i]] = make_float4(gk[i].x, gk[i].y, 0.f, 0.f);
            if (sb[i] >= 0) A4[sb[i]] = make_float4(gm[i].x, gm[i].y, 0.f, 0.f);
        });
        }
        __syncthreads();

        ASX_STAMP(3);
        // inverse row transforms of the G pair, digit-reversed in -> natural j2 out
        if constexpr (STATIC) lds_fft_static<S2, true, false>(A4, Li, P.tw2, pre_i);
        else lds_fft<MAXR, true, false>(A4, PD.st2, Li, P.tw2, pre_i);
        ASX_STAMP(4);

        // inverse four-step twiddle conj(w_M^(k1*j2)) / conj(w_M^(m1*j2)), one per member; unrolled so
        // that the LDS reads of all steps are in flight together (a rolled loop pays the LDS latency
        // once per step).  twa / twb ride in registers from the load phase.  Even rows leave as 16
        // bytes per lane, like they came.
        float2 *go = ga + (size_t)pair * M;
        if (wide) {
            Cx2 g0[ASX_ROW_WSTEPS], g1[ASX_ROW_WSTEPS];
            static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int q = tid + i * nthreads;
                const int qq = q < half ? q : 0;
                g0[i] = lds_get(A4 + 2 * qq);
                g1[i] = lds_get(A4 + 2 * qq + 1);
            });
            static_for<0, ASX_ROW_WSTEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int q = tid + i * nthreads;
                if (q < half) {
                    const float2 wa0 = cmul(twa, tw_step[0][i]), wb0 = cmul(twb, tw_step[1][i]);
                    const float2 wa1 = cmul(wa0, wk1), wb1 = cmul(wb0, wm1);
                    const Cx2 h0 = mul2c(g0[i], Cx2{ v2f{ wa0.x, wb0.x }, v2f{ wa0.y, wb0.y } });
                    const Cx2 h1 = mul2c(g1[i], Cx2{ v2f{ wa1.x, wb1.x }, v2f{ wa1.y, wb1.y } });
                    asx_st16(go + (size_t)pa * M2 + 2 * q, make_float4(h0.re.x, h0.im.x, h1.re.x, h1.im.x), ASX_NT & 2);
                    if (!self)
                        asx_st16(go + (size_t)pb * M2 + 2 * q, make_float4(h0.re.y, h0.im.y, h1.re.y, h1.im.y), ASX_NT & 2);
                }
            });
        } else {
            Cx2 gout[ASX_ROW_STEPS];
            static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int j2 = tid + i * nthreads;
                gout[i] = lds_get(A4 + (j2 < M2 ? j2 : 0));
            });
            static_for<0, ASX_ROW_STEPS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int j2 = tid + i * nthreads;
                if (j2 < M2) {
                    const float2 wa = cmul(twa, tw_step[0][i]), wb = cmul(twb, tw_step[1][i]);
                    const Cx2 g = mul2c(gout[i], Cx2{ v2f{ wa.x, wb.x }, v2f{ wa.y, wb.y } });
                    go[(size_t)pa * M2 + j2] = make_float2(g.re.x, g.im.x);
                    if (!self) go[(size_t)pb * M2 + j2] = make_float2(g.re.y, g.im.y);
                }
            });
        }
        ASX_STAMP(5);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// k_inv_cols: grid (ntiles, npairs).  Inverse column transforms; the time-domain
// correlation r[2j] = Re g[j], r[2j+1] = Im g[j] only lives in LDS/registers.
// ---------------------------------------------------------------------------
template <int MAXR, class S1 = void, int TC = 0, int NT = 0>
__global__ __launch_bounds__(ASX_FFT_THREADS_MAX, 4) void k_inv_cols(const AsxDev *__restrict__ Pp, const float2 *__restrict__ ga,
                                                                   AsxPeakWs W, float *__restrict__ r_out)
{
    const AsxDev &PD = *Pp; // the plan lives in device memory: uniform scalar loads, taken once
    const AsxKP P = asx_kp(PD);
    __shared__ asx_peak_t red[ASX_FFT_THREADS_MAX / 64];
    const size_t pair = blockIdx.y;
    constexpr bool STATIC = !std::is_void<S1>::value;
    int T = P.T, logT = P.logT, M1 = P.M1;
    int nthreads = blockDim.x;
    if constexpr (STATIC) { T = TC; logT = asx_ilog2(TC); M1 = S1::n; nthreads = NT; }
    const int tile = col_tile_of_block(blockIdx.x, logT);
    if (tile >= P.ntiles) return; // grid.x is rounded up (col_grid_x)
    // A digitally silent track (a zero norm, e.g. the zero-filled tail of a short capture): r is exactly zero
    // everywhere, the reference's scan returns index 0 (src/cross_correlation.c:52-67), which is what a running
    // maximum left at zero means to k_finalize.  Without this every one of the 2N lags would be a near-tie of the
    // maximum 0 inside a window of width 0, the lists would overflow and the synchronous entry points would
    // re-evaluate all of them exactly (seconds at N = 1 440 000).  Block-uniform; r_out (tests) still wants zeros.
    if (W.bound2[pair] == 0.f && r_out == nullptr) return;
    const double shift = W.shift ? W.shift[pair] : 0.0; // block-uniform; non-zero only in the second look (second_look, asx_api.hip)
    const int logH = logT - 1, H = T >> 1, M2 = P.M2;
    const int c0 = tile * T;
    const float2 *in = ga + pair * (size_t)P.M;
    const bool even = (M2 & 1) == 0;
    float4 *lds4 = reinterpret_cast<float4 *>(asx_lds);

    const int nelem4 = M1 << logH;
    const LdsLayout Lc = col_layout(T, logT, nthreads);
    const size_t stamp_block = pair * P.ntiles + tile;
    (void)stamp_block;
    ASX_STAMP_AT(2, stamp_block, 0);
    // The pair's running maximum so far (other tiles publish theirs with atomicMax below) and the width
    // of the near-maximum window are consumed after the first pass of the scan, at the very end.  Loaded where
    // they are used, the block waits 2 700 cycles for an L2 round trip there (phase stamps).  Thread 0 fetches
    // them now and parks them in LDS: its wave waits for them together with its tile loads, and everybody reads
    // them behind the barriers of the transform.
    __shared__ asx_peak_t s_run0;
    __shared__ float s_b2;
    if (threadIdx.x == 0) {
        s_run0 = W.pairmax[pair];
        s_b2 = W.bound2[pair];
    }
    TwPre pre;
    if constexpr (STATIC) pre = tw_prefetch_first<S1, true, true, true>(Lc, P.tw1);
    else pre = tw_prefetch<true>(PD.st1, PD.st1.nstages - 1, Lc, P.tw1);
#ifndef ASX_INV_FED
#define ASX_INV_FED 1 // first inverse stage fed straight from HBM (compile-time schedules, full tiles)
#endif
    TwPre pre_last;
    bool filled = false;
    if constexpr (STATIC && ASX_INV_FED) {
        if (even && (c0 + T <= M2)) { // block-uniform: full tile
            // No fill phase: the first stage to run (innermost, 10 consecutive rows per butterfly) takes its
            // inputs from HBM -- the thread's loads are all in flight together, as in the fill loop -- and writes
            // its outputs to LDS: one LDS write + read pass and one barrier less per tile.
            ASX_STAMP_AT(2, stamp_block, 1);
            pre_last = lds_fft_static_head_fed<S1, true, true>(lds4, Lc, P.tw1, pre,
                [&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
                    const float2 *col = in + (size_t)pos0 * M2 + c0 + 2 * g;
                    static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                        constexpr int t = decltype(TT)::value;
                        const float4 x = asx_ld16(col + (size_t)(t * q) * M2, ASX_NT & 16);
                        v[t] = Cx2{ v2f{ x.x, x.z }, v2f{ x.y, x.w } };
                    });
                });
            filled = true;
        }
    }
    if (!filled) {
    for (int e0 = threadIdx.x; e0 < nelem4; e0 += ASX_COL_LOADS * nthreads) {
        float4 v[ASX_COL_LOADS];
        static_for<0, ASX_COL_LOADS>([&](auto I) __attribute__((always_inline)) {
            const int e = e0 + decltype(I)::value * nthreads;
            const int cg = e & (H - 1), p1 = e >> logH;
            const int j2 = c0 + 2 * cg;
            v[I] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (even && (c0 + T <= M2)) { // block-uniform: full tile
                if (e < nelem4) v[I] = *reinterpret_cast<const float4 *>(in + (size_t)p1 * M2 + j2);
            } else if (e < nelem4 && j2 < M2) {
                const float2 *g = in + (size_t)p1 * M2 + j2;
                if (even) {
                    v[I] = *reinterpret_cast<const float4 *>(g);
                } else {
                    const float2 a = g[0];
                    const float2 b = (j2 + 1 < M2) ? g[1] : make_float2(0.f, 0.f);
                    v[I] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        });
        static_for<0, ASX_COL_LOADS>([&](auto I) __attribute__((always_inline)) {
            const int e = e0 + decltype(I)::value * nthreads;
            if (e < nelem4) lds4[e] = v[I];
        });
    }
    __syncthreads();
    ASX_STAMP_AT(2, stamp_block, 1);
    // every inverse stage but the last: the last one's outputs are consumed from registers below
    // (r reaches neither HBM nor LDS; LDS keeps that stage's input, so the stage can be run again)
    if constexpr (STATIC) pre_last = lds_fft_static_head<S1, true, true>(lds4, Lc, P.tw1, pre);
    else lds_fft<MAXR, true, true>(lds4, PD.st1, Lc, P.tw1, pre); // run-time schedule: the whole transform, r into LDS
    }
    ASX_STAMP_AT(2, stamp_block, 2);

    // ANY earlier value of the running maximum is a lower bound of the final one, so a stale read merely admits
    // more candidates (k_finalize filters them against the final maximum).
    const asx_peak_t run0 = s_run0;
    const float b2 = s_b2;
    if constexpr (!STATIC) {
        // Run-time schedules (lengths outside the reference's six): r lies in LDS and is scanned there.  The
        // scan from the last stage's registers below, instantiated inside the switch over eleven radix bodies,
        // pushed these kernels into scratch (k_inv_cols 0.77 ms against 0.43 ms for the compiled-in schedule).
        const bool fastg = even && (c0 + T <= M2) && (P.nout == P.F) && (tile != 0) && (r_out == nullptr) && shift == 0.0;
        auto examine_slot = [&](int e, float4 g, float thr) {
            const int cg = e & (H - 1), j1 = e >> logH;
            const int j2 = c0 + 2 * cg;
            if (j2 >= M2) return;
            const uint32_t i0 = 2u * ((uint32_t)j1 * (uint32_t)M2 + (uint32_t)j2);
            const float val[4] = { g.x, g.y, g.z, g.w }; // slot = {re0, im0, re1, im1}: four consecutive lags
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const uint32_t idx = i0 + h;
                if (idx < P.nout && j2 + (h >> 1) < M2) {
                    const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                    if (key >= thr) cand_append(W, pair, idx, key);
                }
            }
        };
        if (fastg) {
            // pass 1: per thread the largest and second largest slot maximum; a thread meets its slots in
            // increasing lag order, so a strict '>' keeps the earliest of equal maxima
            float best_m = -INFINITY, second_m = -INFINITY;
            int best_e = threadIdx.x;
            for (int e = threadIdx.x; e < nelem4; e += nthreads) {
                const float4 g = lds4[e];
                const float m = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))); // NaNs drop out
                if (m > best_m) { second_m = best_m; best_m = m; best_e = e; }
                else if (m > second_m) second_m = m;
            }
            const float4 gb = lds4[best_e];
            uint32_t my_idx;
            {
                const int cg = best_e & (H - 1), j1 = best_e >> logH;
                const uint32_t i0 = 2u * ((uint32_t)j1 * (uint32_t)M2 + (uint32_t)(c0 + 2 * cg));
                const uint32_t h = fabsf(gb.x) == best_m ? 0u : fabsf(gb.y) == best_m ? 1u : fabsf(gb.z) == best_m ? 2u : 3u;
                my_idx = i0 + h;
            }
            const float wmax = wave_max_nonneg(fmaxf(best_m, 0.f));
            unsigned long long holders = __ballot(best_m == wmax);
            uint32_t widx = 0xFFFFFFFFu;
            while (holders) {
                const int l = __ffsll((long long)holders) - 1;
                const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)my_idx, l);
                widx = li < widx ? li : widx;
                holders &= holders - 1;
            }
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = widx == 0xFFFFFFFFu ? 0 : peak_pack_key(wmax, widx);
            __syncthreads();
            asx_peak_t tb = red[0];
            for (int w = 1; w < (int)((nthreads + 63) >> 6); w++) tb = peak_max(tb, red[w]);
            if (threadIdx.x == 0) atomicMax(&W.pairmax[pair], tb);
            const float thr = near_max_threshold(peak_key(peak_max(tb, run0)), b2);
            if (best_m >= thr) {
                if (second_m >= thr) {
                    for (int e = threadIdx.x; e < nelem4; e += nthreads) {
                        const float4 g = lds4[e];
                        const float m = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
                        if (m >= thr) examine_slot(e, g, thr);
                    }
                } else {
                    examine_slot(best_e, gb, thr);
                }
            }
        } else {
            float best_key = -INFINITY;
            uint32_t best_idx = 0xFFFFFFFFu;
            for (int e = threadIdx.x; e < nelem4; e += nthreads) {
                const int cg = e & (H - 1), j1 = e >> logH;
                const int j2 = c0 + 2 * cg;
                if (j2 < M2) {
                    const uint32_t i0 = 2u * ((uint32_t)j1 * (uint32_t)M2 + (uint32_t)j2);
                    const float4 g = lds4[e];
                    const float val[4] = { g.x, g.y, g.z, g.w };
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const uint32_t idx = i0 + h;
                        if (idx < P.nout && j2 + (h >> 1) < M2) {
                            const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                            if (key > best_key || (key == best_key && idx < best_idx) || best_idx == 0xFFFFFFFFu) { best_key = key; best_idx = idx; }
                            if (r_out) r_out[pair * (size_t)P.nout + idx] = val[h];
                        }
                    }
                }
            }
            asx_peak_t best = best_idx == 0xFFFFFFFFu ? 0 : peak_pack_key(best_key, best_idx);
            best = block_peak_max(best, red);
            if (threadIdx.x == 0) { atomicMax(&W.pairmax[pair], best); red[0] = best; }
            __syncthreads();
            const float thr = near_max_threshold(peak_key(peak_max(red[0], run0)), b2);
            for (int e = threadIdx.x; e < nelem4; e += nthreads) examine_slot(e, lds4[e], thr);
        }
        ASX_STAMP_AT(2, stamp_block, 3);
    } else {
    auto last_stage = [&](auto &&sink) __attribute__((always_inline)) {
        lds_last_stage_static<S1, true, true>(lds4, Lc, P.tw1, pre_last, sink);
    };

    // Output T of a butterfly is row j1 = pos0 + T*q of column pair g: four consecutive lags
    // {re0, im0, re1, im1} from i0 = 2*(j1*M2 + c0 + 2g).
    // Peak search (src/cross_correlation.c:52-67): largest key, smallest lag among equal keys.
    // Fast path (block-uniform): the tile is full, every lag counts, lag 0 (the signed one) is
    // not in it and r is not being dumped -> one packed maximum per slot, indices resolved at the end.
    const bool fast = even && (c0 + T <= M2) && (P.nout == P.F) && (tile != 0) && (r_out == nullptr) && shift == 0.0;
    // second look (rare, one instantiation for both paths): the thread runs its last stage again and
    // appends every valid lag whose key is inside the window
    auto examine_again = [&](float thr) __attribute__((always_inline)) {
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            const int j2 = c0 + 2 * g;
            if (j2 >= M2) return;
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const float val[4] = { v[t].re.x, v[t].im.x, v[t].re.y, v[t].im.y };
                const uint32_t i0 = 2u * ((uint32_t)(pos0 + t * q) * (uint32_t)M2 + (uint32_t)j2);
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t idx = i0 + h;
                    if (idx < P.nout && j2 + (h >> 1) < M2) {
                        const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                        if (key >= thr) cand_append(W, pair, idx, key);
                    }
                }
            });
        });
    };
    float thr_again = 0.f;
    bool again = false;
    if (fast) {
        // pass 1: per thread the largest and second largest slot maximum
        float best_m = -INFINITY, second_m = -INFINITY;
        uint32_t best_i0 = 0xFFFFFFFFu;
        float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const float4 s4 = make_float4(v[t].re.x, v[t].im.x, v[t].re.y, v[t].im.y);
                const float m = fmaxf(fmaxf(fabsf(s4.x), fabsf(s4.y)), fmaxf(fabsf(s4.z), fabsf(s4.w))); // NaNs drop out
                const uint32_t i0 = 2u * ((uint32_t)(pos0 + t * q) * (uint32_t)M2 + (uint32_t)(c0 + 2 * g));
                if (m > best_m || (m == best_m && i0 < best_i0)) { second_m = best_m; best_m = m; gb = s4; best_i0 = i0; }
                else if (m > second_m) second_m = m;
            });
        });
        // lag order inside a slot = its memory order {re0, im0, re1, im1}
        const uint32_t hh = fabsf(gb.x) == best_m ? 0u : fabsf(gb.y) == best_m ? 1u : fabsf(gb.z) == best_m ? 2u : 3u;
        const uint32_t my_idx = best_i0 + hh;
        // wave maximum in registers, smallest lag among the lanes that hold it, one entry per wave
        ASX_STAMP_AT(2, stamp_block, 4);
        const float wmax = wave_max_nonneg(fmaxf(best_m, 0.f));
        unsigned long long holders = __ballot(best_m == wmax);
        uint32_t widx = 0xFFFFFFFFu;
        while (holders) {
            const int l = __ffsll((long long)holders) - 1;
            const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)my_idx, l);
            widx = li < widx ? li : widx;
            holders &= holders - 1;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = widx == 0xFFFFFFFFu ? 0 : peak_pack_key(wmax, widx);
        __syncthreads();
        // every thread folds the wave entries itself: no second barrier to broadcast the result
        asx_peak_t tb = red[0];
        for (int w = 1; w < (int)((nthreads + 63) >> 6); w++) tb = peak_max(tb, red[w]);
        if (threadIdx.x == 0) atomicMax(&W.pairmax[pair], tb);
        // Second look: lags within the float32 error window of the largest key known so far.  Almost every
        // thread is below the threshold; the one that holds the maximum usually has no second slot near
        // it and examines just that slot; a thread with more runs its last stage again.
        ASX_STAMP_AT(2, stamp_block, 5);
        const float thr = near_max_threshold(peak_key(peak_max(tb, run0)), b2);
        thr_again = thr;
        if (best_m >= thr) {
            if (second_m >= thr) {
                again = true;
            } else {
                const float val[4] = { gb.x, gb.y, gb.z, gb.w };
#pragma unroll
                for (int h = 0; h < 4; h++)
                    if (fabsf(val[h]) >= thr) cand_append(W, pair, best_i0 + h, fabsf(val[h]));
            }
        }
    } else {
        // general form: first tile (lag 0 competes signed), ragged or embedded tiles, r dumped for tests
        float best_key = -INFINITY;
        uint32_t best_idx = 0xFFFFFFFFu;
        last_stage([&](auto RC, auto &v, int g, int pos0, int q) __attribute__((always_inline)) {
            const int j2 = c0 + 2 * g;
            if (j2 >= M2) return;
            static_for<0, decltype(RC)::value>([&](auto TT) __attribute__((always_inline)) {
                constexpr int t = decltype(TT)::value;
                const float val[4] = { v[t].re.x, v[t].im.x, v[t].re.y, v[t].im.y };
                const uint32_t i0 = 2u * ((uint32_t)(pos0 + t * q) * (uint32_t)M2 + (uint32_t)j2);
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t idx = i0 + h;
                    if (idx < P.nout && j2 + (h >> 1) < M2) {
                        const float key = shift == 0.0 ? peak_key_of(val[h], idx) : peak_key_shifted(val[h], idx, shift);
                        if (key > best_key || (key == best_key && idx < best_idx) || best_idx == 0xFFFFFFFFu) { best_key = key; best_idx = idx; }
                        if (r_out) r_out[pair * (size_t)P.nout + idx] = val[h];
                    }
                }
            });
        });
        asx_peak_t best = best_idx == 0xFFFFFFFFu ? 0 : peak_pack_key(best_key, best_idx);
        best = block_peak_max(best, red);
        if (threadIdx.x == 0) { atomicMax(&W.pairmax[pair], best); red[0] = best; }
        __syncthreads();
        const float thr = near_max_threshold(peak_key(peak_max(red[0], run0)), b2);
        again = best_key >= thr;
        thr_again = thr;
    }
    if (again) examine_again(thr_again);
    ASX_STAMP_AT(2, stamp_block, 3);
    } // compiled-in schedules
}

#if ASX_HAS_PART(64)
// ---------------------------------------------------------------------------
// k_finalize: grid (npairs).  Reduce tile partials, wrap the lag, pick segments.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(ASX_THREADS) void k_finalize(const AsxDev *__restrict__ Pp, AsxPeakWs W, AsxSeg *__restrict__ seg,
                                                           uint32_t pair_base)
{
    const uint32_t N = Pp->N;
    __shared__ uint32_t nsel;
    const size_t pair = blockIdx.x;
    const asx_peak_t best = W.pairmax[pair];        // float32 maximum, smallest lag among equal keys
    const uint32_t ntot = W.cand_n[pair];
    const uint32_t n = ntot < W.cap ? ntot : W.cap;
    // the tiles collected against the running maximum; keep what is near the FINAL maximum
    const float thr = near_max_threshold(peak_key(best), W.bound2[pair]);
    if (threadIdx.x == 0) nsel = 0;
    __syncthreads();
    const AsxCand *c = W.cand + pair * (size_t)W.cap;
    uint32_t *out = W.refine_idx + pair * (size_t)W.cap;
    for (uint32_t i = threadIdx.x; i < n; i += ASX_THREADS) {
        const AsxCand e = c[i];
        if (e.key >= thr) out[atomicAdd(&nsel, 1u)] = e.idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        AsxSeg sg = make_seg(best ? peak_index(best) : 0u, N);
        // One candidate: the float32 argmax is unambiguous.  More than the list holds (a signal periodic in
        // more than `cap` lags, an offset of hundreds of deviations in both tracks): the float32 argmax is only a
        // placeholder.  The pair is MARKED (its ret becomes ASX_RET_INEXACT in k_pearson_final), counted
        // (asx_plan_peak_overflows) and put on the list the entry points read to take the second look
        // (asx_api.hip: resolve_overflows): the reference's scan has no candidate limit (src/cross_correlation.c:52-67).
        const bool over = ntot > W.cap;
        if (over) sg.flags = ASX_SEG_INEXACT;
        seg[pair] = sg;
        W.refine_n[pair] = (!over && nsel >= 2u) ? nsel : 0u;
        if (over) {
            atomicAdd(W.overflows, 1ull);
            if (W.over_list) {
                const uint32_t slot = atomicAdd(W.over_n, 1u);
                if (slot < W.over_cap) W.over_list[slot] = pair_base + (uint32_t)pair;
                if (W.over_host) (void)__hip_atomic_fetch_add(W.over_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Exact re-evaluation of near-tied lags: r[k] = sum_{n<N} source[(n+k) mod 2N] * sample[n]
// (the identity behind src/cross_correlation.c:232-239), accumulated as an
// unevaluated sum of two doubles (error ~1e-30 relative: the products of float32 inputs are exact
// in float64, those of float64 inputs carry their rounding error along via fma).
// grid (ASX_DOT_BLOCKS, npairs): block x of a pair takes candidates x, x + gridDim.x, ...
// ---------------------------------------------------------------------------
struct dd_t {
    double hi, lo;
};
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b)
{
    const double s = a.hi + b.hi;
    const double bb = s - a.hi;
    double e = (a.hi - (s - bb)) + (b.hi - bb);
    e += a.lo + b.lo;
    dd_t r;
    r.hi = s + e;
    r.lo = e - (r.hi - s);
    return r;
}

template <typename TIn>
__global__ __launch_bounds__(ASX_THREADS) void k_refine_dots(const AsxDev *__restrict__ Pp, const TIn *__restrict__ src,
                                                              const TIn *__restrict__ smp, AsxPeakWs W)
{
    __shared__ double red[2][ASX_THREADS / 64];
    const size_t pair = blockIdx.y;
    const uint32_t ncand = W.refine_n[pair];
    if (blockIdx.x >= ncand) return;
    const uint32_t N = Pp->N, L = 2u * N;
    const TIn *x = src + pair * (size_t)L;
    const TIn *y = smp + pair * (size_t)N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t c = blockIdx.x; c < ncand; c += gridDim.x) {
        const uint32_t k = W.refine_idx[pair * (size_t)W.cap + c];
        double hi = 0.0, lo = 0.0;
        for (uint32_t n = threadIdx.x; n < N; n += ASX_THREADS) {
            uint32_t i = n + k;
            if (i >= L) i -= L;
            const double a = (double)x[i], b = (double)y[n];
            const double p = a * b;
            double pe = 0.0;
            if (sizeof(TIn) == sizeof(double)) pe = fma(a, b, -p);
            const double s = hi + p;
            const double bb = s - hi;
            lo += ((hi - (s - bb)) + (p - bb)) + pe;
            hi = s;
        }
        dd_t acc;
        acc.hi = hi; acc.lo = lo;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dd_t o;
            o.hi = __shfl_xor(acc.hi, off, 64);
            o.lo = __shfl_xor(acc.lo, off, 64);
            acc = dd_add(acc, o);
        }
        if (lane == 0) { red[0][wave] = acc.hi; red[1][wave] = acc.lo; }
        __syncthreads();
        if (threadIdx.x == 0) {
            dd_t t;
            t.hi = red[0][0]; t.lo = red[1][0];
            for (int w = 1; w < ASX_THREADS / 64; w++) {
                dd_t o;
                o.hi = red[0][w]; o.lo = red[1][w];
                t = dd_add(t, o);
            }
            W.refine_val[pair * (size_t)W.cap + c] = t.hi + t.lo;
        }
        __syncthreads();
    }
}

// grid (npairs): the reference's max_abs_index rule (src/cross_correlation.c:52-67) on the exact values:
// key(0) = r[0] signed, key(i) = |r[i]|, largest key, smallest lag among equal keys; a NaN key never
// wins unless it sits at lag 0.
__global__ __launch_bounds__(ASX_THREADS) void k_refine_pick(const AsxDev *__restrict__ Pp, AsxPeakWs W, AsxSeg *__restrict__ seg)
{
    __shared__ double rkey[ASX_THREADS / 64];
    __shared__ uint32_t ridx[ASX_THREADS / 64];
    const size_t pair = blockIdx.x;
    const uint32_t n = W.refine_n[pair];
    if (n < 2u) return;
    double bk = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < n; i += ASX_THREADS) {
        const uint32_t idx = W.refine_idx[pair * (size_t)W.cap + i];
        const double v = W.refine_val[pair * (size_t)W.cap + i];
        double key;
        if (idx == 0u) key = (v != v) ? (double)INFINITY : v + 0.0;
        else { key = fabs(v); if (key != key) key = -(double)INFINITY; }
        if (key > bk || (key == bk && idx < bi)) { bk = key; bi = idx; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_xor(bk, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rkey[wave] = bk; ridx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ASX_THREADS / 64; w++)
            if (rkey[w] > bk || (rkey[w] == bk && ridx[w] < bi)) { bk = rkey[w]; bi = ridx[w]; }
        if (bi != 0xFFFFFFFFu) seg[pair] = make_seg(bi, Pp->N);
    }
}

// ---------------------------------------------------------------------------
// Pearson coefficient (src/cross_correlation.c:74-116).  The reference makes two passes (means, then
// the centred sums); here ONE streaming pass gives the same numbers: every thread accumulates its
// elements relative to a pivot (its own first element), turns the five sums into
// (count, mean, centred second moments) and the partitions are merged pairwise with the exact update
// formulas (Chan et al.): no large-offset cancellation anywhere (offset 1e6, amplitude 0.1 is fine).
// The merge tree is fixed, and x and y go through the same operations, so identical segments give
// bit-identical Mxx, Myy, Cxy and therefore exactly +-1.0 (tests/test_cross_correlation.c:29).
// ---------------------------------------------------------------------------
// (PStat, pstat_merge, pstat_wave_merge: xcorr_dev.h -- shared with pearson_spectral.hip)
// steps of k_pearson_partial's loop whose loads are in flight together (0 = one step at a time, as it was): the direct form 0.200 -> 0.190 ms
// per 124 pairs of N = 1 440 000, the spectral form's remainder 0.090 -> 0.087 (profiles/r5_experiments/24_*); the same additions in the same order
#ifndef ASX_PEARSON_DEEP
#define ASX_PEARSON_DEEP 4
#endif
template <typename TIn, bool SPEC>
__global__ __launch_bounds__(ASX_THREADS) void k_pearson_partial(const TIn *__restrict__ src,
                                                                  const TIn *__restrict__ smp,
                                                                  size_t src_pitch, size_t smp_pitch,
                                                                  uint32_t basis_len,
                                                                  const AsxSeg *__restrict__ seg,
                                                                  double *__restrict__ psums, AsxSpecWs V)
{
    __shared__ double red[6][ASX_THREADS / 64];
    const size_t pair = blockIdx.y;
    AsxSeg s = seg[pair];
    // SPEC (the spectral Pearson form): what this pair still needs read -- nothing, its wrap-around part, or its segment -- follows from
    // the sums k_pearson_prep left; every block works that out for itself (pair-uniform: scalar loads and arithmetic)
    // Block 0 also RECORDS the mode (the spare header slot): k_pearson_final_spec, in another translation unit, reads it instead of
    // deciding again -- with -ffp-contract=fast two inlining contexts may round `bound <= tol` differently at the threshold, and a final
    // kernel that disagreed with the work list would read partial sums written for another mode.  One writer, a later
    // kernel reads: no fence.
    if constexpr (SPEC) {
        const AsxSpecPick d = asx_spec_pick(s, V.part + pair * (size_t)(V.nb * 4), V.nb, V.hdr + pair * ASX_SPEC_HDR, V.tol, V.N);
        s = d.work;
        if (blockIdx.x == 0 && threadIdx.x == 0) V.hdr[pair * ASX_SPEC_HDR + 3] = (double)d.mode;
    }
    // gridDim.x partial blocks per pair (asx_pearson_blocks: by the basis length alone); the final kernel merges
    // exactly gridDim.x entries
    const uint32_t chunk = (basis_len + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    uint64_t hi = lo + chunk;
    if (hi > s.len) hi = s.len;
    if (lo >= s.len) {
        // nothing of the segment falls to this block (block-uniform): an empty record (pstat_merge skips n == 0) -- the whole
        // launch for a pair the spectral form has settled (pearson_spectral.hip), the tail of a short segment otherwise
        if (threadIdx.x == 0) psums[(pair * gridDim.x + blockIdx.x) * 6] = 0.0;
        return;
    }
    const TIn *x = src + pair * src_pitch + s.src_off;
    const TIn *y = smp + pair * smp_pitch + s.smp_off;
    uint64_t i = lo + 4u * threadIdx.x;
    double px = 0, py = 0; // pivots: the first element this thread meets
#if ASX_PEARSON_DEEP
    // (taken from the first 16-byte load below when the thread has one: a scalar load and the wait for it in front of the loop otherwise)
    if (i < hi && !(i + 3 < hi)) { px = (double)x[i]; py = (double)y[i]; }
#else
    if (i < hi) { px = (double)x[i]; py = (double)y[i]; }
#endif
    double sx = 0, sy = 0, sxy = 0, sxx = 0, syy = 0;
    uint32_t cnt = 0;
    auto add = [&](double a, double b) {
        const double da = a - px, db = b - py;
        sx += da;
        sy += db;
        sxy += da * db;
        sxx += da * da;
        syy += db * db;
        cnt++;
    };
    // four consecutive elements per lane and step: 16-byte loads (the segments start at any element,
    // so the vector type only promises element alignment), then the few elements that are left
    typedef TIn vec4u __attribute__((ext_vector_type(4), aligned(sizeof(TIn))));
    auto ld4 = [&](const TIn *p) __attribute__((always_inline)) {
        return (ASX_NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const vec4u *>(p)) : *reinterpret_cast<const vec4u *>(p);
    };
#if ASX_PEARSON_DEEP
    // ASX_PEARSON_DEEP steps at a time: their 2 * ASX_PEARSON_DEEP loads in flight together, then the same additions in the same order (the
    // loop used to wait for each step's two loads before it issued the next step's: 16 memory round trips per block, one after the other)
    {
        bool first = i + 3 < hi;
        constexpr uint32_t STEP = 4u * ASX_THREADS;
        for (; i + 3 + (uint64_t)(ASX_PEARSON_DEEP - 1) * STEP < hi; i += (uint64_t)ASX_PEARSON_DEEP * STEP) {
            vec4u a[ASX_PEARSON_DEEP], b[ASX_PEARSON_DEEP];
#pragma unroll
            for (int u = 0; u < ASX_PEARSON_DEEP; u++) { a[u] = ld4(x + i + (uint64_t)u * STEP); b[u] = ld4(y + i + (uint64_t)u * STEP); }
            if (first) { px = (double)a[0].x; py = (double)b[0].x; first = false; }
#pragma unroll
            for (int u = 0; u < ASX_PEARSON_DEEP; u++) {
                add((double)a[u].x, (double)b[u].x);
                add((double)a[u].y, (double)b[u].y);
                add((double)a[u].z, (double)b[u].z);
                add((double)a[u].w, (double)b[u].w);
            }
        }
        if (first) { px = (double)x[i]; py = (double)y[i]; } // fewer than ASX_PEARSON_DEEP whole steps: the step loop below meets it first
    }
#endif
    for (; i + 3 < hi; i += 4u * ASX_THREADS) {
        const vec4u a = (ASX_NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const vec4u *>(x + i)) : *reinterpret_cast<const vec4u *>(x + i);
        const vec4u b = (ASX_NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const vec4u *>(y + i)) : *reinterpret_cast<const vec4u *>(y + i);
        add((double)a.x, (double)b.x);
        add((double)a.y, (double)b.y);
        add((double)a.z, (double)b.z);
        add((double)a.w, (double)b.w);
    }
    for (; i < hi; i++) add((double)x[i], (double)y[i]); // only the lane that holds the ragged end
    PStat v;
    v.n = (double)cnt;
    v.mx = v.my = v.mxx = v.myy = v.cxy = 0.0;
    if (cnt) {
        v.mx = px + sx / v.n;
        v.my = py + sy / v.n;
        v.mxx = sxx - sx * sx / v.n;
        v.myy = syy - sy * sy / v.n;
        v.cxy = sxy - sx * sy / v.n;
    }
    v = pstat_wave_merge(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = v.n; red[1][wave] = v.mx; red[2][wave] = v.my;
        red[3][wave] = v.mxx; red[4][wave] = v.myy; red[5][wave] = v.cxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ASX_THREADS / 64; w++) {
            PStat o;
            o.n = red[0][w]; o.mx = red[1][w]; o.my = red[2][w];
            o.mxx = red[3][w]; o.myy = red[4][w]; o.cxy = red[5][w];
            v = pstat_merge(v, o);
        }
        double *out = psums + (pair * gridDim.x + blockIdx.x) * 6;
        out[0] = v.n; out[1] = v.mx; out[2] = v.my; out[3] = v.mxx; out[4] = v.myy; out[5] = v.cxy;
    }
}

// grid (npairs), one wave per pair: lane b owns the partial blocks b, b + 64, ... (merged in that order: a fixed tree for a
// given block count, the same for x and y).
// (Pick, partial sums and this merge as ONE kernel -- the last block to arrive at a per-pair counter merges -- was
// measured: the release-acquire increment per block took the batch pass from 0.20 to 0.60 ms and the single pair
// gained nothing, its latency is the chain of dependent loads inside the kernels, not their number.)
__global__ __launch_bounds__(64) void k_pearson_final(const AsxSeg *__restrict__ seg,
                                                       const double *__restrict__ psums, uint32_t nb,
                                                       int64_t *__restrict__ lag,
                                                       double *__restrict__ coef,
                                                       int32_t *__restrict__ ret)
{
    const size_t pair = blockIdx.x;
    PStat v;
    v.n = v.mx = v.my = v.mxx = v.myy = v.cxy = 0.0;
    for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
        const double *p = psums + (pair * nb + b) * 6;
        PStat o;
        o.n = p[0]; o.mx = p[1]; o.my = p[2]; o.mxx = p[3]; o.myy = p[4]; o.cxy = p[5];
        v = pstat_merge(v, o);
    }
    v = pstat_wave_merge(v);
    if (threadIdx.x == 0) {
        const AsxSeg s = seg[pair];
        // src/cross_correlation.c:115; an empty or constant segment gives 0/0 = NaN like the reference
        const double c = v.cxy / sqrt(v.mxx * v.myy);
        if (lag) lag[pair] = s.lag;
        coef[pair] = c;
        // src/cross_correlation.c:276: NaN coefficient -> return -1 (outputs already written)
        // ... unless the lag itself is still the float32 placeholder of an overflowed pair: 1 = "inexact, look again"
        // (only ever visible to callers of the asynchronous mode, include/audiosync/xcorr_hip.h)
        if (ret) ret[pair] = (s.flags & ASX_SEG_INEXACT) ? 1 : (c != c) ? -1 : 0;
    }
}

// ---------------------------------------------------------------------------
// result consumers: the acceptance threshold and frames -> milliseconds of
// src/audiosync.c:254-256 for a whole batch
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(ASX_THREADS) void k_results_to_ms(const int64_t *__restrict__ lag,
                                                                const double *__restrict__ coef,
                                                                const int32_t *__restrict__ ret, size_t batch,
                                                                double min_confidence, double ms_per_frame,
                                                                int64_t *__restrict__ lag_ms,
                                                                int32_t *__restrict__ accept)
{
    const size_t i = (size_t)blockIdx.x * ASX_THREADS + threadIdx.x;
    if (i >= batch) return;
    lag_ms[i] = (int64_t)round((double)lag[i] * ms_per_frame);
    if (accept) accept[i] = (ret[i] == 0 && coef[i] >= min_confidence) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// second look, DC removal (second_look, asx_api.hip): one pair, one block each.  An offset of hundreds of standard deviations
// in BOTH tracks makes the float32 error bound (proportional to |source|_2 |sample|_2) wider than the whole range of
// r, every lag a near-tie.  r'[k] = sum (source[n+k] - m) sample[n] = r[k] - m * sum(sample) for ANY constant m: the
// transforms then run on a zero-mean source, whose norm no longer carries the offset, and the constant goes back in
// when the keys are formed (peak_key_shifted).  The reference needs none of this: float64 (src/cross_correlation.c:34).
// ---------------------------------------------------------------------------
// grid (ASX_DC_BLOCKS): per-block sums of the source (2N) and the sample (N) in float64, 16-byte loads; the last
// kernel (one wave) adds the partials in block order -- a fixed tree, the same for every call
#define ASX_DC_BLOCKS 128
template <typename TIn>
__global__ __launch_bounds__(ASX_THREADS) void k_dc_partial(const TIn *__restrict__ src, const TIn *__restrict__ smp, uint32_t N,
                                                             double *__restrict__ part)
{
    __shared__ double red[2][ASX_THREADS / 64];
    typedef TIn vec4u __attribute__((ext_vector_type(4), aligned(sizeof(TIn))));
    double a = 0.0, b = 0.0;
    const size_t stride = (size_t)gridDim.x * ASX_THREADS * 4, first = ((size_t)blockIdx.x * ASX_THREADS + threadIdx.x) * 4;
    const size_t n2 = 2 * (size_t)N;
    for (size_t i = first; i < n2; i += stride) {
        if (i + 3 < n2) { const vec4u v = *reinterpret_cast<const vec4u *>(src + i); a += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w); }
        else for (size_t k = i; k < n2; k++) a += (double)src[k];
    }
    for (size_t i = first; i < N; i += stride) {
        if (i + 3 < N) { const vec4u v = *reinterpret_cast<const vec4u *>(smp + i); b += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w); }
        else for (size_t k = i; k < N; k++) b += (double)smp[k];
    }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ASX_THREADS / 64; w++) { a += red[0][w]; b += red[1][w]; }
        part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b;
    }
}
__global__ __launch_bounds__(64) void k_dc_final(const double *__restrict__ part, uint32_t N, double scale, double *__restrict__ stats)
{
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < ASX_DC_BLOCKS; i += 64) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a); b = wave_sum(b);
    if (threadIdx.x == 0) {
        const double mean = a / (2.0 * (double)N);
        // the device's r is the unnormalised inverse transform, F times the plain sum of products (asx_api.hip,
        // bound_scale): the constant that goes back into the keys carries the same factor
        stats[0] = mean; stats[1] = b; stats[2] = mean * b * scale;
    }
}
template <typename TIn>
__global__ __launch_bounds__(ASX_THREADS) void k_dc_apply(const TIn *__restrict__ src, uint32_t N, const double *__restrict__ stats,
                                                           float *__restrict__ out)
{
    const double mean = stats[0];
    for (size_t i = (size_t)blockIdx.x * ASX_THREADS + threadIdx.x; i < 2 * (size_t)N; i += (size_t)gridDim.x * ASX_THREADS)
        out[i] = (float)((double)src[i] - mean);
}

// ---------------------------------------------------------------------------
// double -> float conversion of the reference's f64 buffers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(ASX_THREADS) void k_cvt_f64_f32(const double *__restrict__ in,
                                                              float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * ASX_THREADS + threadIdx.x; i < n;
         i += (size_t)gridDim.x * ASX_THREADS)
        out[i] = (float)in[i];
}

// ---------------------------------------------------------------------------
// synthetic pairs: bit-identical to oracle_synth_pair (oracle/xcorr_oracle.c)
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t pair, uint64_t stream)
{
    return mix64(mix64(seed) + 0x632BE59BD9B4E019ull * (pair + 1) + stream);
}
__device__ __forceinline__ float u24(uint64_t h)
{
    const int32_t v = (int32_t)(h >> 40) - (1 << 23);
    return (float)v * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float z22(uint64_t key, uint64_t n)
{
    const uint64_t h1 = mix64(key + 2 * n), h2 = mix64(key + 2 * n + 1);
    const int32_t a = (int32_t)(h1 >> 42), b = (int32_t)((h1 >> 20) & 0x3FFFFF);
    const int32_t c = (int32_t)(h2 >> 42), d = (int32_t)((h2 >> 20) & 0x3FFFFF);
    const int32_t v = a + b + c + d - (1 << 23);
    return (float)v * (1.0f / 4194304.0f);
}

__global__ __launch_bounds__(ASX_THREADS) void k_synth(uint64_t seed, uint64_t first_pair, uint32_t N,
                                                        float amp, float *__restrict__ src,
                                                        float *__restrict__ smp,
                                                        int64_t *__restrict__ true_lag)
{
    const uint64_t pair = first_pair + blockIdx.y;
    const uint64_t ks = stream_key(seed, pair, 1), kn = stream_key(seed, pair, 2);
    const uint64_t kl = stream_key(seed, pair, 3);
    const int64_t span = (int64_t)(3ull * N / 4);
    const int64_t lag = (int64_t)(mix64(kl) % (uint64_t)(2 * span + 1)) - span;
    float *s = src + (size_t)blockIdx.y * 2 * N;
    float *t = smp + (size_t)blockIdx.y * N;
    for (uint32_t i = blockIdx.x * ASX_THREADS + threadIdx.x; i < 2u * N; i += gridDim.x * ASX_THREADS) {
        s[i] = u24(mix64(ks + N + i));
        if (i < N) {
            const float sig = 0.5f * u24(mix64(ks + (uint64_t)((int64_t)N + (int64_t)i + lag)));
            t[i] = __fadd_rn(sig, __fmul_rn(amp, z22(kn, i)));
        }
    }
    if (true_lag && blockIdx.x == 0 && threadIdx.x == 0) true_lag[blockIdx.y] = lag;
}

#endif // ASX_HAS_PART(64)

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static size_t lds_bytes_cols(const AsxDev &P) { return (size_t)P.M1 * P.T * sizeof(float2); }
static size_t lds_bytes_rows(const AsxDev &P) { return (size_t)4 * P.M2 * sizeof(float2); }

static int max_radix(const AsxStages &st)
{
    int m = 2;
    for (int i = 0; i < st.nstages; i++) m = st.radix[i] > m ? st.radix[i] : m;
    return m;
}

static void allow_big_lds(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static bool schedule_is(const AsxStages &st, int n, std::initializer_list<int> radices)
{
    if (st.n != n || st.nstages != (int)radices.size()) return false;
    int i = 0;
    for (int r : radices)
        if (st.radix[i++] != r) return false;
    return true;
}
// Column schedules of the production sample lengths (plan_math.cpp's tuned table), compiled in:
//   X(M1, tile width, block size, MAXR for the launch bounds, radices...)
#define ASX_STATIC_COLS(X) \
    X(1200, 8, 512, 12, 12, 10, 10) X(800, 8, 320, 10, 10, 10, 8) X(600, 16, 512, 10, 10, 10, 6) \
    X(400, 16, 320, 10, 10, 8, 5) X(300, 16, 256, 10, 10, 6, 5)

// Every transform kernel has two launchers, each in its own part of the build: *_static returns false
// when no compiled-in schedule matches the plan, *_generic takes any plan.
bool asx_launch_fwd_cols_static(const AsxDev &P, const float *src, const float *smp, float2 *zxa, float2 *zya,
                                const AsxPeakWs &W, int npairs, hipStream_t s);
void asx_launch_fwd_cols_generic(const AsxDev &P, const float *src, const float *smp, float2 *zxa, float2 *zya,
                                 const AsxPeakWs &W, int npairs, hipStream_t s);
bool asx_launch_rows_static(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga, const AsxPeakWs &W,
                            int npairs, hipStream_t s);
void asx_launch_rows_generic(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga, const AsxPeakWs &W,
                             int npairs, hipStream_t s);
bool asx_launch_inv_cols_static(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out, int npairs,
                                hipStream_t s);
void asx_launch_inv_cols_generic(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out, int npairs,
                                 hipStream_t s);

#define ASX_FWD_LAUNCH(...) \
    do { allow_big_lds((const void *)k_fwd_cols<__VA_ARGS__>, lds_bytes_cols(P)); \
         hipLaunchKernelGGL((k_fwd_cols<__VA_ARGS__>), grid, dim3(P.threads_cols), lds_bytes_cols(P), s, P.self_dev, src, smp, zxa, zya, W.nrm_part); } while (0)
#if ASX_HAS_PART(1)
bool asx_launch_fwd_cols_static(const AsxDev &P, const float *src, const float *smp, float2 *zxa, float2 *zya,
                                const AsxPeakWs &W, int npairs, hipStream_t s)
{
    dim3 grid(col_grid_x(P.ntiles, P.logT), 2, npairs);
#define ASX_TRY_STATIC(m1, t, nt, maxr, ...) \
    if (P.T == (t) && P.threads_cols == (nt) && schedule_is(P.st1, m1, { __VA_ARGS__ })) { ASX_FWD_LAUNCH(maxr, Sched<m1, __VA_ARGS__>, t, nt); return true; }
    ASX_STATIC_COLS(ASX_TRY_STATIC)
#undef ASX_TRY_STATIC
    return false;
}
#endif
#if ASX_HAS_PART(2)
void asx_launch_fwd_cols_generic(const AsxDev &P, const float *src, const float *smp, float2 *zxa, float2 *zya,
                                 const AsxPeakWs &W, int npairs, hipStream_t s)
{
    dim3 grid(col_grid_x(P.ntiles, P.logT), 2, npairs);
    const int mr = max_radix(P.st1);
    if (mr <= 10) ASX_FWD_LAUNCH(10); else if (mr <= 12) ASX_FWD_LAUNCH(12); else ASX_FWD_LAUNCH(16);
}
#endif
#undef ASX_FWD_LAUNCH

#define ASX_ROWS_LAUNCH(...) \
    do { allow_big_lds((const void *)k_rows<__VA_ARGS__>, lds); \
         hipLaunchKernelGGL((k_rows<__VA_ARGS__>), dim3(ntasks), dim3(P.threads_rows), lds, s, P.self_dev, zxa, zya, ga, P.row_tasks, P.M1, P.M2, P.M, W); } while (0)
static size_t rows_lds_request(const AsxDev &P)
{
    size_t lds = lds_bytes_rows(P);
    // diagnostic: a larger request caps the blocks per CU (occupancy sweep, tools/README.md)
    if (const char *e = getenv("ASX_DBG_ROWS_LDS")) lds = std::max(lds, (size_t)atol(e));
    return lds;
}
#if ASX_HAS_PART(4)
// row lengths of the production sample lengths (plan_math.cpp's tuned table): schedule compiled in
bool asx_launch_rows_static(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga, const AsxPeakWs &W,
                            int npairs, hipStream_t s)
{
    const int ntasks = (P.M1 / 2 + 1) * npairs; // one task (pair, k1) per block
    const size_t lds = rows_lds_request(P);
    if (P.threads_rows == 256 && schedule_is(P.st2, 1200, { 12, 10, 10 })) { ASX_ROWS_LAUNCH(12, Sched<1200, 12, 10, 10>, 256); return true; }
    if (P.threads_rows == 128 && schedule_is(P.st2, 480, { 10, 8, 6 })) { ASX_ROWS_LAUNCH(10, Sched<480, 10, 8, 6>, 128); return true; }
    return false;
}
#endif
#if ASX_HAS_PART(8)
void asx_launch_rows_generic(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga, const AsxPeakWs &W,
                             int npairs, hipStream_t s)
{
    const int ntasks = (P.M1 / 2 + 1) * npairs;
    const size_t lds = rows_lds_request(P);
    const int mr = max_radix(P.st2);
    if (mr <= 10) ASX_ROWS_LAUNCH(10); else if (mr <= 12) ASX_ROWS_LAUNCH(12); else ASX_ROWS_LAUNCH(16);
}
#endif
#undef ASX_ROWS_LAUNCH

#define ASX_INV_LAUNCH(...) \
    do { allow_big_lds((const void *)k_inv_cols<__VA_ARGS__>, lds_bytes_cols(P)); \
         hipLaunchKernelGGL((k_inv_cols<__VA_ARGS__>), grid, dim3(P.threads_cols), lds_bytes_cols(P), s, P.self_dev, ga, W, r_out); } while (0)
#if ASX_HAS_PART(16)
bool asx_launch_inv_cols_static(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out, int npairs,
                                hipStream_t s)
{
    dim3 grid(col_grid_x(P.ntiles, P.logT), npairs);
#define ASX_TRY_STATIC(m1, t, nt, maxr, ...) \
    if (P.T == (t) && P.threads_cols == (nt) && schedule_is(P.st1, m1, { __VA_ARGS__ })) { ASX_INV_LAUNCH(maxr, Sched<m1, __VA_ARGS__>, t, nt); return true; }
    ASX_STATIC_COLS(ASX_TRY_STATIC)
#undef ASX_TRY_STATIC
    return false;
}
#endif
#if ASX_HAS_PART(32)
void asx_launch_inv_cols_generic(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out, int npairs,
                                 hipStream_t s)
{
    dim3 grid(col_grid_x(P.ntiles, P.logT), npairs);
    const int mr = max_radix(P.st1);
    if (mr <= 10) ASX_INV_LAUNCH(10); else if (mr <= 12) ASX_INV_LAUNCH(12); else ASX_INV_LAUNCH(16);
}
#endif
#undef ASX_INV_LAUNCH

#if ASX_HAS_PART(64)
size_t asx_lds_bytes_cols(const AsxDev &P) { return lds_bytes_cols(P); }
size_t asx_lds_bytes_rows(const AsxDev &P) { return lds_bytes_rows(P); }

// Block size: a multiple of 64 (<= ASX_FFT_THREADS_MAX).  First enough waves per CU to hide
// LDS/HBM latency given how many blocks the LDS footprint admits (target >= 12 waves per CU),
// then the size that wastes the fewest thread slots over the stages (each stage has
// `groups * n / radix` work items); ties go to the larger block.
int asx_pick_threads(const AsxStages &st, int groups, int min_threads, size_t lds_bytes)
{
    long blocks_per_cu = lds_bytes ? (long)(160 * 1024 / lds_bytes) : 8;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu > 8) blocks_per_cu = 8;
    int want = 64 * (int)((12 + blocks_per_cu - 1) / blocks_per_cu);
    if (want > ASX_FFT_THREADS_MAX) want = ASX_FFT_THREADS_MAX;
    if (min_threads < want) min_threads = want;
    long best_cost = -1;
    int best = ASX_FFT_THREADS_MAX;
    for (int bd = 64; bd <= ASX_FFT_THREADS_MAX; bd += 64) {
        if (bd < min_threads) continue;
        long cost = 0;
        for (int i = 0; i < st.nstages; i++) {
            const long items = (long)groups * st.nbf[i];
            cost += (items + bd - 1) / bd * bd;
        }
        if (st.nstages == 0) cost = bd;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = bd; }
    }
    return best;
}

static bool generic_only()
{
    static const bool g = getenv("ASX_GENERIC") != nullptr; // diagnostic: never use the compiled-in schedules
    return g;
}

void asx_launch_fwd_cols(const AsxDev &P, const float *src, const float *smp, float2 *zxa,
                         float2 *zya, const AsxPeakWs &W, int npairs, hipStream_t s)
{
    if (P.rlayout && asx_launch_fwd_cols_r(P, src, smp, zxa, zya, W, npairs, s)) return;
    if (generic_only() || !asx_launch_fwd_cols_static(P, src, smp, zxa, zya, W, npairs, s))
        asx_launch_fwd_cols_generic(P, src, smp, zxa, zya, W, npairs, s);
}

void asx_launch_rows(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga,
                     const AsxPeakWs &W, int npairs, hipStream_t s)
{
    if (P.rlayout && asx_launch_rows_r(P, zxa, zya, ga, W, npairs, s)) return;
    if (generic_only() || !asx_launch_rows_static(P, zxa, zya, ga, W, npairs, s))
        asx_launch_rows_generic(P, zxa, zya, ga, W, npairs, s);
}

void asx_launch_inv_cols(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out, int npairs,
                         hipStream_t s)
{
    if (P.rlayout && asx_launch_inv_cols_r(P, ga, W, r_out, npairs, s)) return;
    if (generic_only() || !asx_launch_inv_cols_static(P, ga, W, r_out, npairs, s))
        asx_launch_inv_cols_generic(P, ga, W, r_out, npairs, s);
}

void asx_launch_finalize(const AsxDev &P, const AsxPeakWs &W, AsxSeg *seg, int npairs, hipStream_t s, uint32_t pair_base)
{
    hipLaunchKernelGGL(k_finalize, dim3(npairs), dim3(ASX_THREADS), 0, s, P.self_dev, W, seg, pair_base);
}

void asx_launch_refine_f32(const AsxDev &P, const float *src, const float *smp, const AsxPeakWs &W,
                           AsxSeg *seg, int npairs, hipStream_t s, int dot_blocks, bool pick)
{
    hipLaunchKernelGGL(k_refine_dots<float>, dim3(dot_blocks, npairs), dim3(ASX_THREADS), 0, s, P.self_dev, src, smp, W);
    if (pick) hipLaunchKernelGGL(k_refine_pick, dim3(npairs), dim3(ASX_THREADS), 0, s, P.self_dev, W, seg);
}

void asx_launch_refine_f64(const AsxDev &P, const double *src, const double *smp, const AsxPeakWs &W,
                           AsxSeg *seg, int npairs, hipStream_t s, int dot_blocks)
{
    hipLaunchKernelGGL(k_refine_dots<double>, dim3(dot_blocks, npairs), dim3(ASX_THREADS), 0, s, P.self_dev, src, smp, W);
    hipLaunchKernelGGL(k_refine_pick, dim3(npairs), dim3(ASX_THREADS), 0, s, P.self_dev, W, seg);
}

// Partial blocks per pair: a function of the segment's BASIS LENGTH ONLY -- one block per 16 sweeps of 256 threads x 4
// elements (a block that only makes two sweeps pays more for its reduction than for its loads: 1024 pairs of
// N = 144 000 ran at 3.0 TB/s with 64 blocks per pair against 5.7 TB/s at N = 1 440 000), at most
// ASX_PEARSON_BLOCKS_MAX.  The chunks a pair is cut into, the pivots and the merge tree -- and so the last bits of its
// coefficient -- are therefore the same whatever the batch it travels in, alone through cross_correlation(), in a group
// of 124, re-run by the second look or on another shard (round 3 derived the count from the launch size).
unsigned asx_pearson_blocks(uint32_t basis_len)
{
    const uint32_t per_block = 16u * 4u * ASX_THREADS;
    uint32_t nb = (basis_len + per_block - 1) / per_block;
    if (nb > ASX_PEARSON_BLOCKS_MAX) nb = ASX_PEARSON_BLOCKS_MAX;
    if (nb < 1) nb = 1;
    return nb;
}

void asx_launch_pearson_f32(const float *src, const float *smp, size_t src_pitch, size_t smp_pitch,
                            uint32_t basis_len, const AsxSeg *seg, double *psums, int64_t *lag,
                            double *coef, int32_t *ret, int npairs, hipStream_t s)
{
    const unsigned nb = asx_pearson_blocks(basis_len);
    hipLaunchKernelGGL((k_pearson_partial<float, false>), dim3(nb, npairs), dim3(ASX_THREADS), 0, s,
                       src, smp, src_pitch, smp_pitch, basis_len, seg, psums, AsxSpecWs{});
    hipLaunchKernelGGL(k_pearson_final, dim3(npairs), dim3(64), 0, s, seg, psums, nb, lag, coef, ret);
}

void asx_launch_pearson_partial_spec_f32(const float *src, const float *smp, size_t src_pitch, size_t smp_pitch, uint32_t basis_len,
                                         const AsxSeg *seg, const AsxSpecWs &S, double *psums, int npairs, hipStream_t s)
{
    const unsigned nb = asx_pearson_blocks(basis_len);
    hipLaunchKernelGGL((k_pearson_partial<float, true>), dim3(nb, npairs), dim3(ASX_THREADS), 0, s,
                       src, smp, src_pitch, smp_pitch, basis_len, seg, psums, S);
}

void asx_launch_pearson_f64(const double *src, const double *smp, size_t src_pitch, size_t smp_pitch,
                            uint32_t basis_len, const AsxSeg *seg, double *psums, int64_t *lag,
                            double *coef, int32_t *ret, int npairs, hipStream_t s)
{
    const unsigned nb = asx_pearson_blocks(basis_len);
    hipLaunchKernelGGL((k_pearson_partial<double, false>), dim3(nb, npairs), dim3(ASX_THREADS), 0, s,
                       src, smp, src_pitch, smp_pitch, basis_len, seg, psums, AsxSpecWs{});
    hipLaunchKernelGGL(k_pearson_final, dim3(npairs), dim3(64), 0, s, seg, psums, nb, lag, coef, ret);
}

void asx_launch_results_to_ms(const int64_t *lag, const double *coef, const int32_t *ret, size_t batch,
                              double min_confidence, double sample_rate, int64_t *lag_ms, int32_t *accept,
                              hipStream_t s)
{
    if (batch == 0) return;
    const unsigned blocks = (unsigned)((batch + ASX_THREADS - 1) / ASX_THREADS);
    hipLaunchKernelGGL(k_results_to_ms, dim3(blocks), dim3(ASX_THREADS), 0, s, lag, coef, ret, batch, min_confidence,
                       1000.0 / sample_rate, lag_ms, accept);
}

void asx_launch_cvt_f64_f32(const double *in, float *out, size_t n, hipStream_t s)
{
    size_t blocks = (n + ASX_THREADS - 1) / ASX_THREADS;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_cvt_f64_f32, dim3((unsigned)blocks), dim3(ASX_THREADS), 0, s, in, out, n);
}

void asx_launch_dc_remove_f32(const float *src, const float *smp, uint32_t N, double scale, double *stats, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_dc_partial<float>, dim3(ASX_DC_BLOCKS), dim3(ASX_THREADS), 0, s, src, smp, N, stats + 4);
    hipLaunchKernelGGL(k_dc_final, dim3(1), dim3(64), 0, s, stats + 4, N, scale, stats);
    hipLaunchKernelGGL(k_dc_apply<float>, dim3(512), dim3(ASX_THREADS), 0, s, src, N, stats, out);
}
void asx_launch_dc_remove_f64(const double *src, const double *smp, uint32_t N, double scale, double *stats, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_dc_partial<double>, dim3(ASX_DC_BLOCKS), dim3(ASX_THREADS), 0, s, src, smp, N, stats + 4);
    hipLaunchKernelGGL(k_dc_final, dim3(1), dim3(64), 0, s, stats + 4, N, scale, stats);
    hipLaunchKernelGGL(k_dc_apply<double>, dim3(512), dim3(ASX_THREADS), 0, s, src, N, stats, out);
}

void asx_launch_synth(uint64_t seed, uint64_t first_pair, size_t count, uint32_t N, int noise_shift,
                      float *src, float *smp, int64_t *true_lag, hipStream_t s)
{
    unsigned bx = (2u * N + ASX_THREADS - 1) / ASX_THREADS;
    if (bx > 1024) bx = 1024;
    const float amp = ldexpf(1.0f, -noise_shift);
    hipLaunchKernelGGL(k_synth, dim3(bx, (unsigned)count), dim3(ASX_THREADS), 0, s, seed, first_pair, N,
                       amp, src, smp, true_lag);
}
#endif // ASX_HAS_PART(64)
