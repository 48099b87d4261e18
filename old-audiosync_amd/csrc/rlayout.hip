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
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) float2 asx_lds_r[];

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

#ifndef ASX_ROWSR_WAVES
#define ASX_ROWSR_WAVES 4 // waves per SIMD the register allocation must allow: 8 blocks of 2 waves per CU
#endif

// ---------------------------------------------------------------------------
// k_rows_r: grid ((M1 + 1) * npairs).  One block = row k1 of both spectra of one pair.
//   S = Sched<M2, R0, R1, R2>: three DIF stages.  Stage 0 spans the row (block-wide, barrier after it); stages 1
//   and 2 stay inside the R0 sub-blocks of length M2/R0 ("units"), and so do the first two stages of the inverse:
//   a wave that owns whole units runs  forward 1 -> forward 2 -> product -> inverse 2 -> inverse 1  on them with no
//   block barrier in between.  Inverse stage 0 spans the row again; its outputs leave from registers.
// ---------------------------------------------------------------------------
template <class S, int NT>
__global__ __launch_bounds__(NT, ASX_ROWSR_WAVES) void k_rows_r(const AsxDev *__restrict__ Pp, const float2 *__restrict__ cx,
                                                                 const float2 *__restrict__ cy, float2 *__restrict__ qo,
                                                                 int nrows, size_t pair_pitch, AsxPeakWs W)
{
    static_assert(S::nstages == 3, "three-stage row schedules only");
    constexpr int M2 = S::n;
    constexpr StageK K0 = S::stage(0), K1 = S::stage(1), K2 = S::stage(2);
    constexpr int R0 = K0.R, R1 = K1.R, R2 = K2.R;
    constexpr int Q0 = K0.q;  // butterflies of stage 0 = length of a unit
    constexpr int UN = K1.ns; // = Q0
    static_assert(UN == R1 * R2 && Q0 == UN && K2.q == 1, "unit = R1 x R2");
    static_assert((M2 & 1) == 0, "rows move as 16 bytes per lane");
    constexpr int HALF = M2 / 2, WSTEPS = (HALF + NT - 1) / NT;
    constexpr int LPU = R1 > R2 ? R1 : R2, UPW = 64 / LPU, NW = NT / 64;
    static_assert(NT % 64 == 0 && UPW >= 1, "whole waves");

    const AsxDev &PD = *Pp;
    const AsxKP P = asx_kp(PD);
    float4 *A4 = reinterpret_cast<float4 *>(asx_lds_r);
    __shared__ float2 tw_step[WSTEPS]; // w_F^(k1 * 2*NT*i): the four-step twiddle from load step to load step
    __shared__ float2 leg[R0];         // w_F^(k1 * Q0*t): ... and from leg to leg of the last inverse stage

    const int task = blockIdx.x, tid = threadIdx.x;
    const int pair = task / nrows;
    const uint32_t k1 = (uint32_t)(task - pair * nrows);
    const size_t row = (size_t)pair * pair_pitch + (size_t)k1 * M2;
    if (k1 == 0 && tid < 64) {
        // Row 0 of a pair also prepares the pair's peak search (k_inv_cols_r runs after this kernel): the float32
        // error bound from the norms k_fwd_cols_r left, the running maximum and the candidate count back to zero.
        const float *np = W.nrm_part + (size_t)pair * 2 * P.ntiles;
        float sx = 0.f, sy = 0.f;
        for (int t = tid; t < P.ntiles; t += 64) { sx += np[t]; sy += np[P.ntiles + t]; }
        sx = wave_sum_f32(sx); sy = wave_sum_f32(sy);
        if (tid == 0) {
            W.bound2[pair] = PD.bound_scale * sqrtf(sx) * sqrtf(sy);
            W.pairmax[pair] = 0;
            W.cand_n[pair] = 0;
        }
    }
    // every row load of the thread first; the twiddle lookups below overlap them
    float4 lx[WSTEPS], ly[WSTEPS];
    {
        const float2 *gx = cx + row, *gy = cy + row;
        static_for<0, WSTEPS>([&](auto I) __attribute__((always_inline)) {
            const int q = tid + decltype(I)::value * NT;
            if (q < HALF) {
                lx[I] = *reinterpret_cast<const float4 *>(gx + 2 * q);
                ly[I] = *reinterpret_cast<const float4 *>(gy + 2 * q);
            }
        });
    }
    if (tid < WSTEPS) tw_step[tid] = tw_F(P, k1 * (uint32_t)(2 * NT * tid));
    if (tid >= NT - R0) { const int t = tid - (NT - R0); leg[t] = tw_F(P, k1 * (uint32_t)(Q0 * t)); }
    const float2 twa = tw_F(P, k1 * (uint32_t)(2 * tid < M2 ? 2 * tid : 0)); // w_F^(k1 * 2 tid)
    const float2 wk1 = tw_F(P, k1);
    // stage twiddle seeds (they depend on the thread only): stage 0 / inverse stage 0, and the wave-local stage 1
    const int j0 = tid < Q0 ? tid : 0;
    const float2 s0w1 = P.tw2[j0], s0w4 = P.tw2[4 * j0];
    const float2 tw0base = tw_F(P, k1 * (uint32_t)j0); // four-step factor of the thread's outputs of inverse stage 0
    const int lane = tid & 63, wave = tid >> 6;
    const int ul = lane / LPU, jl = lane - ul * LPU;
    const int j1c = jl < R2 ? jl : 0;
    const float2 s1w1 = P.tw2[K1.twmul * j1c], s1w4 = P.tw2[(R1 > 4 ? 4 : 1) * K1.twmul * j1c];
    __syncthreads(); // tw_step, leg
    static_for<0, WSTEPS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int q = tid + i * NT;
        if (q < HALF) {
            const float2 wa0 = cmul(twa, tw_step[i]), wa1 = cmul(wa0, wk1);
            lds_put(A4 + 2 * q, mulw(Cx2{ v2f{ lx[I].x, ly[I].x }, v2f{ lx[I].y, ly[I].y } }, wa0));
            lds_put(A4 + 2 * q + 1, mulw(Cx2{ v2f{ lx[I].z, ly[I].z }, v2f{ lx[I].w, ly[I].w } }, wa1));
        }
    });
    __syncthreads();

    // ---- forward stage 0: butterflies j < Q0, legs Q0 apart ------------------------------------------------
    for (int j = tid; j < Q0; j += NT) {
        float4 *p = A4 + j;
        Cx2 v[R0];
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + decltype(T)::value * Q0); });
        float2 w1 = s0w1, w4 = s0w4;
        if (j != tid) { w1 = P.tw2[j]; w4 = P.tw2[4 * j]; }
        float2 tww[R0];
        stage_twiddles_from<R0>(w1, w4, tww);
        Bfly<R0, false>::run(v);
        static_for<1, R0>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tww[U]); });
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) { lds_put(p + decltype(T)::value * Q0, v[T]); });
    }
    __syncthreads();

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

    // ---- inverse stage 0 from LDS, conjugate four-step twiddle, straight to HBM ----------------------------
    float2 *go = qo + row;
    for (int j = tid; j < Q0; j += NT) {
        const float4 *p = A4 + j;
        Cx1 v[R0];
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get1(p + decltype(T)::value * Q0); });
        float2 w1 = s0w1, w4 = s0w4, fb = tw0base;
        if (j != tid) { w1 = P.tw2[j]; w4 = P.tw2[4 * j]; fb = tw_F(P, k1 * (uint32_t)j); }
        float2 tww[R0];
        stage_twiddles_from<R0>(w1, w4, tww);
        static_for<1, R0>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tww[U]); });
        Bfly<R0, true>::run(v);
        static_for<0, R0>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            const Cx1 y = mulwc(v[t], t == 0 ? fb : cmul(fb, leg[t]));
            go[j + t * Q0] = make_float2(y.re, y.im);
        });
    }
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
#define ASX_ROWSR_CASE(nt, n, ...)                                                                                              \
    if (schedule_is_r(P.st2, n, { __VA_ARGS__ })) {                                                                             \
        hipLaunchKernelGGL((k_rows_r<Sched<n, __VA_ARGS__>, nt>), dim3((unsigned)nrows * (unsigned)npairs), dim3(nt), lds, s,   \
                           P.self_dev, cx, cy, q, nrows, pitch, W);                                                             \
        return true;                                                                                                            \
    }
    ASX_ROWSR_CASE(128, 1200, 12, 10, 10)
    ASX_ROWSR_CASE(128, 480, 10, 8, 6)
#undef ASX_ROWSR_CASE
    return false;
}
