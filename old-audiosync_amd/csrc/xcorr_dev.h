// csrc/xcorr_dev.h -- device helpers shared by the kernel translation units (xcorr_kernels.hip, rows2.hip).
#pragma once

#include "asx_internal.h"
#include "lds_fft.h"

// The scalars and table pointers of the plan a kernel uses, copied into registers ONCE at its
// start.  Read through the plan pointer where they are used, every use after a barrier is another
// scalar load plus a wait that also drains the LDS queue (the spectral combine of k_rows alone
// re-read two pointers in each of its five steps).
struct AsxKP {
    const float2 *tw1, *tw2, *tw2s, *tw_lo, *tw_hi;
    const int *pos2_of_k2;
    uint32_t N, F, M, nout, src_valid, src_period;
    int M1, M2, T, logT, ntiles;
    unsigned long long *stamps;
    int stamp_kernel;
};
__device__ __forceinline__ AsxKP asx_kp(const AsxDev &D)
{
    AsxKP k;
    k.tw1 = D.tw1; k.tw2 = D.tw2; k.tw2s = D.tw2s; k.tw_lo = D.tw_lo; k.tw_hi = D.tw_hi;
    k.pos2_of_k2 = D.pos2_of_k2;
    k.N = D.N; k.F = D.F; k.M = D.M; k.nout = D.nout; k.src_valid = D.src_valid; k.src_period = D.src_period;
    k.M1 = D.M1; k.M2 = D.M2; k.T = D.T; k.logT = D.logT; k.ntiles = D.ntiles;
    k.stamps = D.stamps; k.stamp_kernel = D.stamp_kernel;
    return k;
}

// w_F^p for p < F from the two-level table (one complex multiply, ~1.5e-7 accurate).
__device__ __forceinline__ float2 tw_F(const AsxKP &P, uint32_t p)
{
    const float2 lo = P.tw_lo[p & (ASX_TW_LO - 1u)];
    const float2 hi = P.tw_hi[p >> ASX_TW_LOG];
    return cmul(lo, hi);
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Spectral combine of one pair of bins (src/cross_correlation.c:232-233 between the real-FFT untangling of the
// forward transforms and the tangling of the inverse one).
__device__ __forceinline__ void combine_pair(Cx2 Za, Cx2 Zb, float2 w2, float2 &Gk, float2 &Gm)
{
    // Za = (Zx[k], Zy[k]), Zb = (Zx[M-k], Zy[M-k]); w2 = w_M^k.  With E' = a + conj b and
    // O' = -i (a - conj b) (twice the even/odd parts of the real-FFT untangling), X[k] = (E'x + w O'x)/2
    // etc.  Expanding P = X conj(Y) (src/cross_correlation.c:232-233) for k and M-k and the inverse
    // tangling G[k] = (P[k] + conj P[M-k]) + i conj(w)(P[k] - conj P[M-k]) collapses to
    //     W = E'x conj(E'y) + O'x conj(O'y),   U = O'x conj(E'y) + conj(w^2) E'x conj(O'y)
    //     G[k] = (W + i U)/2,   G[M-k] = (conj W + i conj U)/2
    // (checked against the step-by-step form in tests/model_fourstep.py): 40 real operations
    // per pair of bins instead of 60, and X, Y, P never exist.
    const Cx2 E = Cx2{ Za.re + Zb.re, Za.im - Zb.im };
    const Cx2 O = Cx2{ Za.im + Zb.im, Zb.re - Za.re };
    const float2 Ex = make_float2(E.re.x, E.im.x), Ey = make_float2(E.re.y, E.im.y);
    const float2 Ox = make_float2(O.re.x, O.im.x), Oy = make_float2(O.re.y, O.im.y);
    const float2 W = cadd(cmulc(Ex, Ey), cmulc(Ox, Oy));
    const float2 U = cadd(cmulc(Ox, Ey), cmulc(cmulc(Ex, Oy), w2));
    Gk = make_float2(0.5f * (W.x - U.y), 0.5f * (W.y + U.x));
    Gm = make_float2(0.5f * (W.x + U.y), 0.5f * (U.x - W.y));
}

