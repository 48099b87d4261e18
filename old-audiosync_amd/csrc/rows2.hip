// csrc/rows2.hip -- k_rows2: the row kernel with TWO passes over LDS per transform instead of three.
//
// Same job as k_rows (xcorr_kernels.hip): for the spectrum rows k1 and m1 = M1 - k1 of one pair -- forward row
// transforms of both spectra, real-FFT untangling, X * conj(Y) (src/cross_correlation.c:232-233), inverse
// tangling, inverse row transforms, both four-step twiddles.  Replaces FFTW's r2c / c2r row work of
// src/cross_correlation.c:34-39,237-239 for row lengths M2 = RA * RB with RA, RB <= 40.
//
// k_rows is bound by its on-chip chain: six passes over LDS and eight barriers per block with
// two-member radix-10/12 butterflies.  Here a transform is ONE exchange through LDS:
//
//   forward   pass 1  thread (row, j), j < RB: its RA inputs z[j + RB*t] come straight from HBM (no fill
//                     phase), times the four-step twiddle, radix-RA butterfly, times w_M2^(j*u), to LDS [u][j]
//             pass 2  thread (row, u), u < RA: reads its RB contiguous values, radix-RB butterfly,
//                     bin k2 = u + RA*v back to LDS slot (u, v)
//   combine   slot pairs (s, M2-1-s) as in k_rows (digit reversal complements both digits), G rows in place
//   inverse   pass 2  thread (row, u): radix-RB inverse butterfly over v, in place
//             pass 1  thread (row, j): reads [u][j], times conj(w_M2^(j*u)), radix-RA inverse butterfly, times
//                     the conjugate four-step twiddle, and its RA outputs go straight to HBM (no store phase)
//
// 2 + 2 + 1 + 1 + 1 LDS write passes over a row instead of 4 + 4*3 + 2 + 2*3, 5 barriers instead of 8, one
// stage twiddle per point and transform instead of two.  A butterfly of radix 30 / 40 holds 60 / 80 data
// registers, so every thread transforms ONE sequence (Cx1), rows sit in LDS as float2 with a pitch of RB + 1
// (pass 2 strides 2*(RB+1) dwords from lane to lane: conflict-free for even RB).
#include "asx_internal.h"
#include "lds_fft.h"
#include "xcorr_dev.h"

#include <stdlib.h>

// (The ablations whose numbers DESIGN.md 5.0 quotes -- no loads, no butterflies, no twiddles, no combine, cache-resident
// rows -- are tools/experiments/kernel_switches.patch.)
#ifndef ASX_ROWS2_WAVES
#define ASX_ROWS2_WAVES 3 // launch bound: waves per SIMD = blocks per CU (4, by LDS) * NT / 256
#endif

// tw[u] = base * w^(j*u) for u < R, w = w_M2 (table `tw2`, natural order; j*u < M2 for j < M2/R).
// u = a + G*b: four table reads (w^j, w^2j, w^Gj, w^2Gj; the "seeds", fetched early), products of depth <= 3
// made where the twiddles are used (60-80 registers that must not be live across a butterfly).
template <int R> struct TwSplit {
    static constexpr int pick()
    {
        int g = R;
        for (int c = 1; c <= R; c++)
            if (R % c == 0 && c >= R / c) { g = c; break; }
        return g;
    }
    static constexpr int G = pick(), B = R / G;
};
struct Rows2Seeds {
    float2 w1, w2, v1, v2, base; // w^j, w^2j, w^Gj, w^2Gj and the four-step factor folded into every twiddle
};
template <int R> __device__ __forceinline__ Rows2Seeds rows2_seeds(const float2 *__restrict__ tw2, int j, float2 base)
{
    constexpr int G = TwSplit<R>::G;
    return Rows2Seeds{ tw2[j], tw2[2 * j], tw2[G * j], tw2[2 * G * j], base };
}
template <int R, bool NOBASE> __device__ __forceinline__ void rows2_twiddles(const Rows2Seeds &sd, float2 (&tw)[R])
{
    constexpr int G = TwSplit<R>::G, B = TwSplit<R>::B;
    static_assert(G >= 3 && B >= 3, "radix too small for the two-digit twiddle scheme");
    float2 wa[G], vb[B];
    wa[1] = sd.w1; wa[2] = sd.w2;
    vb[1] = sd.v1; vb[2] = sd.v2;
    static_for<3, G>([&](auto A) __attribute__((always_inline)) {
        constexpr int a = decltype(A)::value;
        wa[a] = (a % 2 == 0) ? cmul(wa[a / 2], wa[a / 2]) : cmul(wa[a - 1], wa[1]);
    });
    static_for<3, B>([&](auto BB) __attribute__((always_inline)) {
        constexpr int b = decltype(BB)::value;
        vb[b] = (b % 2 == 0) ? cmul(vb[b / 2], vb[b / 2]) : cmul(vb[b - 1], vb[1]);
    });
    static_for<0, B>([&](auto BB) __attribute__((always_inline)) {
        constexpr int b = decltype(BB)::value;
        const float2 bb = NOBASE ? (b == 0 ? make_float2(1.f, 0.f) : vb[b]) : (b == 0 ? sd.base : cmul(sd.base, vb[b]));
        tw[G * b] = bb;
        static_for<1, G>([&](auto A) __attribute__((always_inline)) {
            constexpr int a = decltype(A)::value;
            tw[a + G * b] = (NOBASE && b == 0) ? wa[a] : cmul(bb, wa[a]);
        });
    });
}

// FSX (diagnostic, wrong results): no four-step twiddles, as if the column kernels applied them.  Measured: k_rows2
// 1.17 -> 1.03 ms, but the same work costs k_fwd_cols +0.17 ms and k_inv_cols +0.07 ms (their VALU is not idle either):
// tools/experiments/fourstep_in_cols.patch.
template <int RA, int RB, int NT, bool FSX>
__global__ __launch_bounds__(NT, ASX_ROWS2_WAVES) void k_rows2(const AsxDev *__restrict__ Pp, const float2 *__restrict__ zxa,
                                                                const float2 *__restrict__ zya, float2 *__restrict__ ga,
                                                                const int4 *__restrict__ row_tasks, int M1, uint32_t M,
                                                                AsxPeakWs W)
{
    constexpr int M2 = RA * RB, PT = RB + 1, RS = RA * PT; // row pitch and row-region size in float2 slots
    // pass 2 strides 2*PT dwords from lane to lane: with an odd pitch the 32 lanes of a ds_read_b64 group fall on 32
    // different even banks
    static_assert(RB % 2 == 0, "the pitch RB + 1 must be odd");
    const AsxDev &PD = *Pp;
    const AsxKP P = asx_kp(PD);
    const int nrows = M1 / 2 + 1;
    __shared__ float2 buf[4 * RS];  // rows Xa, Ya, Xb, Yb; after the combine rows 0 and 2 hold G_k1, G_m1
    __shared__ float2 leg[2][RA];   // w_M^(row * RB * t): the four-step twiddle's step from leg to leg

    const int task = blockIdx.x, tid = threadIdx.x;
    const int pair = task / nrows;
    const int4 rt = row_tasks[task - pair * nrows];
    const int pa = rt.x, pb = rt.y;
    const int k1 = rt.z, m1 = rt.w;
    const bool self = (k1 == m1);
    if (k1 == 0 && tid < 64) {
        // Row 0 of a pair also prepares the pair's peak search (as k_rows does): the float32 error bound from
        // the norms k_fwd_cols left, the running maximum and the candidate count back to zero.
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
    const size_t wpair = (size_t)pair;
    const float2 *gx = zxa + wpair * M, *gy = zya + wpair * M;
    float2 *go = ga + wpair * M;
    const uint32_t krow[2] = { (uint32_t)k1, (uint32_t)m1 };
    const int prow[2] = { pa, pb };

    // ---- forward pass 1: fed from HBM ------------------------------------------------------------------
    // items (r, j): r = 0 Xa, 1 Ya, 2 Xb, 3 Yb.  All RA loads of an item are in flight before anything else.
    const int nfwd1 = (self ? 2 : 4) * RB;
    static_assert(4 * RB <= NT, "one forward pass-1 item per thread");
    float2 z[RA];
    const int r1 = tid / RB, j1 = tid - r1 * RB; // also the (g, j) of the inverse pass 1 (items 0..2*RB)
    const bool has1 = tid < nfwd1;
    if (has1) {
        const float2 *src = ((r1 & 1) ? gy : gx) + (size_t)prow[r1 >> 1] * M2 + j1;
        static_for<0, RA>([&](auto T) __attribute__((always_inline)) { z[T] = src[decltype(T)::value * RB]; });
    }
    if (!FSX && tid < 2 * RA) {
        const int which = tid >= RA, t = tid - which * RA;
        leg[which][t] = tw_F(P, 2u * krow[which] * (uint32_t)(RB * t));
    }
    Rows2Seeds sd{};
    if (has1) sd = rows2_seeds<RA>(P.tw2, j1, FSX ? make_float2(1.f, 0.f) : tw_F(P, 2u * krow[r1 >> 1] * (uint32_t)j1));
    if (!FSX) __syncthreads(); // leg[] visible
    if (has1) {
        Cx1 a[RA];
        const float2 *lg = leg[r1 >> 1];
        static_for<0, RA>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            a[t] = (t == 0 || FSX) ? Cx1{ z[t].x, z[t].y } : mulw(Cx1{ z[t].x, z[t].y }, lg[t]);
        });
        Bfly<RA, false>::run(a);
        float2 tw[RA];
        rows2_twiddles<RA, FSX>(sd, tw);
        float2 *o = buf + r1 * RS + j1;
        static_for<0, RA>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            const Cx1 y = mulw(a[u], tw[u]);
            o[u * PT] = make_float2(y.re, y.im);
        });
    }
    __syncthreads();

    // ---- forward pass 2: contiguous radix-RB butterflies, bins k2 = u + RA*v to slot (u, v) ---------------
    {
        const int n2 = (self ? 2 : 4) * RA;
        static_assert(4 * RA <= NT, "one forward pass-2 item per thread");
        if (tid < n2) {
            const int r = tid / RA, u = tid - r * RA;
            float2 *p = buf + r * RS + u * PT;
            Cx1 v[RB];
            static_for<0, RB>([&](auto X) __attribute__((always_inline)) { const float2 t = p[decltype(X)::value]; v[X] = Cx1{ t.x, t.y }; });
            Bfly<RB, false>::run(v);
            static_for<0, RB>([&](auto X) __attribute__((always_inline)) { p[decltype(X)::value] = make_float2(v[X].re, v[X].im); });
        }
    }
    __syncthreads();

    // ---- spectral combine -----------------------------------------------------------------------------
    const float2 wA = tw_F(P, 2u * (uint32_t)k1); // w_M^k1; w_M^(k1 + M1*k2) = wA * w_M2^k2
    float2 *Xa = buf, *Ya = buf + RS, *Xb = buf + 2 * RS, *Yb = buf + 3 * RS;
    if (!self) {
        // A thread walks slot pairs (s, s' = M2-1-s), s < M2/2: bin(s) of row k1 pairs with bin(s') of row m1 and
        // bin(s') of k1 with bin(s) of m1.  It reads eight values and rewrites four slots nobody else touches:
        // no barrier between the reads and the writes.  Slot s = u*RB + v sits at u*PT + v, its mirror at
        // RS - 2 - (u*PT + v).
        constexpr int NP = M2 / 2, STEPS = (NP + NT - 1) / NT;
        // steps whose 64 slots all lie past the end are skipped by the whole wave (the last step of most waves)
        const int wave0 = tid & ~63;
        auto step = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            int s = tid + i * NT;
            const bool mine = s < NP;
            s = mine ? s : 0; // clamped: the loads are unconditional
            const int a0 = s + s / RB, a1 = RS - 2 - a0; // u*PT + v and its mirror
            const float2 w2a = PD.tw2r[s], w2b = PD.tw2r[M2 - 1 - s];
            const float2 xa0 = Xa[a0], ya0 = Ya[a0], xb1 = Xb[a1], yb1 = Yb[a1];
            const float2 xa1 = Xa[a1], ya1 = Ya[a1], xb0 = Xb[a0], yb0 = Yb[a0];
            float2 gk0, gm0, gk1, gm1;
            combine_pair(Cx2{ v2f{ xa0.x, ya0.x }, v2f{ xa0.y, ya0.y } }, Cx2{ v2f{ xb1.x, yb1.x }, v2f{ xb1.y, yb1.y } },
                         cmul(wA, w2a), gk0, gm0);
            combine_pair(Cx2{ v2f{ xa1.x, ya1.x }, v2f{ xa1.y, ya1.y } }, Cx2{ v2f{ xb0.x, yb0.x }, v2f{ xb0.y, yb0.y } },
                         cmul(wA, w2b), gk1, gm1);
            if (mine) {
                Xa[a0] = gk0; Xb[a1] = gm0; // G_k1 at slot s, G_m1 at slot s'
                Xa[a1] = gk1; Xb[a0] = gm1;
            }
        };
        static_for<0, STEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr ((i + 1) * NT <= NP) step(I); // every wave has live slots
            else if (wave0 + i * NT < NP) step(I);     // wave-uniform
        });
    } else {
        // Self-paired rows (k1 = 0, M1/2; two blocks per pair): bins pair up inside the row.  Every thread
        // computes its G values into registers, then, after a barrier, writes them.  Bin k2 = u + RA*v sits at
        // slot address u*PT + v.
        constexpr int STEPS = (M2 / 2 + 1 + NT - 1) / NT;
        float2 gk[STEPS], gm[STEPS];
        int sa[STEPS], sb[STEPS];
        static_for<0, STEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int k2 = tid + i * NT;
            sa[i] = -1; sb[i] = -1;
            gk[i] = gm[i] = make_float2(0.f, 0.f);
            auto slot = [](int k) { return (k % RA) * PT + k / RA; };
            auto at = [&](int ad) { return Cx2{ v2f{ Xa[ad].x, Ya[ad].x }, v2f{ Xa[ad].y, Ya[ad].y } }; };
            if (k1 == 0) {
                if (k2 == 0) {
                    // DC and Nyquist bins are real: X[0] = Re Z0 + Im Z0, X[M] = Re Z0 - Im Z0
                    const Cx2 zz = at(0);
                    const float P0 = (zz.re.x + zz.im.x) * (zz.re.y + zz.im.y);
                    const float PM = (zz.re.x - zz.im.x) * (zz.re.y - zz.im.y);
                    sa[i] = 0;
                    gk[i] = make_float2(P0 + PM, P0 - PM);
                } else if (k2 <= M2 / 2) {
                    const int m2 = M2 - k2;
                    sa[i] = slot(k2);
                    const int s2 = slot(m2);
                    combine_pair(at(sa[i]), at(s2), P.tw2[k2], gk[i], gm[i]);
                    if (m2 != k2) sb[i] = s2;
                }
            } else { // k1 == M1/2, M1 even
                if (k2 < M2 / 2) {
                    const int m2 = M2 - 1 - k2;
                    sa[i] = slot(k2);
                    sb[i] = slot(m2);
                    combine_pair(at(sa[i]), at(sb[i]), cmul(wA, P.tw2[k2]), gk[i], gm[i]);
                }
            }
        });
        __syncthreads();
        static_for<0, STEPS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if (sa[i] >= 0) Xa[sa[i]] = gk[i];
            if (sb[i] >= 0) Xa[sb[i]] = gm[i];
        });
    }
    __syncthreads();

    // ---- inverse pass 2: radix-RB inverse butterflies over v, in place (rows 0 and 2) ---------------------
    {
        const int n2 = (self ? 1 : 2) * RA;
        if (tid < n2) {
            const int g = tid / RA, u = tid - g * RA;
            float2 *p = buf + 2 * g * RS + u * PT;
            Cx1 v[RB];
            static_for<0, RB>([&](auto X) __attribute__((always_inline)) { const float2 t = p[decltype(X)::value]; v[X] = Cx1{ t.x, t.y }; });
            Bfly<RB, true>::run(v);
            static_for<0, RB>([&](auto X) __attribute__((always_inline)) { p[decltype(X)::value] = make_float2(v[X].re, v[X].im); });
        }
    }
    // the twiddles of the inverse pass 1 (same as the forward ones of the same row and j) while the others transform
    const int ninv1 = (self ? 1 : 2) * RB;
    const int gi = tid / RB, ji = tid - gi * RB;
    const bool hasi = tid < ninv1;
    if (hasi) sd = rows2_seeds<RA>(P.tw2, ji, FSX ? make_float2(1.f, 0.f) : tw_F(P, 2u * krow[gi] * (uint32_t)ji));
    __syncthreads();

    // ---- inverse pass 1: reads [u][j], conj twiddles, radix-RA inverse butterfly, straight to HBM ----------
    if (hasi) {
        const float2 *p = buf + 2 * gi * RS + ji;
        float2 tw[RA];
        rows2_twiddles<RA, FSX>(sd, tw);
        Cx1 a[RA];
        static_for<0, RA>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            const float2 t = p[u * PT];
            a[u] = mulwc(Cx1{ t.x, t.y }, tw[u]);
        });
        Bfly<RA, true>::run(a);
        const float2 *lg = leg[gi];
        float2 *dst = go + (size_t)prow[gi] * M2 + ji;
        static_for<0, RA>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            const Cx1 y = (t == 0 || FSX) ? a[t] : mulwc(a[t], lg[t]);
            dst[t * RB] = make_float2(y.re, y.im);
        });
    }
}

// ---------------------------------------------------------------------------
// launcher: true when the plan's row length has a two-pass kernel compiled in
// ---------------------------------------------------------------------------
bool asx_launch_rows2(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga, const AsxPeakWs &W, int npairs,
                      hipStream_t s)
{
    // (plan creation uploads tw2r only when the environment asks for this kernel, ASX_ROWS2=1: opt-in, asx_api.hip)
    if (!P.tw2r) return false;
    const int ntasks = (P.M1 / 2 + 1) * npairs;
#define ASX_ROWS2_CASE(ra, rb, nt)                                                                                      \
    if (P.rows2_ra == (ra) && P.rows2_rb == (rb)) {                                                                     \
        hipLaunchKernelGGL((k_rows2<ra, rb, nt, ASX_ROWS2_FSX>), dim3(ntasks), dim3(nt), 0, s, P.self_dev, zxa, zya, ga,    \
                           P.row_tasks, P.M1, P.M, W);                                                                  \
        return true;                                                                                                    \
    }
#ifndef ASX_ROWS2_NT
#define ASX_ROWS2_NT 192
#endif
#ifndef ASX_ROWS2_FSX
#define ASX_ROWS2_FSX false
#endif
    ASX_ROWS2_CASE(30, 40, ASX_ROWS2_NT)
    ASX_ROWS2_CASE(40, 30, ASX_ROWS2_NT)
#undef ASX_ROWS2_CASE
    return false;
}
