// csrc/lds_fft.h — in-place mixed-radix transforms inside LDS for gfx950.
//
// Forward = decimation in frequency (Gentleman-Sande): natural order in,
// digit-reversed order out.  Inverse = decimation in time with the stages run
// backwards: digit-reversed in, natural order out.  Both are in place, so a
// tile needs ONE LDS copy (no Stockham ping-pong), each butterfly touches only
// its own R slots (no intra-stage hazard) and the permuted spectrum is never
// un-permuted: the spectral product only needs X and Y in the SAME order, and
// index tables (pos2_of_k2) locate the k <-> M-k partners.
//
// Radices 2,3,4,5 are hand-written; 6,8,9,10,12,15,16 are built from them at
// compile time (one Cooley-Tukey step in registers, constant inner twiddles),
// so a length of 800..2000 takes THREE passes over LDS instead of five or six
// -- LDS bandwidth, not HBM, is what the row kernel runs out of first.
// Stage twiddles w^(j*u) come from ONE (or two) table reads per butterfly and
// a short product tree (depth <= 3), not R-1 gathered reads.
// Index algebra is modelled and tested in tests/model_fourstep.py.
//
// Replaces FFTW's r2c/c2r kernels as used at src/cross_correlation.c:34-39,237-239.
#pragma once

#include "asx_internal.h"

#include <type_traits>

#ifndef ASX_WIDE_MAXR
#define ASX_WIDE_MAXR 10   // largest radix whose two members are held in registers together
#endif

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// x / d and x % d for 0 <= x < 2^23 using a precomputed float reciprocal; exact
// (the float quotient is off by at most one, fixed up with the remainder).
__device__ __forceinline__ int div_exact(int x, int d, float inv_d, int &rem)
{
    int qq = (int)((float)x * inv_d);
    int r = x - qq * d;
    if (r < 0) { qq--; r += d; }
    else if (r >= d) { qq++; r -= d; }
    rem = r;
    return qq;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- compile-time roots of unity -------------------------------------------------
constexpr double asx_ct_pi = 3.14159265358979323846264338327950288;
constexpr double asx_ct_sin(double x) // |x| <= pi
{
    double term = x, sum = x;
    for (int k = 1; k < 16; k++) {
        term *= -x * x / (double)((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double asx_ct_cos(double x)
{
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 16; k++) {
        term *= -x * x / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
// w_R^m = exp(-2 pi i m / R)
template <int R, int M> struct Root {
    static constexpr int m = ((M % R) + R) % R;
    static constexpr int ms = m > R / 2 ? m - R : m; // angle in (-pi, pi]
    static constexpr float re = (float)asx_ct_cos(2.0 * asx_ct_pi * (double)ms / (double)R);
    static constexpr float im = (float)(-asx_ct_sin(2.0 * asx_ct_pi * (double)ms / (double)R));
};

// a * w_R^M (forward) or a * conj(w_R^M) (INV)
template <int R, int M, bool INV> __device__ __forceinline__ float2 mul_root(float2 a)
{
    constexpr int m = ((M % R) + R) % R;
    if constexpr (m == 0) {
        return a;
    } else if constexpr (2 * m == R) {
        return make_float2(-a.x, -a.y);
    } else if constexpr (4 * m == R) { // w = -i ; conj = +i
        return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    } else if constexpr (4 * m == 3 * R) { // w = +i
        return INV ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
    } else {
        constexpr float c = Root<R, m>::re;
        constexpr float s = INV ? -Root<R, m>::im : Root<R, m>::im;
        return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

// ---- radix butterflies: v <- DFT_R(v) in natural order (INV: conjugate kernel, unnormalised) ----
template <int R, bool INV> struct Bfly;

template <bool INV> struct Bfly<2, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[2])
    {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV> struct Bfly<4, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[4])
    {
        float2 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
        float2 s1 = cadd(v[1], v[3]), d1 = csub(v[1], v[3]);
        // forward: -i*d1 ; inverse: +i*d1
        float2 r = INV ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
        v[0] = cadd(s0, s1);
        v[2] = csub(s0, s1);
        v[1] = cadd(d0, r);
        v[3] = csub(d0, r);
    }
};

template <bool INV> struct Bfly<3, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[3])
    {
        const float sn = 0.86602540378443864676f; // sin(2*pi/3)
        float2 t1 = cadd(v[1], v[2]), t2 = csub(v[1], v[2]);
        float2 m = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        float2 r = INV ? make_float2(-sn * t2.y, sn * t2.x) : make_float2(sn * t2.y, -sn * t2.x);
        v[0] = cadd(v[0], t1);
        v[1] = cadd(m, r);
        v[2] = csub(m, r);
    }
};

template <bool INV> struct Bfly<5, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[5])
    {
        const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
        const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
        float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
        float2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
        float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        float2 u1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
        float2 u2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
        // forward: -i*u ; inverse: +i*u
        float2 r1 = INV ? make_float2(-u1.y, u1.x) : make_float2(u1.y, -u1.x);
        float2 r2 = INV ? make_float2(-u2.y, u2.x) : make_float2(u2.y, -u2.x);
        v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        v[1] = cadd(m1, r1);
        v[4] = csub(m1, r1);
        v[2] = cadd(m2, r2);
        v[3] = csub(m2, r2);
    }
};

// Composite radix R = R1*R2 in registers:
//   t = R2*t1 + t2,  u = u1 + R1*u2
//   w_R^(t*u) = w_R1^(t1*u1) * w_R^(t2*u1) * w_R2^(t2*u2)
template <int R1, int R2, bool INV> struct BflyC {
    static constexpr int R = R1 * R2;
    static __device__ __forceinline__ void run(float2 (&v)[R])
    {
        // DFT_R1 over t1 for every t2, then the inner twiddle w_R^(t2*u1)
        static_for<0, R2>([&](auto T2) __attribute__((always_inline)) {
            constexpr int t2 = decltype(T2)::value;
            float2 x[R1];
            static_for<0, R1>([&](auto T1) __attribute__((always_inline)) { x[T1] = v[R2 * T1 + t2]; });
            Bfly<R1, INV>::run(x);
            static_for<0, R1>([&](auto U1) __attribute__((always_inline)) { v[R2 * U1 + t2] = mul_root<R, t2 * decltype(U1)::value, INV>(x[U1]); });
        });
        // DFT_R2 over t2 for every u1; outputs land in natural order u = u1 + R1*u2
        float2 y[R];
        static_for<0, R1>([&](auto U1) __attribute__((always_inline)) {
            constexpr int u1 = decltype(U1)::value;
            float2 x[R2];
            static_for<0, R2>([&](auto T2) __attribute__((always_inline)) { x[T2] = v[R2 * u1 + T2]; });
            Bfly<R2, INV>::run(x);
            static_for<0, R2>([&](auto U2) __attribute__((always_inline)) { y[u1 + R1 * U2] = x[U2]; });
        });
        static_for<0, R>([&](auto I) __attribute__((always_inline)) { v[I] = y[I]; });
    }
};

template <bool INV> struct Bfly<6, INV> : BflyC<2, 3, INV> {};
template <bool INV> struct Bfly<8, INV> : BflyC<2, 4, INV> {};
template <bool INV> struct Bfly<9, INV> : BflyC<3, 3, INV> {};
template <bool INV> struct Bfly<10, INV> : BflyC<2, 5, INV> {};
template <bool INV> struct Bfly<12, INV> : BflyC<3, 4, INV> {};
template <bool INV> struct Bfly<15, INV> : BflyC<3, 5, INV> {};
template <bool INV> struct Bfly<16, INV> : BflyC<4, 4, INV> {};

// Stage twiddles w[u] = W^u, u = 1..R-1, W = tw[tj]: table reads at u = 1 and 4
// (u*tj < n always holds), the rest by products of depth <= 3.
template <int R> __device__ __forceinline__ void stage_twiddles(const float2 *__restrict__ tw, int tj, float2 (&w)[R])
{
    w[0] = make_float2(1.f, 0.f);
    if constexpr (R >= 2) w[1] = tw[tj];
    static_for<2, R>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        if constexpr (u == 4) {
            w[u] = tw[4 * tj];
        } else if constexpr (u < 4) {
            w[u] = cmul(w[u - 1], w[1]);
        } else {
            constexpr int lo = u % 4, hi = u - lo;
            if constexpr (lo == 0) w[u] = cmul(w[u - 4], w[4]);
            else w[u] = cmul(w[hi], w[lo]);
        }
    });
}

// ---------------------------------------------------------------------------
// One stage.  A work item is one butterfly position (sub-block b, offset j) of one
// GROUP of CPT transforms that share the stage twiddles (generated once per work item):
//   element e of member m of group g  ->  lds[g*group_stride + m*member_stride + e*elem_stride]
//   ADJ  : CPT == 2 and member_stride == 1 with 16-byte aligned pairs: both members move
//          with ONE ds_read_b128 / ds_write_b128 (column tiles: two adjacent columns;
//          row tiles: the X and Y spectra interleaved)
//   GFAST: consecutive lanes walk the group index first (ngroups is a power of two) so a
//          wave touches contiguous LDS in column tiles; otherwise lanes walk butterflies.
//   UNIT_TW: the stage's twiddles are all 1 (q == 1, the innermost stage).
// ---------------------------------------------------------------------------
struct LdsLayout {
    int ngroups, log_ngroups;  // groups of CPT transforms
    int elem_stride;           // in float2 units
    int group_stride;
    int member_stride;
};

template <int R, bool INV, int CPT, bool ADJ, bool GFAST, bool UNIT_TW>
__device__ __forceinline__ void lds_stage(float2 *lds, const AsxStages &st, int i, const LdsLayout &L,
                                          const float2 *__restrict__ tw)
{
    static_assert(!ADJ || CPT == 2, "ADJ means two adjacent members");
    // both members in registers at once only while that fits 128 VGPRs; larger radices
    // run the members one after the other (twiddles stay in registers either way)
    constexpr bool WIDE = ADJ && (R <= ASX_WIDE_MAXR);
    const int ns = st.ns[i], q = st.q[i], nbf = st.nbf[i], twmul = st.twmul[i];
    const float inv_q = st.inv_q[i], inv_nbf = st.inv_nbf[i];
    const int total = L.ngroups * nbf;
    const int step = q * L.elem_stride;
    const int mstride = ADJ ? 1 : L.member_stride;
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        int g, bf;
        if (GFAST) {
            g = w & (L.ngroups - 1);
            bf = w >> L.log_ngroups;
        } else {
            g = div_exact(w, nbf, inv_nbf, bf);
        }
        int j;
        const int b = div_exact(bf, q, inv_q, j);
        float2 *p = lds + g * L.group_stride + (b * ns + j) * L.elem_stride;
        float2 tww[R];
        if constexpr (!UNIT_TW) stage_twiddles<R>(tw, j * twmul, tww);

        auto compute = [&](float2(&v)[R]) __attribute__((always_inline)) {
            if constexpr (UNIT_TW) {
                Bfly<R, INV>::run(v);
            } else if constexpr (!INV) {
                Bfly<R, false>::run(v);
                static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = cmul(v[U], tww[U]); });
            } else {
                static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = cmulc(v[U], tww[U]); });
                Bfly<R, true>::run(v);
            }
        };

        if constexpr (WIDE) {
            float2 v0[R], v1[R];
            static_for<0, R>([&](auto T) __attribute__((always_inline)) {
                const float4 x = *reinterpret_cast<const float4 *>(p + T * step);
                v0[T] = make_float2(x.x, x.y);
                v1[T] = make_float2(x.z, x.w);
            });
            compute(v0);
            compute(v1);
            static_for<0, R>([&](auto T) __attribute__((always_inline)) {
                *reinterpret_cast<float4 *>(p + T * step) = make_float4(v0[T].x, v0[T].y, v1[T].x, v1[T].y);
            });
        } else {
            static_for<0, CPT>([&](auto M) __attribute__((always_inline)) {
                float2 v[R];
                float2 *pm = p + M * mstride;
                static_for<0, R>([&](auto T) __attribute__((always_inline)) { v[T] = pm[T * step]; });
                compute(v);
                static_for<0, R>([&](auto T) __attribute__((always_inline)) { pm[T * step] = v[T]; });
            });
        }
    }
}

template <int R, bool INV, int CPT, bool ADJ, bool GFAST>
__device__ __forceinline__ void lds_stage_r(float2 *lds, const AsxStages &st, int i, const LdsLayout &L,
                                            const float2 *__restrict__ tw)
{
    if (st.q[i] == 1) // wave-uniform
        lds_stage<R, INV, CPT, ADJ, GFAST, true>(lds, st, i, L, tw);
    else
        lds_stage<R, INV, CPT, ADJ, GFAST, false>(lds, st, i, L, tw);
}

// MAXR: largest radix this kernel variant carries code for.  Register allocation is per
// kernel, so a variant without the radix-15/16 bodies keeps the occupancy of the small ones.
template <int MAXR, bool INV, int CPT, bool ADJ, bool GFAST>
__device__ __forceinline__ void lds_stage_any(float2 *lds, const AsxStages &st, int i, const LdsLayout &L,
                                              const float2 *__restrict__ tw)
{
#define ASX_STAGE_CASE(R) \
    case R: if constexpr (R <= MAXR) lds_stage_r<R, INV, CPT, ADJ, GFAST>(lds, st, i, L, tw); break;
    switch (st.radix[i]) { // wave-uniform
        ASX_STAGE_CASE(16)
        ASX_STAGE_CASE(15)
        ASX_STAGE_CASE(12)
        ASX_STAGE_CASE(10)
        ASX_STAGE_CASE(9)
        ASX_STAGE_CASE(8)
        ASX_STAGE_CASE(6)
        ASX_STAGE_CASE(5)
        ASX_STAGE_CASE(4)
        ASX_STAGE_CASE(3)
    default: lds_stage_r<2, INV, CPT, ADJ, GFAST>(lds, st, i, L, tw); break;
    }
#undef ASX_STAGE_CASE
}

// Whole transform.  Caller has filled LDS and called __syncthreads(); on return
// all lanes see the result (the routine ends with a barrier).
template <int MAXR, bool INV, int CPT, bool ADJ, bool GFAST>
__device__ __forceinline__ void lds_fft(float2 *lds, const AsxStages &st, const LdsLayout &L,
                                        const float2 *__restrict__ tw)
{
    if (!INV) {
        for (int i = 0; i < st.nstages; i++) {
            lds_stage_any<MAXR, false, CPT, ADJ, GFAST>(lds, st, i, L, tw);
            __syncthreads();
        }
    } else {
        for (int i = st.nstages - 1; i >= 0; i--) {
            lds_stage_any<MAXR, true, CPT, ADJ, GFAST>(lds, st, i, L, tw);
            __syncthreads();
        }
    }
}
