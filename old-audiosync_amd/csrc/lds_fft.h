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
// Radices 2,3,4,5 are hand-written; 6,10,12,15 are built from them at compile time
// by the prime-factor maps (no inner twiddles), 8,9,16 by one Cooley-Tukey step in
// registers with constant inner twiddles,
// so a length of 800..2000 takes THREE passes over LDS (three barriers) instead
// of five or six.  Two transforms per thread share twiddles and index arithmetic; the
// floating-point work itself is plain scalar VALU (see v2f).
// Stage twiddles w^(j*u) come from ONE (or two) table reads per butterfly and
// a short product tree (depth <= 3), not R-1 gathered reads.
// Index algebra is modelled and tested in tests/model_fourstep.py.
//
// Replaces FFTW's r2c/c2r kernels as used at src/cross_correlation.c:34-39,237-239.
#pragma once

#include "asx_internal.h"

#include <type_traits>


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

// ---------------------------------------------------------------------------
// Pair-planar complex arithmetic.  Every thread transforms TWO sequences at once (two
// adjacent tile columns; or the X and Y spectra of one row; or the two rows of G) with the
// same twiddles:
//     Cx2.re = (re of member 0, re of member 1),  Cx2.im = (im of member 0, im of member 1)
// Stage twiddles are generated once per pair, multiplications by +-i or conjugations are
// register renames with sign modifiers, and in LDS an element pair is one float4
// {re0, im0, re1, im1}: one ds_read_b128 fills a Cx2.
// ---------------------------------------------------------------------------
// Two floats moved together, computed separately.  An ext_vector_type(2) here makes hipcc emit
// v_pk_add/mul/fma_f32.  On gfx950 a packed fp32 instruction costs 1.8-2x a scalar one
// (tools/micro/pk_forms.hip), so packing buys nothing by itself, and packing the two MEMBERS
// costs shuffles at every LDS access (k_rows 2.29 ms packed vs 1.86 ms scalar for the same
// algebra); packing re/im instead needs no shuffles and came out even
// (tools/experiments/packed_aos.patch).  The arithmetic is scalar on purpose and the kernels
// are built with -fno-slp-vectorize so that LLVM does not re-pack it.
struct v2f {
    float x, y;
};
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ v2f operator-(v2f a) { return v2f{ -a.x, -a.y }; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{ a.x * b.x, a.y * b.y }; }
__device__ __forceinline__ v2f operator*(v2f a, float b) { return v2f{ a.x * b, a.y * b }; }
__device__ __forceinline__ v2f operator*(float b, v2f a) { return v2f{ a.x * b, a.y * b }; }

struct Cx2 {
    v2f re, im;
};
// One complex value per thread ("single member"): the inverse row transforms of the real-column kernels (rlayout.hip)
// run on the single product row X conj(Y).
struct Cx1 {
    float re, im;
};

__device__ __forceinline__ Cx2 operator+(Cx2 a, Cx2 b) { return Cx2{ a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ Cx2 operator-(Cx2 a, Cx2 b) { return Cx2{ a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ Cx1 operator+(Cx1 a, Cx1 b) { return Cx1{ a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ Cx1 operator-(Cx1 a, Cx1 b) { return Cx1{ a.re - b.re, a.im - b.im }; }
// multiply by -i: (re, im) -> (im, -re);  by +i: (-im, re)
template <class C> __device__ __forceinline__ C mul_neg_i(C a) { return C{ a.im, -a.re }; }
template <class C> __device__ __forceinline__ C mul_pos_i(C a) { return C{ -a.im, a.re }; }
template <bool INV, class C> __device__ __forceinline__ C rot(C a) { return INV ? mul_pos_i(a) : mul_neg_i(a); }
// by a scalar twiddle shared by both members: a * w, a * conj(w)
template <class C> __device__ __forceinline__ C mulw(C a, float2 w)
{
    return C{ a.re * w.x - a.im * w.y, a.re * w.y + a.im * w.x };
}
template <class C> __device__ __forceinline__ C mulwc(C a, float2 w)
{
    return C{ a.re * w.x + a.im * w.y, a.im * w.x - a.re * w.y };
}
// member-wise complex products (different twiddle per member)
__device__ __forceinline__ Cx2 mul2(Cx2 a, Cx2 w) { return Cx2{ a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re }; }
__device__ __forceinline__ Cx2 mul2c(Cx2 a, Cx2 w) { return Cx2{ a.re * w.re + a.im * w.im, a.im * w.re - a.re * w.im }; }

// LDS slot = the two members' complex values side by side, {re0, im0, re1, im1}: for two adjacent
// tile columns that is exactly how they lie in HBM, so tiles move in and out without a shuffle.
__device__ __forceinline__ Cx2 lds_get(const float4 *p)
{
    const float4 x = *p;
    return Cx2{ v2f{ x.x, x.z }, v2f{ x.y, x.w } };
}
__device__ __forceinline__ void lds_put(float4 *p, Cx2 v) { *p = make_float4(v.re.x, v.im.x, v.re.y, v.im.y); }

// a * w_R^M (forward) or a * conj(w_R^M) (INV), compile-time root
template <int R, int M, bool INV, class C> __device__ __forceinline__ C mul_root(C a)
{
    constexpr int m = ((M % R) + R) % R;
    if constexpr (m == 0) {
        return a;
    } else if constexpr (2 * m == R) {
        return C{ -a.re, -a.im };
    } else if constexpr (4 * m == R) { // w = -i ; conj = +i
        return INV ? mul_pos_i(a) : mul_neg_i(a);
    } else if constexpr (4 * m == 3 * R) { // w = +i
        return INV ? mul_neg_i(a) : mul_pos_i(a);
    } else {
        constexpr float c = Root<R, m>::re;
        constexpr float s = INV ? -Root<R, m>::im : Root<R, m>::im;
        return C{ a.re * c - a.im * s, a.re * s + a.im * c };
    }
}

// ---- radix butterflies: v <- DFT_R(v) in natural order (INV: conjugate kernel, unnormalised) ----
// run() is generic in the element type: Cx2 (two transforms per thread) or Cx1 (one).
template <int R, bool INV> struct Bfly;

template <bool INV> struct Bfly<2, INV> {
    template <class C> static __device__ __forceinline__ void run(C (&v)[2])
    {
        const C a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <bool INV> struct Bfly<4, INV> {
    template <class C> static __device__ __forceinline__ void run(C (&v)[4])
    {
        const C s0 = v[0] + v[2], d0 = v[0] - v[2];
        const C s1 = v[1] + v[3], d1 = v[1] - v[3];
        const C r = rot<INV>(d1); // forward: -i*d1 ; inverse: +i*d1
        v[0] = s0 + s1;
        v[2] = s0 - s1;
        v[1] = d0 + r;
        v[3] = d0 - r;
    }
};

template <bool INV> struct Bfly<3, INV> {
    template <class C> static __device__ __forceinline__ void run(C (&v)[3])
    {
        constexpr float sn = 0.86602540378443864676f; // sin(2*pi/3)
        const C t1 = v[1] + v[2], t2 = v[1] - v[2];
        const C m = C{ v[0].re - 0.5f * t1.re, v[0].im - 0.5f * t1.im };
        const C st = C{ sn * t2.re, sn * t2.im };
        const C r = rot<INV>(st);
        v[0] = v[0] + t1;
        v[1] = m + r;
        v[2] = m - r;
    }
};

template <bool INV> struct Bfly<5, INV> {
    template <class C> static __device__ __forceinline__ void run(C (&v)[5])
    {
        constexpr float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
        constexpr float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
        const C a1 = v[1] + v[4], b1 = v[1] - v[4];
        const C a2 = v[2] + v[3], b2 = v[2] - v[3];
        const C m1 = C{ v[0].re + c1 * a1.re + c2 * a2.re, v[0].im + c1 * a1.im + c2 * a2.im };
        const C m2 = C{ v[0].re + c2 * a1.re + c1 * a2.re, v[0].im + c2 * a1.im + c1 * a2.im };
        const C u1 = C{ s1 * b1.re + s2 * b2.re, s1 * b1.im + s2 * b2.im };
        const C u2 = C{ s2 * b1.re - s1 * b2.re, s2 * b1.im - s1 * b2.im };
        const C r1 = rot<INV>(u1), r2 = rot<INV>(u2); // forward: -i*u ; inverse: +i*u
        v[0] = C{ v[0].re + a1.re + a2.re, v[0].im + a1.im + a2.im };
        v[1] = m1 + r1;
        v[4] = m1 - r1;
        v[2] = m2 + r2;
        v[3] = m2 - r2;
    }
};

// Composite radix R = R1*R2 in registers:
//   t = R2*t1 + t2,  u = u1 + R1*u2
//   w_R^(t*u) = w_R1^(t1*u1) * w_R^(t2*u1) * w_R2^(t2*u2)
template <int R1, int R2, bool INV> struct BflyC {
    static constexpr int R = R1 * R2;
    template <class C> static __device__ __forceinline__ void run(C (&v)[R])
    {
        // DFT_R1 over t1 for every t2, then the inner twiddle w_R^(t2*u1)
        static_for<0, R2>([&](auto T2) __attribute__((always_inline)) {
            constexpr int t2 = decltype(T2)::value;
            C x[R1];
            static_for<0, R1>([&](auto T1) __attribute__((always_inline)) { x[T1] = v[R2 * T1 + t2]; });
            Bfly<R1, INV>::run(x);
            static_for<0, R1>([&](auto U1) __attribute__((always_inline)) {
                v[R2 * U1 + t2] = mul_root<R, t2 * decltype(U1)::value, INV>(x[U1]);
            });
        });
        // DFT_R2 over t2 for every u1; outputs land in natural order u = u1 + R1*u2
        C y[R];
        static_for<0, R1>([&](auto U1) __attribute__((always_inline)) {
            constexpr int u1 = decltype(U1)::value;
            C x[R2];
            static_for<0, R2>([&](auto T2) __attribute__((always_inline)) { x[T2] = v[R2 * u1 + T2]; });
            Bfly<R2, INV>::run(x);
            static_for<0, R2>([&](auto U2) __attribute__((always_inline)) { y[u1 + R1 * U2] = x[U2]; });
        });
        static_for<0, R>([&](auto I) __attribute__((always_inline)) { v[I] = y[I]; });
    }
};

// Coprime composite radix R = R1*R2 in registers by the prime-factor (Good-Thomas) maps:
//   input  t = (R2*t1 + R1*t2) mod R,   output u with u = u1 (mod R1), u = u2 (mod R2)
//   DFT_R[u] = sum_{t1,t2} x[t] w_R1^(t1*u1) w_R2^(t2*u2)
// -- no inner twiddles at all, and both maps are compile-time register renames.  Radix 10 costs
// 92 real operations per butterfly instead of 108, radix 12 104 instead of 120.
template <int R1, int R2, bool INV> struct BflyP {
    static constexpr int R = R1 * R2;
    static constexpr int crt(int u1, int u2)
    {
        for (int u = 0; u < R; u++)
            if (u % R1 == u1 && u % R2 == u2) return u;
        return 0;
    }
    template <class C> static __device__ __forceinline__ void run(C (&v)[R])
    {
        C a[R]; // a[R2*u1 + t2]
        static_for<0, R2>([&](auto T2) __attribute__((always_inline)) {
            constexpr int t2 = decltype(T2)::value;
            C x[R1];
            static_for<0, R1>([&](auto T1) __attribute__((always_inline)) {
                x[T1] = v[(R2 * decltype(T1)::value + R1 * t2) % R];
            });
            Bfly<R1, INV>::run(x);
            static_for<0, R1>([&](auto U1) __attribute__((always_inline)) { a[R2 * decltype(U1)::value + t2] = x[U1]; });
        });
        static_for<0, R1>([&](auto U1) __attribute__((always_inline)) {
            constexpr int u1 = decltype(U1)::value;
            C x[R2];
            static_for<0, R2>([&](auto T2) __attribute__((always_inline)) { x[T2] = a[R2 * u1 + decltype(T2)::value]; });
            Bfly<R2, INV>::run(x);
            static_for<0, R2>([&](auto U2) __attribute__((always_inline)) {
                constexpr int u = crt(u1, decltype(U2)::value); // forced constant: a run-time crt() of radix 30/40 would index registers dynamically
                v[u] = x[U2];
            });
        });
    }
};

template <bool INV> struct Bfly<6, INV> : BflyP<2, 3, INV> {};
template <bool INV> struct Bfly<8, INV> : BflyC<2, 4, INV> {};
template <bool INV> struct Bfly<9, INV> : BflyC<3, 3, INV> {};
template <bool INV> struct Bfly<10, INV> : BflyP<2, 5, INV> {};
template <bool INV> struct Bfly<12, INV> : BflyP<3, 4, INV> {};
template <bool INV> struct Bfly<15, INV> : BflyP<3, 5, INV> {};
template <bool INV> struct Bfly<16, INV> : BflyC<4, 4, INV> {};

// ---------------------------------------------------------------------------
// One stage.  The LDS image is an array of float4 slots; slot (g, e) = element e of the
// transform PAIR g sits at lds4[g*group_stride + e*elem_stride].  A work item is one
// butterfly position (sub-block b, offset j) of one pair.
//   GFAST: consecutive lanes walk the pair index first (ngroups is a power of two) so a
//          wave touches contiguous LDS in column tiles; otherwise lanes walk butterflies.
//   UNIT_TW: the stage's twiddles are all 1 (q == 1, the innermost stage).
// ---------------------------------------------------------------------------
struct LdsLayout {
    int ngroups, log_ngroups;  // transform pairs in the tile
    int elem_stride;           // in float4 slots
    int group_stride;
    int nthreads;              // block size (a literal in the kernels with a compile-time schedule)
    int tid;                   // the thread's index in the block (threadIdx.x; a persistent kernel passes an opaque
                               // copy per task so that per-thread table reads are not hoisted out of its task loop)
};

// Twiddle prefetch: the two table reads (W^1, W^4) a thread needs for its FIRST work item of a
// stage depend only on (stage, threadIdx), not on data, so they are issued before the barrier
// that ends the previous stage (or before the tile load) and ride in registers across it;
// the L2 latency of the table read then overlaps the previous stage instead of stalling this one.
struct TwPre {
    float2 w1, w4;
};

// Geometry of one DIF stage, passed BY VALUE.  From a runtime schedule it is seven loads from the
// plan; from a compile-time schedule (Sched below) it is seven literals, and since every stage
// function is force-inlined the compiler folds them: no scalar loads, no radix switch, divisions
// by constants, and the R slot offsets of a butterfly become immediate offsets of the LDS
// instructions.  The production lengths run on such specialised instantiations (xcorr_kernels.hip).
struct StageK {
    int R, ns, q, nbf, twmul;
    float inv_q, inv_nbf;
};
__device__ __forceinline__ StageK stage_k(const AsxStages &st, int i)
{
    return StageK{ st.radix[i], st.ns[i], st.q[i], st.nbf[i], st.twmul[i], st.inv_q[i], st.inv_nbf[i] };
}
// compile-time schedule of an n-point transform: radices of the DIF stages in order
template <int N, int... Rs> struct Sched {
    static constexpr int n = N;
    static constexpr int nstages = (int)sizeof...(Rs);
    static constexpr int radix(int i)
    {
        constexpr int r[] = { Rs... };
        return r[i];
    }
    static constexpr int max_radix()
    {
        int m = 2;
        for (int i = 0; i < nstages; i++) m = radix(i) > m ? radix(i) : m;
        return m;
    }
    static constexpr StageK stage(int i)
    {
        int ns = N;
        for (int j = 0; j < i; j++) ns /= radix(j);
        const int R = radix(i), q = ns / R, nbf = N / R;
        return StageK{ R, ns, q, nbf, N / ns, 1.0f / (float)q, 1.0f / (float)nbf };
    }
};

template <bool GFAST>
__device__ __forceinline__ TwPre tw_prefetch_k(const StageK K, const LdsLayout &L, const float2 *__restrict__ tw)
{
    TwPre pre;
    pre.w1 = make_float2(1.f, 0.f);
    pre.w4 = make_float2(1.f, 0.f);
    const int w = L.tid;
    if (K.q == 1 || w >= L.ngroups * K.nbf) return pre;
    int bf;
    if (GFAST) bf = w >> L.log_ngroups;
    else (void)div_exact(w, K.nbf, K.inv_nbf, bf);
    int j;
    (void)div_exact(bf, K.q, K.inv_q, j);
    const int tj = j * K.twmul;
    pre.w1 = tw[tj];
    if (K.R > 4) pre.w4 = tw[4 * tj];
    return pre;
}
template <bool GFAST>
__device__ __forceinline__ TwPre tw_prefetch(const AsxStages &st, int i, const LdsLayout &L,
                                             const float2 *__restrict__ tw)
{
    if (i < 0 || i >= st.nstages) return TwPre{ make_float2(1.f, 0.f), make_float2(1.f, 0.f) };
    return tw_prefetch_k<GFAST>(stage_k(st, i), L, tw);
}

// Stage twiddles from (W^1, W^4): products of depth <= 3
template <int R> __device__ __forceinline__ void stage_twiddles_from(float2 w1, float2 w4, float2 (&w)[R])
{
    w[0] = make_float2(1.f, 0.f);
    if constexpr (R >= 2) w[1] = w1;
    static_for<2, R>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        if constexpr (u == 4) {
            w[u] = w4;
        } else if constexpr (u < 4) {
            w[u] = cmul(w[u - 1], w[1]);
        } else {
            constexpr int lo = u % 4, hi = u - lo;
            if constexpr (lo == 0) w[u] = cmul(w[u - 4], w[4]);
            else w[u] = cmul(w[hi], w[lo]);
        }
    });
}

// Sink: what happens to a butterfly's R outputs.  NoSink = written back to their slots (in place).
// Any other type is called as sink(std::integral_constant<int, R>{}, v, g, pos0, q): v[T] is element
// pos0 + T*q of transform pair g -- the outputs stay in registers and LDS keeps the stage's INPUT (the
// peak search of k_inv_cols consumes the last inverse stage this way: r never reaches LDS either).
struct NoSink {};
// Source: where a butterfly's R inputs come from.  NoSource = read from their slots.  Any other type is
// called as src(std::integral_constant<int, R>{}, v, g, pos0, q) and fills v[T] with element pos0 + T*q of
// transform pair g (k_inv_cols feeds its first stage straight from HBM this way: the tile never makes
// the LDS write + read + barrier of a separate fill phase).
struct NoSource {};

template <int R, bool INV, bool GFAST, bool UNIT_TW, class Sink = NoSink, class Source = NoSource>
__device__ __forceinline__ void lds_stage(float4 *lds, const StageK K, const LdsLayout &L,
                                          const float2 *__restrict__ tw, TwPre pre, Sink &&sink = Sink{},
                                          Source &&source = Source{})
{
    const int ns = K.ns, q = K.q, nbf = K.nbf, twmul = K.twmul;
    const float inv_q = K.inv_q, inv_nbf = K.inv_nbf;
    const int total = L.ngroups * nbf;
    const int step = q * L.elem_stride;
    for (int w = L.tid; w < total; w += L.nthreads) {
        int g, bf;
        if (GFAST) {
            g = w & (L.ngroups - 1);
            bf = w >> L.log_ngroups;
        } else {
            g = div_exact(w, nbf, inv_nbf, bf);
        }
        int j, b;
        if (UNIT_TW) { b = bf; j = 0; }                 // q == 1
        else if (nbf == q) { b = 0; j = bf; }           // first stage: one sub-block (wave-uniform test)
        else b = div_exact(bf, q, inv_q, j);
        float4 *p = lds + g * L.group_stride + (b * ns + j) * L.elem_stride;
        Cx2 v[R];
        if constexpr (std::is_same<typename std::decay<Source>::type, NoSource>::value)
            static_for<0, R>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + T * step); });
        else
            source(std::integral_constant<int, R>{}, v, g, b * ns + j, q);
        if constexpr (UNIT_TW) {
            Bfly<R, INV>::run(v);
        } else {
            float2 w1 = pre.w1, w4 = pre.w4;
            if (w != L.tid) { // later work items of this thread: read the table now
                const int tj = j * twmul;
                w1 = tw[tj];
                if constexpr (R > 4) w4 = tw[4 * tj];
            }
            float2 tww[R];
            stage_twiddles_from<R>(w1, w4, tww);
            if constexpr (!INV) {
                Bfly<R, false>::run(v);
                static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tww[U]); });
            } else {
                static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tww[U]); });
                Bfly<R, true>::run(v);
            }
        }
        if constexpr (std::is_same<typename std::decay<Sink>::type, NoSink>::value)
            static_for<0, R>([&](auto T) __attribute__((always_inline)) { lds_put(p + T * step, v[T]); });
        else
            sink(std::integral_constant<int, R>{}, v, g, b * ns + j, q);
    }
}

template <int R, bool INV, bool GFAST, class Sink = NoSink>
__device__ __forceinline__ void lds_stage_r(float4 *lds, const StageK K, const LdsLayout &L,
                                            const float2 *__restrict__ tw, TwPre pre, Sink &&sink = Sink{})
{
    if (K.q == 1) // wave-uniform
        lds_stage<R, INV, GFAST, true>(lds, K, L, tw, pre, sink);
    else
        lds_stage<R, INV, GFAST, false>(lds, K, L, tw, pre, sink);
}

// MAXR: largest radix this kernel variant carries code for.  Register allocation is per
// kernel, so a variant without the radix-15/16 bodies keeps the occupancy of the small ones.
template <int MAXR, bool INV, bool GFAST, class Sink = NoSink>
__device__ __forceinline__ void lds_stage_any(float4 *lds, const StageK K, const LdsLayout &L,
                                              const float2 *__restrict__ tw, TwPre pre, Sink &&sink = Sink{})
{
#define ASX_STAGE_CASE(R) \
    case R: if constexpr (R <= MAXR) lds_stage_r<R, INV, GFAST>(lds, K, L, tw, pre, sink); break;
    switch (K.R) { // wave-uniform
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
    default: lds_stage_r<2, INV, GFAST>(lds, K, L, tw, pre, sink); break;
    }
#undef ASX_STAGE_CASE
}

// Whole transform.  `pre` = tw_prefetch() of the first stage to run (stage 0 forward,
// stage nstages-1 inverse), issued by the caller before it waited for the tile.  Caller has
// filled LDS and called __syncthreads(); on return all lanes see the result (the routine
// ends with a barrier).
template <int MAXR, bool INV, bool GFAST>
__device__ __forceinline__ void lds_fft(float4 *lds, const AsxStages &st, const LdsLayout &L,
                                        const float2 *__restrict__ tw, TwPre pre)
{
    if (!INV) {
        for (int i = 0; i < st.nstages; i++) {
            const TwPre next = tw_prefetch<GFAST>(st, i + 1, L, tw);
            lds_stage_any<MAXR, false, GFAST>(lds, stage_k(st, i), L, tw, pre);
            pre = next;
            __syncthreads();
        }
    } else {
        for (int i = st.nstages - 1; i >= 0; i--) {
            const TwPre next = tw_prefetch<GFAST>(st, i - 1, L, tw);
            lds_stage_any<MAXR, true, GFAST>(lds, stage_k(st, i), L, tw, pre);
            pre = next;
            __syncthreads();
        }
    }
}

// ---- two consecutive stages without a block barrier between them ---------------------------------
// DIF stages I and I+1 of a schedule both stay inside the sub-blocks of length ns_I ("units").  When
// their radices are equal, a unit has the same number Q = ns_I / R of butterflies in both stages, so a
// WAVE that owns whole units (64 / Q of them, Q lanes each) can run both stages on them back to back:
// LDS operations of one wave execute in order, nothing another wave touches is involved, and the
// s_barrier between the two stages (with the wait for the slowest of the block's waves) disappears.
// Forward runs stage I then I+1, inverse I+1 then I.  Units are numbered over (pair, sub-block);
// GFAST walks the pair index first, like lds_stage.
template <class S, int I> struct WavePair {
    static constexpr StageK K0 = S::stage(I), K1 = S::stage(I + 1);
    static constexpr int R = K0.R, Q = K0.q, UPW = Q <= 64 ? 64 / Q : 0, NSUB = S::n / K0.ns;
    static constexpr bool ok = (I + 1 < S::nstages) && K0.R == K1.R && Q >= 1 && Q <= 64;
};
template <class S, int I> constexpr bool wave_pair_ok()
{
    if constexpr (I + 1 < S::nstages) return WavePair<S, I>::ok;
    else return false;
}

template <class S, int I, bool GFAST>
__device__ __forceinline__ TwPre tw_prefetch_wavepair(const LdsLayout &L, const float2 *__restrict__ tw)
{
    using WP = WavePair<S, I>;
    TwPre pre{ make_float2(1.f, 0.f), make_float2(1.f, 0.f) };
    const int lane = L.tid & 63;
    if (WP::K0.q == 1 || lane >= WP::UPW * WP::Q) return pre;
    const int j = lane % WP::Q;
    const int tj = j * WP::K0.twmul;
    pre.w1 = tw[tj];
    if (WP::R > 4) pre.w4 = tw[4 * tj];
    return pre;
}

template <class S, int I, bool INV, bool GFAST>
__device__ __forceinline__ void lds_stage_wavepair(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre)
{
    using WP = WavePair<S, I>;
    constexpr int R = WP::R, Q = WP::Q, UPW = WP::UPW;
    constexpr StageK K0 = WP::K0, K1 = WP::K1;
    const int lane = L.tid & 63, wave = L.tid >> 6, nwaves = L.nthreads >> 6;
    const int nunits = L.ngroups * WP::NSUB;
    const int ul = lane / Q, j = lane - ul * Q;   // unit within the wave, butterfly within the unit
    for (int u0 = wave * UPW; u0 < nunits; u0 += nwaves * UPW) { // wave-uniform trip count
        const int unit = u0 + ul;
        const bool active = (ul < UPW) && (unit < nunits);
        int g, b;
        if (GFAST) { g = unit & (L.ngroups - 1); b = unit >> L.log_ngroups; }
        else { g = unit / WP::NSUB; b = unit - g * WP::NSUB; }
        float4 *base = lds + g * L.group_stride + (b * K0.ns) * L.elem_stride;
        // stage I: butterfly j of the unit, legs Q elements apart; stage I+1: butterfly j, sub-block j of the unit
        float4 *p0 = base + j * L.elem_stride;
        const int b1 = j / K1.q, j1 = j - b1 * K1.q; // stage I+1: sub-block b1 of the unit, position j1
        float4 *p1 = base + (b1 * K1.ns + j1) * L.elem_stride;
        const int step0 = K0.q * L.elem_stride, step1 = K1.q * L.elem_stride;
        auto run = [&](auto WHICH) __attribute__((always_inline)) {
            constexpr bool first = decltype(WHICH)::value == 0; // stage I (twiddled unless q == 1)
            constexpr StageK K = first ? K0 : K1;
            float4 *p = first ? p0 : p1;
            const int step = first ? step0 : step1;
            if (active) {
                Cx2 v[R];
                static_for<0, R>([&](auto T) __attribute__((always_inline)) { v[T] = lds_get(p + T * step); });
                if constexpr (K.q == 1) {
                    Bfly<R, INV>::run(v);
                } else {
                    float2 w1 = pre.w1, w4 = pre.w4;
                    if (!first) { // stage I+1 with q > 1: its own twiddles (position inside its sub-block)
                        w1 = tw[j1 * K.twmul];
                        if constexpr (R > 4) w4 = tw[4 * j1 * K.twmul];
                    }
                    float2 tww[R];
                    stage_twiddles_from<R>(w1, w4, tww);
                    if constexpr (!INV) {
                        Bfly<R, false>::run(v);
                        static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tww[U]); });
                    } else {
                        static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = mulwc(v[U], tww[U]); });
                        Bfly<R, true>::run(v);
                    }
                }
                static_for<0, R>([&](auto T) __attribute__((always_inline)) { lds_put(p + T * step, v[T]); });
            }
        };
        if constexpr (!INV) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
        // same wave: the second stage reads what the first wrote.  LDS executes a wave's operations in order;
        // the explicit wait for the writes costs one LDS latency per transform and does not lean on that.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (!INV) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
    }
}

// The same with a compile-time schedule S = Sched<n, radices...>: the stage loop is unrolled and
// every stage is instantiated for its own radix and geometry only.
// Execution plan of a compile-time schedule: stage I and I+1 run as a wave pair (no barrier between
// them) for the LAST such I (ASX_WAVEPAIR, default on); every other stage is an ordinary block-wide stage.
#ifndef ASX_WAVEPAIR
#define ASX_WAVEPAIR 1
#endif
// HEAD: the last stage to execute is left to lds_last_stage_static and must stay a block-wide stage.
template <class S, bool INV = false, bool HEAD = false, bool GFAST = false> constexpr int wave_pair_index()
{
    int r = -1;
    // Row kernels only.  In a column tile (GFAST) a wave that owns whole units has its lanes 40 slots apart
    // in the innermost stage: 8-way bank conflicts, k_inv_cols 0.46 -> 0.69 ms [measured].
    if (GFAST) return -1;
#if ASX_WAVEPAIR
    if constexpr (S::nstages >= 2) { if (wave_pair_ok<S, 0>()) r = 0; }
    if constexpr (S::nstages >= 3) { if (wave_pair_ok<S, 1>()) r = 1; }
    if constexpr (S::nstages >= 4) { if (wave_pair_ok<S, 2>()) r = 2; }
#endif
    constexpr int last = INV ? 0 : S::nstages - 1;
    if (HEAD && r >= 0 && (last == r || last == r + 1)) r = -1;
    return r;
}
// twiddle prefetch of stage i as it will be executed (wave-pair mapping or block mapping)
template <class S, int i, bool GFAST, bool INV = false, bool HEAD = false>
__device__ __forceinline__ TwPre tw_prefetch_exec(const LdsLayout &L, const float2 *__restrict__ tw)
{
    constexpr int WPI = wave_pair_index<S, INV, HEAD, GFAST>();
    if constexpr (i < 0 || i >= S::nstages) return TwPre{ make_float2(1.f, 0.f), make_float2(1.f, 0.f) };
    else if constexpr (WPI >= 0 && (i == WPI || i == WPI + 1)) return tw_prefetch_wavepair<S, WPI, GFAST>(L, tw);
    else return tw_prefetch_k<GFAST>(S::stage(i), L, tw);
}
// first stage to run of a compile-time schedule
template <class S, bool INV, bool GFAST, bool HEAD = false>
__device__ __forceinline__ TwPre tw_prefetch_first(const LdsLayout &L, const float2 *__restrict__ tw)
{
    return tw_prefetch_exec<S, (INV ? S::nstages - 1 : 0), GFAST, INV, HEAD>(L, tw);
}

// Runs the stages [FIRST_STEP, FIRST_STEP + NSTEPS) of the execution order (forward: stage = step;
// inverse: stage = nstages - 1 - step), each followed by a barrier; returns the prefetch of the next step.
template <class S, bool INV, bool GFAST, int NSTEPS, bool HEAD = false, int FIRST = 0>
__device__ __forceinline__ TwPre lds_fft_static_steps(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre)
{
    constexpr int WPI = wave_pair_index<S, INV, HEAD, GFAST>();
    static_for<FIRST, NSTEPS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = INV ? S::nstages - 1 - decltype(I)::value : decltype(I)::value;
        constexpr int inext = INV ? i - 1 : i + 1;
        constexpr bool second_of_pair = WPI >= 0 && (INV ? i == WPI : i == WPI + 1);
        constexpr bool first_of_pair = WPI >= 0 && (INV ? i == WPI + 1 : i == WPI);
        if constexpr (second_of_pair) {
            // already done together with its partner
        } else if constexpr (first_of_pair) {
            constexpr int iafter = INV ? WPI - 1 : WPI + 2;
            const TwPre next = tw_prefetch_exec<S, iafter, GFAST, INV, HEAD>(L, tw);
            lds_stage_wavepair<S, WPI, INV, GFAST>(lds, L, tw, pre);
            pre = next;
            __syncthreads();
        } else {
            constexpr StageK K = S::stage(i);
            const TwPre next = tw_prefetch_exec<S, inext, GFAST, INV, HEAD>(L, tw);
            lds_stage<K.R, INV, GFAST, K.q == 1>(lds, K, L, tw, pre);
            pre = next;
            __syncthreads();
        }
    });
    return pre;
}

// The same with a compile-time schedule S = Sched<n, radices...>: the stage loop is unrolled and
// every stage is instantiated for its own radix and geometry only.
template <class S, bool INV, bool GFAST>
__device__ __forceinline__ void lds_fft_static(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre)
{
    (void)lds_fft_static_steps<S, INV, GFAST, S::nstages>(lds, L, tw, pre);
}

// ---- transforms whose LAST stage is consumed from registers (Sink) -------------------------------
// lds_fft*_head runs every stage but the last one to execute (stage 0 of an inverse, stage nstages-1
// of a forward transform) and returns the twiddle prefetch of that last stage; the caller then runs
// lds_last_stage*(..., sink) as often as it likes: the stage reads LDS and writes nothing.
template <int MAXR, bool INV, bool GFAST>
__device__ __forceinline__ TwPre lds_fft_head(float4 *lds, const AsxStages &st, const LdsLayout &L,
                                              const float2 *__restrict__ tw, TwPre pre)
{
    if (!INV) {
        for (int i = 0; i < st.nstages - 1; i++) {
            const TwPre next = tw_prefetch<GFAST>(st, i + 1, L, tw);
            lds_stage_any<MAXR, false, GFAST>(lds, stage_k(st, i), L, tw, pre);
            pre = next;
            __syncthreads();
        }
    } else {
        for (int i = st.nstages - 1; i >= 1; i--) {
            const TwPre next = tw_prefetch<GFAST>(st, i - 1, L, tw);
            lds_stage_any<MAXR, true, GFAST>(lds, stage_k(st, i), L, tw, pre);
            pre = next;
            __syncthreads();
        }
    }
    return pre;
}
template <int MAXR, bool INV, bool GFAST, class Sink>
__device__ __forceinline__ void lds_last_stage(float4 *lds, const AsxStages &st, const LdsLayout &L,
                                               const float2 *__restrict__ tw, TwPre pre, Sink &&sink)
{
    if (st.nstages < 1) {
        // a one-point transform (M1 = 1): the tile itself is the result
        for (int w = L.tid; w < L.ngroups * st.n; w += L.nthreads) {
            int g, e;
            if (GFAST) { g = w & (L.ngroups - 1); e = w >> L.log_ngroups; }
            else { g = w / st.n; e = w - g * st.n; }
            Cx2 v[1] = { lds_get(lds + g * L.group_stride + e * L.elem_stride) };
            sink(std::integral_constant<int, 1>{}, v, g, e, 1);
        }
        return;
    }
    lds_stage_any<MAXR, INV, GFAST>(lds, stage_k(st, INV ? 0 : st.nstages - 1), L, tw, pre, sink);
}

template <class S, bool INV, bool GFAST>
__device__ __forceinline__ TwPre lds_fft_static_head(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre)
{
    // HEAD: the stage left for lds_last_stage_static is never half of a wave pair (wave_pair_index)
    return lds_fft_static_steps<S, INV, GFAST, S::nstages - 1, true>(lds, L, tw, pre);
}
// The head with its FIRST stage fed by `source` instead of LDS (block-wide stage, never a wave pair: GFAST
// kernels have none); the stage writes its outputs to LDS, so the caller needs no fill phase and no barrier
// before it.  Returns the prefetch of the last stage like lds_fft_static_head.
template <class S, bool INV, bool GFAST, class Source>
__device__ __forceinline__ TwPre lds_fft_static_head_fed(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre,
                                                         Source &&source)
{
    static_assert(wave_pair_index<S, INV, true, GFAST>() < 0, "fed first stage with wave pairs is not supported");
    static_assert(S::nstages >= 2, "needs a last stage of its own");
    constexpr int i0 = INV ? S::nstages - 1 : 0, i1 = INV ? i0 - 1 : i0 + 1;
    constexpr StageK K = S::stage(i0);
    const TwPre next = tw_prefetch_exec<S, i1, GFAST, INV, true>(L, tw);
    lds_stage<K.R, INV, GFAST, K.q == 1>(lds, K, L, tw, pre, NoSink{}, source);
    __syncthreads();
    return lds_fft_static_steps<S, INV, GFAST, S::nstages - 1, true, 1>(lds, L, tw, next);
}
template <class S, bool INV, bool GFAST, class Sink>
__device__ __forceinline__ void lds_last_stage_static(float4 *lds, const LdsLayout &L, const float2 *__restrict__ tw, TwPre pre,
                                                      Sink &&sink)
{
    constexpr StageK K = S::stage(INV ? 0 : S::nstages - 1);
    lds_stage<K.R, INV, GFAST, K.q == 1>(lds, K, L, tw, pre, sink);
}
