// csrc/lds_fft.h — in-place mixed-radix {2,3,4,5} transforms inside LDS for gfx950.
//
// Forward = decimation in frequency (Gentleman-Sande): natural order in,
// digit-reversed order out.  Inverse = decimation in time with the stages run
// backwards: digit-reversed in, natural order out.  Both are in place, so a
// tile needs ONE LDS copy (no Stockham ping-pong), each butterfly touches only
// its own R slots (no intra-stage hazard) and the permuted spectrum is never
// un-permuted: the spectral product only needs X and Y in the SAME order, and
// index tables (pos2_of_k2) locate the k <-> M-k partners.
// Index algebra is modelled and tested in tests/model_fourstep.py.
//
// Replaces FFTW's r2c/c2r kernels as used at src/cross_correlation.c:34-39,237-239.
#pragma once

#include "asx_internal.h"

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

// ---- radix butterflies: v <- DFT_R(v) (INV: the conjugate kernel, unnormalised) ----
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

// One stage over `nbatch` transforms laid out as
//   element e of transform c  ->  lds[c*batch_stride + e*elem_stride]
// BATCH_FASTEST: consecutive lanes walk the batch index first (column tiles:
// nbatch = T is a power of two and elem_stride = T, so a wave touches
// contiguous LDS); otherwise consecutive lanes walk butterflies (row tiles).
template <int R, bool INV, bool BATCH_FASTEST>
__device__ __forceinline__ void lds_stage(float2 *lds, const AsxStages &st, int i, int nbatch,
                                          int log_nbatch, int elem_stride, int batch_stride,
                                          const float2 *__restrict__ tw)
{
    const int ns = st.ns[i], q = st.q[i], nbf = st.nbf[i], twmul = st.twmul[i];
    const float inv_q = st.inv_q[i], inv_nbf = st.inv_nbf[i];
    const int total = nbatch * nbf;
    const int step = q * elem_stride;
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        int c, bf;
        if (BATCH_FASTEST) {
            c = w & (nbatch - 1);
            bf = w >> log_nbatch;
        } else {
            c = div_exact(w, nbf, inv_nbf, bf);
        }
        int j;
        const int b = div_exact(bf, q, inv_q, j);
        float2 *p = lds + c * batch_stride + (b * ns + j) * elem_stride;
        const int tj = j * twmul;
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; t++) v[t] = p[t * step];
        if (!INV) {
            Bfly<R, false>::run(v);
#pragma unroll
            for (int u = 1; u < R; u++) v[u] = cmul(v[u], tw[u * tj]);
        } else {
#pragma unroll
            for (int u = 1; u < R; u++) v[u] = cmulc(v[u], tw[u * tj]);
            Bfly<R, true>::run(v);
        }
#pragma unroll
        for (int t = 0; t < R; t++) p[t * step] = v[t];
    }
}

template <bool INV, bool BATCH_FASTEST>
__device__ __forceinline__ void lds_stage_any(float2 *lds, const AsxStages &st, int i, int nbatch,
                                              int log_nbatch, int elem_stride, int batch_stride,
                                              const float2 *__restrict__ tw)
{
    switch (st.radix[i]) { // wave-uniform
    case 4: lds_stage<4, INV, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw); break;
    case 5: lds_stage<5, INV, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw); break;
    case 3: lds_stage<3, INV, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw); break;
    default: lds_stage<2, INV, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw); break;
    }
}

// Whole transform.  Caller has filled LDS and called __syncthreads(); on return
// all lanes see the result (the routine ends with a barrier).
template <bool INV, bool BATCH_FASTEST>
__device__ __forceinline__ void lds_fft(float2 *lds, const AsxStages &st, int nbatch, int log_nbatch,
                                        int elem_stride, int batch_stride,
                                        const float2 *__restrict__ tw)
{
    if (!INV) {
        for (int i = 0; i < st.nstages; i++) {
            lds_stage_any<false, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw);
            __syncthreads();
        }
    } else {
        for (int i = st.nstages - 1; i >= 0; i--) {
            lds_stage_any<true, BATCH_FASTEST>(lds, st, i, nbatch, log_nbatch, elem_stride, batch_stride, tw);
            __syncthreads();
        }
    }
}
