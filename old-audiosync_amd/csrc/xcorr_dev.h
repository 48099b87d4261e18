// csrc/xcorr_dev.h -- device helpers shared by the kernel translation units (xcorr_kernels.hip, rlayout.hip).
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


// ---------------------------------------------------------------------------
// column tiles (shared by xcorr_kernels.hip and rlayout.hip)
// ---------------------------------------------------------------------------
// Tile <-> block mapping of the column kernels.  A tile row is T complex values = 8T bytes; L = 16/T tiles
// share a 128-byte L2 line (T = 8: two tiles, 64 bytes each).  Blocks are dealt round-robin over the 8
// XCDs (each with a private L2), so with the identity mapping the pieces of a line are fetched by different
// L2s: the PMC read counters showed exactly 2x the needed bytes at T = 8.  This mapping puts the L tiles of
// a line on blocks b, b + 8, ..., b + 8(L-1) (same XCD, dispatched together), so all but the first hit in
// L2.  Speed only; any placement is correct.  Grids are rounded up to a multiple of 8L blocks.
__device__ __host__ __forceinline__ int col_tiles_per_line_log(int logT) { return logT >= 4 ? 0 : 4 - logT; }
__device__ __forceinline__ int col_tile_of_block(int b, int logT)
{
    const int ll = col_tiles_per_line_log(logT);       // log2 L
    const int grp = 8 << ll;                           // blocks that cover 8 lines
    return (b & ~(grp - 1)) + ((b & 7) << ll) + ((b >> 3) & ((1 << ll) - 1));
}
static inline unsigned col_grid_x(int ntiles, int logT)
{
    const unsigned grp = 8u << col_tiles_per_line_log(logT);
    return ((unsigned)ntiles + grp - 1) / grp * grp;
}

__device__ __forceinline__ LdsLayout col_layout(int T, int logT, int nthreads)
{
    LdsLayout L;
    L.ngroups = T >> 1;
    L.log_ngroups = logT - 1;
    L.elem_stride = T >> 1;
    L.group_stride = 1;
    L.nthreads = nthreads;
    L.tid = threadIdx.x;
    return L;
}
constexpr int asx_ilog2(int v) { return v <= 1 ? 0 : 1 + asx_ilog2(v >> 1); }


// ---------------------------------------------------------------------------
// peak search helpers (src/cross_correlation.c:52-67)
// ---------------------------------------------------------------------------
// key(0) = arr[0] SIGNED (:56), key(i) = fabs(arr[i]) (:59).  A NaN key never wins the strict
// '>' of :60 (so it needs no mapping in a running maximum); a NaN at index 0 is never beaten.
__device__ __forceinline__ float peak_key_of(float value, uint32_t idx)
{
    const float a = fabsf(value);
    const float z = (value != value) ? INFINITY : (value + 0.0f); // -0.0 -> +0.0 so it ties with |0|
    return idx == 0u ? z : a;
}
// The same when the transforms ran on (source - mean) (second look at a pair with a large offset in both tracks,
// second_look, asx_api.hip): r[k] = value + c with c = mean * sum(sample), the same for every k.  Keys are taken RELATIVE to
// |c| -- key' = |r| - |c|, lag 0: r - |c| -- so that float32 keeps the differences between lags when |c| >> |value|;
// a common shift changes neither the order of the keys nor the width of the near-maximum window.
__device__ __forceinline__ float peak_key_shifted(float value, uint32_t idx, double c)
{
    if (value != value) return idx == 0u ? INFINITY : value;
    const double r = (double)value + c;
    return (float)((idx == 0u ? r : fabs(r)) - fabs(c)) + 0.0f;
}
__device__ __forceinline__ asx_peak_t peak_pack_key(float key, uint32_t idx)
{
    if (key != key) key = -INFINITY;
    uint32_t b = __float_as_uint(key);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((asx_peak_t)b << 32) | (asx_peak_t)(0xFFFFFFFFu - idx);
}

__device__ __forceinline__ asx_peak_t peak_max(asx_peak_t a, asx_peak_t b) { return a > b ? a : b; }

__device__ __forceinline__ asx_peak_t wave_peak_max(asx_peak_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const asx_peak_t o = __shfl_xor(v, off, 64);
        v = peak_max(v, o);
    }
    return v;
}

// block-wide max; result valid in thread 0.  `scratch` = one entry per wave of the block.
__device__ __forceinline__ asx_peak_t block_peak_max(asx_peak_t v, asx_peak_t *scratch)
{
    v = wave_peak_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nwaves = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nwaves; w++) v = peak_max(v, scratch[w]);
    }
    return v;
}

// Wave-wide maximum of non-negative floats without LDS traffic: four DPP steps leave every lane of
// a 16-lane row with the row's maximum, three readlanes combine the rows.  (__shfl_xor goes through
// ds_bpermute: one LDS round trip per step.)
__device__ __forceinline__ float wave_max_nonneg(float v)
{
#define ASX_DPP_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xF, 0xF, true)))
    ASX_DPP_MAX(0xB1);  // quad_perm [1,0,3,2]
    ASX_DPP_MAX(0x4E);  // quad_perm [2,3,0,1]
    ASX_DPP_MAX(0x141); // row_half_mirror
    ASX_DPP_MAX(0x140); // row_mirror
#undef ASX_DPP_MAX
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}


__device__ __forceinline__ float peak_key(asx_peak_t v)
{
    uint32_t b = (uint32_t)(v >> 32);
    b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint32_t peak_index(asx_peak_t v) { return 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull); }
// Everything at or above this key is "as large as the maximum" for float32 transforms: b2 = 2B,
// B = the bound on |float32 r[k] - exact r[k]| (asx_internal.h).  If the exact maximum is at k*, then
// key32(k*) >= exact(k*) - B >= exact(kmax32) - B >= key32(kmax32) - 2B.
__device__ __forceinline__ float near_max_threshold(float kmax, float b2) { return kmax - b2; }

// Append one near-maximum lag to the pair's candidate list (called from divergent code: the lanes of
// the wave that are here together take ONE slot range with one atomic).
__device__ __forceinline__ void cand_append(const AsxPeakWs &W, size_t pair, uint32_t idx, float key)
{
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&W.cand_n[pair], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < W.cap) {
        AsxCand c; c.idx = idx; c.key = key;
        W.cand[pair * (size_t)W.cap + slot] = c;
    }
}

// ---------------------------------------------------------------------------
// lag wrap + segments (src/cross_correlation.c:256-271)
// ---------------------------------------------------------------------------
__device__ __forceinline__ AsxSeg make_seg(uint32_t peak, uint32_t N)
{
    AsxSeg s;
    s.peak = peak;
    s.flags = 0;
    if (peak >= N) {
        // src/cross_correlation.c:256-263: lag = (lag % N) - N; source[0 .. N+lag), sample[-lag .. N)
        const long long l = (long long)(peak % N) - (long long)N;
        s.lag = l;
        s.src_off = 0;
        s.smp_off = (uint32_t)(-l);
        s.len = (uint32_t)((long long)N + l);
    } else {
        // :264-271: source[lag .. lag+N), sample[0 .. N)
        s.lag = (long long)peak;
        s.src_off = peak;
        s.smp_off = 0;
        s.len = N;
    }
    return s;
}

// ---------------------------------------------------------------------------
// The spectral Pearson form's decision for one pair (pearson_spectral.hip), from what k_pearson_prep left: the blocks' shares of the window
// sums (added here in block order: the same bits whoever adds them), r[peak] and the bound on its error.  Pure: every block of
// k_pearson_partial and k_pearson_final_spec work it out for themselves and agree.
//   mode   ASX_PM_FAST / CORR / DIRECT;   work = what k_pearson_partial walks: nothing, the wrap-around part, or the segment itself
// ---------------------------------------------------------------------------
struct AsxSpecPick {
    int mode;
    double n, Sx, Sxx, Sy, Syy, r, bound;
    AsxSeg work;
};
__device__ __forceinline__ AsxSpecPick asx_spec_pick(const AsxSeg s, const double *__restrict__ part, int nb, const double *__restrict__ hdr,
                                                     double tol, uint32_t N)
{
    AsxSpecPick d;
    d.mode = ASX_PM_DIRECT;
    d.n = (double)s.len;
    d.Sx = d.Sxx = d.Sy = d.Syy = d.r = 0.0;
    d.bound = INFINITY;
    d.work = s;
    if (hdr[2] == 0.0) {
        double v[4];
        for (int k = 0; k < 4; k++) {
            double t = part[k];
            for (int b = 1; b < nb; b++) t += part[b * 4 + k];
            v[k] = t;
        }
        const double n = d.n, Sx = v[0], Sxx = v[1], Sy = v[2], Syy = v[3], rb = hdr[1];
        d.Sx = Sx; d.Sxx = Sxx; d.Sy = Sy; d.Syy = Syy; d.r = hdr[0];
        const double A = Sxx - Sx * Sx / n, B = Syy - Sy * Sy / n;
        if (A > 0.0 && B > 0.0) {
            // The float32 band sums of k_fwd_cols_r (rlayout.hip), u = 2^-24.  Sum of squares of a band: eight fused multiply-adds per
            // row pair, the RPQ <= 5 pairs added as a tree (<= 3 levels), two DPP additions: <= 13 u relative.  Plain sum: a tree of depth 3
            // per row pair, RPQ additions down the pairs, the same two DPP steps: <= 10 u of sum|x| <= 10 u sqrt(n Sxx).  es = 16 u covers
            // both (round 5 summed the squares as one chain of 8 RPQ + 2 = 42 roundings, which it did not: ADVICE r5; the kernel changed,
            // not the constant).  The cells are added in float64 (k_pearson_prep).  First order: products of errors are dropped.
            const double es = 16.0 * 5.9604645e-8;
            const double dSx = es * sqrt(n * Sxx), dSy = es * sqrt(n * Syy);
            const double dC = rb + (fabs(Sy) * dSx + fabs(Sx) * dSy) / n;
            const double dA = es * Sxx + 2.0 * fabs(Sx) / n * dSx, dB = es * Syy + 2.0 * fabs(Sy) / n * dSy;
            d.bound = dC / sqrt(A * B) + 0.5 * (dA / A + dB / B); // |coefficient| <= 1
            if (d.bound <= tol) {
                if (s.peak < N) d.mode = ASX_PM_FAST;
                else if (N - s.len < s.len) d.mode = ASX_PM_CORR; // |lag| products instead of L
            }
        }
    }
    if (d.mode == ASX_PM_FAST) d.work.len = 0;
    else if (d.mode == ASX_PM_CORR) { d.work.src_off = s.peak; d.work.smp_off = 0; d.work.len = N - s.len; }
    return d;
}

// ---------------------------------------------------------------------------
// Pearson partial statistics and their exact pairwise merge (Chan et al.); see k_pearson_partial
// ---------------------------------------------------------------------------
struct PStat {
    double n, mx, my, mxx, myy, cxy;
};
__device__ __forceinline__ PStat pstat_merge(const PStat A, const PStat B)
{
    if (B.n == 0.0) return A;
    if (A.n == 0.0) return B;
    const double n = A.n + B.n;
    const double dx = B.mx - A.mx, dy = B.my - A.my;
    const double fb = B.n / n, w = A.n * fb;
    PStat R;
    R.n = n;
    R.mx = A.mx + dx * fb;
    R.my = A.my + dy * fb;
    R.mxx = (A.mxx + B.mxx) + (dx * dx) * w;
    R.myy = (A.myy + B.myy) + (dy * dy) * w;
    R.cxy = (A.cxy + B.cxy) + (dx * dy) * w;
    return R;
}
__device__ __forceinline__ PStat pstat_wave_merge(PStat v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        PStat o;
        o.n = __shfl_xor(v.n, off, 64);
        o.mx = __shfl_xor(v.mx, off, 64);
        o.my = __shfl_xor(v.my, off, 64);
        o.mxx = __shfl_xor(v.mxx, off, 64);
        o.myy = __shfl_xor(v.myy, off, 64);
        o.cxy = __shfl_xor(v.cxy, off, 64);
        // the lane with the lower index is always "A": both partners then compute the same merge
        const bool lower = (threadIdx.x & off) == 0;
        v = lower ? pstat_merge(v, o) : pstat_merge(o, v);
    }
    return v;
}
