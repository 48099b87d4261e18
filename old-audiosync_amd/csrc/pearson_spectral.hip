// csrc/pearson_spectral.hip -- the Pearson coefficient (src/cross_correlation.c:74-116) WITHOUT a second pass over the inputs.
//
// The reference reads both segments twice (means, then centred sums).  Everything those sums are made of has already
// passed through this path's kernels by the time the lag is known:
//
//   coefficient = (Sxy - Sx Sy / n) / sqrt((Sxx - Sx^2 / n) (Syy - Sy^2 / n))         (the same formula, expanded)
//
//   lag >= 0   source[lag .. lag + N) against sample[0 .. N)   (:264-271)
//              Sxy = r[peak] EXACTLY the sum the transforms computed (no wrap-around: lag + N <= 2N);
//   lag <  0   source[0 .. L) against sample[N - L .. N), L = N + lag   (:256-263)
//              Sxy = r[peak] - sum_{n < -lag} source[peak + n] sample[n]   (the part of the circular sum that did not wrap):
//              |lag| products instead of L, worth it when |lag| < L;
//   Sx, Sxx, Sy, Syy   window sums: k_fwd_cols_r leaves {sum, sum of squares} of every (band of AsxDev::band_rows rows) x (tile)
//              of the [2 M1][M2] sample matrix (AsxPeakWs::band) from the loads it makes anyway; a window is whole bands (their
//              cells summed here, <= 144 KB per track) plus two band edges (at most 2 * band_rows * M2 samples, read here).
//
// r[peak] is the float32 transforms' value (k_inv_cols_r leaves the SIGNED value of each tile's best lag, AsxPeakWs::tile_peak),
// or the exact one when the pair's near-ties were re-evaluated (k_refine_dots).  Its error is bounded by the SAME bound B that
// guards the lag (asx_internal.h); the band sums are float32 sums of 128 - 160 terms: the squares eight fused multiply-adds per row pair and a
// tree over the pairs (<= 13 u relative), the plain sums a short tree (<= 10 u of sum |x|): both inside the 16 u asx_spec_pick uses.  asx_spec_pick (xcorr_dev.h; every block of k_pearson_partial runs
// it on what k_pearson_prep left, block 0 records the mode for k_pearson_final_spec) turns both into a first-order
// bound on the coefficient's error and keeps the spectral form only when that bound is below `tol` (1e-5, north_star's tolerance);
// otherwise -- quiet windows of a loud track, offsets, short segments -- the pair takes the reference's own reduction
// (k_pearson_partial over the segment, ASX_PM_DIRECT).  Float32 inputs on real-column plans only; the double ABI,
// the second look and the packed-sample kernels keep the direct reduction.
//
// Per pair the pass reads: FAST <= 2 band edges of each track (<= 0.3 MB at N = 1 440 000 instead of 11.5 MB); CORR + 8 |lag|
// bytes; DIRECT 8 L bytes as before.
#include "asx_internal.h"
#include "xcorr_dev.h"


static_assert(ASX_PREP_BLOCKS <= ASX_PREP_BLOCKS_MAX, "AsxSpecWs::part is sized for ASX_PREP_BLOCKS_MAX blocks per pair");
#ifndef ASX_PREP_THREADS
#define ASX_PREP_THREADS 1024 // per block of k_pearson_prep on the long tracks (ASX_PREP_BLOCKS blocks per pair)
#endif

namespace {

struct Acc2 {
    double s1, s2;
};

__device__ __forceinline__ void acc1(Acc2 &a, float f)
{
    const double v = (double)f;
    a.s1 += v;
    a.s2 = fma(v, v, a.s2);
}
__device__ __forceinline__ void acc4(Acc2 &a, float4 v) { acc1(a, v.x); acc1(a, v.y); acc1(a, v.z); acc1(a, v.w); }

// sum and sum of squares of track[lo .. hi) (float64 from the float32 samples), this thread's share: 16-byte loads over the
// aligned middle (the track starts on a 16-byte boundary: 2N and N are multiples of four), the ragged ends by single lanes.
// Four loads in flight per thread, each into its own accumulator (a fixed order all the same: the four are added at the end).
// NTP = the threads that share the range (all blocks of the pair), tid = this thread's index among them.
template <uint32_t NTP> __device__ __forceinline__ void direct_range(const float *__restrict__ x, uint32_t lo, uint32_t hi, Acc2 &a, uint32_t tid)
{
    if (lo >= hi) return;
    const uint32_t lo4 = (lo + 3u) & ~3u, hi4 = hi & ~3u;
    if (lo4 >= hi4) { // no aligned quad inside
        for (uint32_t i = lo + tid; i < hi; i += NTP) acc1(a, x[i]);
        return;
    }
    if (tid < lo4 - lo) acc1(a, x[lo + tid]);
    if (tid < hi - hi4) acc1(a, x[hi4 + tid]);
    const float4 *q = reinterpret_cast<const float4 *>(x);
    const uint32_t end = hi4 >> 2;
    uint32_t i = (lo4 >> 2) + tid;
    Acc2 b{ 0.0, 0.0 }, c{ 0.0, 0.0 }, d{ 0.0, 0.0 };
    for (; i + 3u * NTP < end; i += 4u * NTP) {
        const float4 v0 = q[i], v1 = q[i + NTP], v2 = q[i + 2u * NTP], v3 = q[i + 3u * NTP];
        acc4(a, v0); acc4(b, v1); acc4(c, v2); acc4(d, v3);
    }
    for (; i < end; i += NTP) acc4(a, q[i]);
    a.s1 += (b.s1 + c.s1) + d.s1;
    a.s2 += (b.s2 + c.s2) + d.s2;
}

// This thread's share of the window sums of one track over [lo, hi): whole bands from the band sums, the two edges from the samples.
//   band: [ntiles][nbands] {sum, sum of squares} as k_fwd_cols_r left them (every block its own run); a band = gs consecutive samples.
//   The (tile, band) cells of the window are dealt to the threads in order -- consecutive lanes walk the bands of a tile --
//   and added in float64: a fixed order.
template <uint32_t NTP> __device__ __forceinline__ Acc2 window_share(const float *__restrict__ x, const float2 *__restrict__ band, uint32_t gs, int ntiles,
                                             int nbands, uint32_t lo, uint32_t hi, uint32_t tid)
{
    Acc2 a{ 0.0, 0.0 };
    const uint32_t ba = (lo + gs - 1) / gs, bb = hi / gs;
    if (ba < bb) {
        direct_range<NTP>(x, lo, ba * gs, a, tid);
        direct_range<NTP>(x, bb * gs, hi, a, tid);
        const uint32_t w = bb - ba, cells = w * (uint32_t)ntiles;
        const uint32_t dq = NTP / w, dr = NTP - dq * w; // one step of the cell index, as (tiles, bands)
        uint32_t t = tid / w, b = tid - t * w;
        Acc2 e{ 0.0, 0.0 };
        uint32_t i = tid;
        for (; i + NTP < cells; i += 2u * NTP) { // two cells in flight
            uint32_t t2 = t + dq, b2 = b + dr;
            if (b2 >= w) { b2 -= w; t2++; }
            const float2 v = band[(size_t)t * nbands + ba + b], v2 = band[(size_t)t2 * nbands + ba + b2];
            a.s1 += (double)v.x; a.s2 += (double)v.y;
            e.s1 += (double)v2.x; e.s2 += (double)v2.y;
            t = t2 + dq; b = b2 + dr;
            if (b >= w) { b -= w; t++; }
        }
        if (i < cells) {
            const float2 v = band[(size_t)t * nbands + ba + b];
            a.s1 += (double)v.x; a.s2 += (double)v.y;
        }
        a.s1 += e.s1; a.s2 += e.s2;
    } else {
        direct_range<NTP>(x, lo, hi, a, tid);
    }
    return a;
}

// block-wide sums of four numbers, fixed order (lane tree, then waves in order); valid in every thread
template <int NTP> __device__ __forceinline__ void block_sum4(double (&v)[4], double (*red)[NTP / 64])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] += __shfl_xor(v[k], off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 4; k++) red[k][wave] = v[k];
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        double t = red[k][0];
        for (int w = 1; w < NTP / 64; w++) t += red[k][w];
        v[k] = t;
    }
}

} // namespace

// grid (NB, npairs), NTP threads (1024 for the long tracks -- up to 77 000 samples of band edges and 18 000 cells per pair -- 256 below:
// 1024 pairs of N = 144 000 took 61 us with 1024-thread blocks, the sixteen-wave fold of a few thousand numbers).  Picks the winner among the re-evaluated near-ties (if any) and leaves the pair's window-sum shares and r[peak] (AsxSpecWs).
// NB > 1 (the long tracks): NB blocks share a pair's cells and edges -- NB * NTP threads dealt the same way -- and each leaves its SHARE of the
// four sums; the kernels behind this one add the shares in block order (asx_spec_pick): the same bits whoever adds them, and NB is a
// constant of the plan (the same pair takes the same tree alone, in a batch, on another shard).  No merged record, hence no fence and no
// ticket: profiles/r5_experiments/18_* (one block per pair: 25 us for a single pair behind ONE block's loads) and 22_*.
template <int NTP, int NB> __global__ __launch_bounds__(NTP) void k_pearson_prep(const AsxDev *__restrict__ Pp, const float *__restrict__ src,
                                                                    const float *__restrict__ smp, AsxPeakWs W, AsxSpecWs S,
                                                                    AsxSeg *__restrict__ seg)
{
    __shared__ double red[4][NTP / 64];
    __shared__ double s_exact;
    __shared__ int s_have_exact;
    __shared__ double rkey[NTP / 64], rval[NTP / 64];
    __shared__ uint32_t ridx[NTP / 64];
    __shared__ AsxSeg s_seg;
    const size_t pair = blockIdx.y;
    const uint32_t blk = blockIdx.x, gtid = blk * (uint32_t)NTP + threadIdx.x;
    const uint32_t N = Pp->N;
    const int M2 = Pp->M2, nbands = Pp->nbands;
    const uint32_t gs = (uint32_t)Pp->band_rows * (uint32_t)M2;
    const asx_peak_t best = W.pairmax[pair];
    // ---- k_refine_pick's part (xcorr_kernels.hip; this kernel stands in for it in the spectral form: one launch less): the
    // reference's max_abs_index rule (src/cross_correlation.c:52-67) on the exact values of the re-evaluated near-ties --
    // key(0) = r[0] signed, key(i) = |r[i]|, largest key, smallest lag among equal keys, a NaN never wins unless at lag 0 --
    // and, kept here, the winner's exact SIGNED value: the cross term of the coefficient
    const uint32_t nref = W.refine_n[pair];
    if (nref >= 2u) { // block-uniform
        double bk = -INFINITY, bv = 0.0;
        uint32_t bi = 0xFFFFFFFFu;
        for (uint32_t i = threadIdx.x; i < nref; i += NTP) {
            const uint32_t idx = W.refine_idx[pair * (size_t)W.cap + i];
            const double v = W.refine_val[pair * (size_t)W.cap + i];
            double key;
            if (idx == 0u) key = (v != v) ? (double)INFINITY : v + 0.0;
            else { key = fabs(v); if (key != key) key = -(double)INFINITY; }
            if (key > bk || (key == bk && idx < bi)) { bk = key; bi = idx; bv = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ok = __shfl_xor(bk, off, 64), ov = __shfl_xor(bv, off, 64);
            const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
            if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; bv = ov; }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { rkey[wave] = bk; ridx[wave] = bi; rval[wave] = bv; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NTP / 64; w++)
                if (rkey[w] > bk || (rkey[w] == bk && ridx[w] < bi)) { bk = rkey[w]; bi = ridx[w]; bv = rval[w]; }
            AsxSeg sg = seg[pair];
            if (bi != 0xFFFFFFFFu) { sg = make_seg(bi, N); if (blk == 0) seg[pair] = sg; } // every block of the pair finds the same winner
            s_seg = sg;
            s_exact = bv;
            s_have_exact = bi != 0xFFFFFFFFu;
        }
    } else if (threadIdx.x == 0) {
        s_seg = seg[pair];
        s_have_exact = 0;
    }
    __syncthreads();
    const AsxSeg s = s_seg;
    // A pair the transforms had nothing to say about (silent or NaN track: no maximum), an empty segment, or a lag that is still
    // the float32 placeholder of an overflowed list (the second look redoes it): the direct reduction, whatever it yields.
    const bool direct = best == 0 || s.len == 0 || (s.flags & ASX_SEG_INEXACT) != 0;
    if (blk == 0 && threadIdx.x == 0) {
        // the header: r[peak] in the plain-sum scale and the bound on its error
        double r = 0.0, rb = 0.0;
        if (!direct) {
            if (s_have_exact) r = s_exact;
            else {
                const int T = Pp->T;
                r = (double)W.tile_peak[pair * (size_t)(M2 / T) + (s.peak % (uint32_t)M2) / (uint32_t)T] / (double)Pp->F;
                rb = 0.5 * (double)W.bound2[pair] / (double)Pp->F; // bound2 = 2B in the device's scale (F times the plain sum)
            }
        }
        double *hdr = S.hdr + pair * ASX_SPEC_HDR;
        hdr[0] = r; hdr[1] = rb; hdr[2] = direct ? 1.0 : 0.0; hdr[3] = 0.0;
    }
    if (direct) return; // block-uniform, and the same in every block of the pair: nobody reads its shares
    const float *x = src + pair * (size_t)(2u * N), *y = smp + pair * (size_t)N;
    const int ntiles = Pp->ntiles;
    const float2 *bx = W.band + (size_t)pair * 2 * ntiles * nbands, *by = bx + (size_t)ntiles * nbands;
    const Acc2 ax = window_share<(uint32_t)NTP * NB>(x, bx, gs, ntiles, nbands, s.src_off, s.src_off + s.len, gtid);
    const Acc2 ay = window_share<(uint32_t)NTP * NB>(y, by, gs, ntiles, nbands, s.smp_off, s.smp_off + s.len, gtid);
    double v[4] = { ax.s1, ax.s2, ay.s1, ay.s2 };
    block_sum4<NTP>(v, red);
    if (threadIdx.x == 0) {
        double *mine = S.part + (pair * NB + blk) * 4;
        mine[0] = v[0]; mine[1] = v[1]; mine[2] = v[2]; mine[3] = v[3];
    }
}

// grid (npairs), one wave per pair: k_pearson_final (xcorr_kernels.hip) with the two spectral modes in front of it.
__global__ __launch_bounds__(64) void k_pearson_final_spec(const AsxSeg *__restrict__ seg, const double *__restrict__ psums, uint32_t nb,
                                                            AsxSpecWs S, int64_t *__restrict__ lag,
                                                            double *__restrict__ coef, int32_t *__restrict__ ret)
{
    const size_t pair = blockIdx.x;
    const AsxSeg s = seg[pair];
    const AsxSpecPick d = asx_spec_pick(s, S.part + pair * (size_t)(S.nb * 4), S.nb, S.hdr + pair * ASX_SPEC_HDR, S.tol, S.N); // the sums
    // ... and the mode k_pearson_partial's block 0 recorded: the one its blocks wrote the partial sums for (never decided twice)
    const int mode = (int)S.hdr[pair * ASX_SPEC_HDR + 3];
    PStat v;
    v.n = v.mx = v.my = v.mxx = v.myy = v.cxy = 0.0;
    if (mode != ASX_PM_FAST) { // wave-uniform
        for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
            const double *p = psums + (pair * nb + b) * 6;
            PStat o;
            o.n = p[0];
            if (o.n != 0.0) { o.mx = p[1]; o.my = p[2]; o.mxx = p[3]; o.myy = p[4]; o.cxy = p[5]; v = pstat_merge(v, o); }
        }
        v = pstat_wave_merge(v);
    }
    if (threadIdx.x == 0) {
        double c;
        if (mode == ASX_PM_DIRECT) {
            c = v.cxy / sqrt(v.mxx * v.myy); // src/cross_correlation.c:115
        } else {
            const double n = d.n, Sx = d.Sx, Sxx = d.Sxx, Sy = d.Sy, Syy = d.Syy;
            double sxy = d.r;
            if (mode == ASX_PM_CORR) sxy -= v.cxy + v.n * v.mx * v.my; // minus the products that did not wrap around
            c = (sxy - Sx * Sy / n) / sqrt((Sxx - Sx * Sx / n) * (Syy - Sy * Sy / n));
            c = c > 1.0 ? 1.0 : c < -1.0 ? -1.0 : c; // the reference's value cannot leave [-1, 1]; NaN stays NaN
        }
        if (lag) lag[pair] = s.lag;
        coef[pair] = c;
        if (ret) ret[pair] = (s.flags & ASX_SEG_INEXACT) ? 1 : (c != c) ? -1 : 0; // as k_pearson_final (:276)
        atomicAdd(S.mode_count + mode, 1ull);
    }
}

void asx_launch_pearson_spectral_f32(const AsxDev &P, const float *src, const float *smp, const AsxPeakWs &W, const AsxSpecWs &S0,
                                     AsxSeg *seg, double *psums, int64_t *lag, double *coef, int32_t *ret, int npairs,
                                     hipStream_t s)
{
    AsxSpecWs S = S0;
    S.N = P.N;
    if ((size_t)P.band_rows * (size_t)P.M2 >= 16384) {
        S.nb = ASX_PREP_BLOCKS;
        hipLaunchKernelGGL((k_pearson_prep<ASX_PREP_THREADS, ASX_PREP_BLOCKS>), dim3(ASX_PREP_BLOCKS, npairs), dim3(ASX_PREP_THREADS), 0, s, P.self_dev, src, smp, W, S, seg);
    } else {
        S.nb = 1;
        hipLaunchKernelGGL((k_pearson_prep<256, 1>), dim3(1, npairs), dim3(256), 0, s, P.self_dev, src, smp, W, S, seg);
    }
    asx_launch_pearson_partial_spec_f32(src, smp, 2 * (size_t)P.N, P.N, P.N, seg, S, psums, npairs, s);
    hipLaunchKernelGGL(k_pearson_final_spec, dim3(npairs), dim3(64), 0, s, seg, psums, asx_pearson_blocks(P.N), S, lag, coef, ret);
}
