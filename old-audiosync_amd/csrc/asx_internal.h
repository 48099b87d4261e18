// csrc/asx_internal.h — shared between the host side of the HIP layer
// (asx_api.hip) and the gfx950 kernels (xcorr_kernels.hip).
//
// Notation (also used in DESIGN.md and tests/model_fourstep.py):
//   N   sample_len                      F   real transform length (even)
//   M   F/2, complex transform length   M = M1*M2
//   j = j1*M2 + j2  (time index of the packed complex sequence)
//   k = k1 + M1*k2  (frequency index)
//   every intermediate in HBM is row-major [M1][M2] per pair
//   rows of the spectra sit at the DIGIT-REVERSED position pos1[k1]
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASX_MAX_STAGES 12
#define ASX_TW_LOG 11                       // two-level twiddle: low table has 2^11 entries
#define ASX_TW_LO (1u << ASX_TW_LOG)
#define ASX_THREADS 256                     // block size of the streaming kernels
#define ASX_FFT_THREADS_MAX 512             // upper bound for the three transform kernels
#define ASX_COL_LOADS 10                    // tile loads a thread keeps in flight in the column kernels
#define ASX_ROW_STEPS 5                     // max ceil(M2 / blockDim) in k_rows (bins a thread owns in the load / combine / store phases)
#define ASX_PEARSON_BLOCKS_MAX 512          // upper limit of the partial-sum blocks per pair (asx_pearson_blocks: by the length alone)
#define ASX_DC_STATS_DOUBLES (4 + 2 * 128)  // second look: {mean, sum, shift, -} + the per-block partials of k_dc_partial
// Peak refinement (src/cross_correlation.c:52-67 is a float64 scan).  The transforms run in float32, so
// every float32 r[k] is off by at most B = ASX_BOUND_C * eps32 * log2(F) * |source|_2 * |sample|_2
// (a worst-case bound of the three transforms is ~3 eps log2 F; measured maximum 0.4 eps log2 F).
// Every lag whose float32 key is within 2B of the float32 maximum is re-evaluated EXACTLY (float64
// dot product of the inputs, compensated summation) and the reference's rule is applied to the exact
// values -- for any number of such lags up to the per-pair capacity below (pure tones at N = 1 440 000
// have ~5 000).  The norms come for free from k_fwd_cols (it reads every input sample anyway).
#define ASX_BOUND_C 4.0f
#define ASX_CAND_MAX 16384                  // upper limit of a plan's per-pair candidate capacity
#define ASX_CAND_MIN 2048
#define ASX_DOT_BLOCKS 128                  // blocks per pair that walk the pair's candidate list

// Radix schedule of one in-LDS transform of length n.
// DIF stage i works on sub-blocks of length ns[i] = n / (radix[0]*...*radix[i-1]).
struct AsxStages {
    int n;
    int nstages;
    int radix[ASX_MAX_STAGES];
    int ns[ASX_MAX_STAGES];                 // sub-block length at DIF stage i
    int q[ASX_MAX_STAGES];                  // ns / radix  (butterflies per sub-block, and the leg stride)
    int nbf[ASX_MAX_STAGES];                // n / radix   (butterflies per transform)
    int twmul[ASX_MAX_STAGES];              // n / ns      (step in the w_n table for this stage)
    float inv_q[ASX_MAX_STAGES];            // 1/q and 1/nbf for the exact float division helper
    float inv_nbf[ASX_MAX_STAGES];
};

// Everything a kernel needs to know about a plan.  Kernels get a POINTER to a device copy:
// passed by value, the dynamically indexed stage arrays made hipcc spill the whole struct to scratch.
struct AsxDev;
struct AsxDev {
    uint32_t N;            // sample_len
    uint32_t F, M;         // real / complex transform length
    int M1, M2;            // M = M1*M2
    int T, logT;           // tile width (columns per block) of the column kernels, power of two
    int ntiles;            // ceil(M2 / T)
    int threads_cols, threads_rows; // block sizes of the column / row kernels
    uint32_t src_valid;    // how many leading real samples of the (periodically extended) source are non-zero
    uint32_t src_period;   // 2N
    uint32_t nout;         // 2N: lags searched
    float bound_scale;     // 2 * ASX_BOUND_C * eps32 * log2(F) * F: bound2 = bound_scale * |source|_2 * |sample|_2 (r is scaled by F)
    AsxStages st1, st2;    // schedules for length M1 and M2
    const float2 *tw1;     // w_{M1}^q, q < M1
    const float2 *tw2;     // w_{M2}^q, q < M2
    const float2 *tw2s;    // the same in slot order: tw2s[pos2_of_k2[k2]] = w_{M2}^k2
    const float2 *tw_lo;   // w_F^q, q < 2^ASX_TW_LOG
    const float2 *tw_hi;   // w_F^(h * 2^ASX_TW_LOG)
    const int *k1_of_pos1; // row slot -> k1
    const int *pos1_of_k1; // k1 -> row slot
    const int *pos2_of_k2; // k2 -> slot inside a row after the forward row transform
    const int4 *row_tasks; // [M1/2+1] {slot of row k1, slot of row M1-k1, k1, M1-k1}: one load starts a k_rows block
    int band_rows, nbands; // spectral Pearson form: rows of the [2 M1][M2] sample matrix per band (what one lane group of k_fwd_cols_r
                           // loads: asx_rlayout_band_rows), bands of the source = 2 M1 / band_rows; 0 = not available
    int rlayout;           // 1: this plan runs the real-column kernels (rlayout.hip); 0: the packed-sample kernels (xcorr_kernels.hip)
    const int4 *col_pairs; // real-column kernels (rlayout.hip): [M1/2 + 1] {u, slot of u, slot of M1 - u, 0}; null = not available
    const float2 *col_tw;  // w_{2 M1}^u, same order
    const AsxDev *self_dev; // device copy of this struct (what the kernels read)
    unsigned long long *stamps; // diagnostic builds (-DASX_STAMPS) only: per-block phase clocks, 8 slots per block
    int stamp_kernel;      // which kernel records them: 0 k_rows, 1 k_fwd_cols, 2 k_inv_cols ($ASX_STAMPS = 1 | fwd | inv)
};

// Peak-search partial: order-preserving key in the high word, ~index in the low word,
// so that a plain unsigned max picks the largest key and, among equals, the smallest index
// (src/cross_correlation.c:60 uses a strict '>').
typedef unsigned long long asx_peak_t;

struct AsxSeg {           // per pair, produced by k_finalize
    long long lag;        // wrapped lag (src/cross_correlation.c:256-271)
    uint32_t src_off;     // first source frame of the compared segment
    uint32_t smp_off;     // first sample frame
    uint32_t len;         // segment length (N, or N-|lag|; may be 0)
    uint32_t peak;        // raw argmax index in [0, 2N)
    uint32_t flags;       // ASX_SEG_INEXACT: the pair's near-tie list overflowed, `peak` is the float32 argmax (to be looked at again)
};
#define ASX_SEG_INEXACT 1u

// kernel launchers (defined in xcorr_kernels.hip, called from asx_api.hip)
struct AsxCand {          // one near-maximum lag found by a column tile
    uint32_t idx;
    float key;
};

// per-group scratch of the peak search
struct AsxPeakWs {
    float *nrm_part;       // [pairs][2][ntiles] sum of squares of the samples a k_fwd_cols block loaded
    float *bound2;         // [pairs] 2B: width of the "as large as the maximum" window (k_rows, row 0)
    asx_peak_t *pairmax;   // [pairs] running float32 maximum (atomicMax by the column tiles), zeroed by k_rows
    uint32_t *cand_n;      // [pairs] candidates appended by the tiles (may exceed cap), zeroed by k_rows
    AsxCand *cand;         // [pairs][cap]
    uint32_t *refine_n;    // [pairs] lags to re-evaluate (0 = the float32 argmax stands)
    uint32_t *refine_idx;  // [pairs][cap]
    double *refine_val;    // [pairs][cap] exact r[idx]
    unsigned long long *overflows; // [1] pairs whose candidate list did not fit, cumulative
    uint32_t *over_list;   // [over_cap] or null: indices (pair_base + pair) of those pairs since the list was last emptied --
    uint32_t *over_n;      // [1] ... and their number: what the entry points read to take the second look (asx_api.hip)
    uint32_t *over_host;   // [1] or null: the same count in page-locked HOST memory (system-scope add, only when a pair overflows):
                           // the entry points read it behind a stream synchronisation, with no device-to-host copy in the way
    uint32_t over_cap;
    const double *shift;   // [pairs] or null: c with r[k] = (what the transforms deliver) + c for every k -- the second look at a pair
                           // runs the transforms on (source - mean), see second_look (asx_api.hip); null / 0 everywhere else
    uint32_t cap;          // candidate capacity per pair
    // spectral Pearson (pearson_spectral.hip), both null when the plan does not use it:
    float2 *band;          // [pairs][2][ntiles][nbands] {sum, sum of squares} of the samples of band x tile a k_fwd_cols_r block loaded
                           // (band = AsxDev::band_rows consecutive rows of the [2 M1][M2] sample matrix; the sample fills the first half)
    float *tile_peak;      // [pairs][M2 / T] SIGNED float32 r (times F) at the best lag of each k_inv_cols_r tile
};

// Spectral Pearson: the coefficient from r[peak] and window sums instead of a second pass over the inputs.
// Modes a pair can take (k_pearson_prep decides, k_pearson_partial / k_pearson_final_spec act on it):
#define ASX_PM_FAST 0    // lag >= 0: cross term = r[peak], window sums from the band sums + two band edges; nothing else is read
#define ASX_PM_CORR 1    // lag < 0:  cross term = r[peak] - (wrap-around part, |lag| products, exact), sums as above
#define ASX_PM_DIRECT 2  // the reference's own reduction over the segment (src/cross_correlation.c:74-116): the error bound of the
                         // spectral form is not below the tolerance, or the segment is shorter than the wrap-around part
#define ASX_PM_NMODES 3
#ifndef ASX_PREP_BLOCKS
#define ASX_PREP_BLOCKS 4 // blocks of k_pearson_prep that share a pair's window sums on the long tracks (a function of the plan alone)
#endif
#define ASX_PREP_BLOCKS_MAX 16 // what the workspaces are sized for
#define ASX_SPEC_HDR 4    // per pair: r[peak] (plain sum scale), the bound on its error, 1.0 = "the direct reduction, whatever it yields", the mode k_pearson_partial's block 0 used (read by k_pearson_final_spec)
// What k_pearson_prep leaves and what reads it: the blocks' SHARES of the four window sums and a header -- no merged record.  The mode of a
// pair (asx_spec_pick, xcorr_dev.h: a few dozen float64 operations on 4 * nb + 3 numbers) is worked out again by every block of
// k_pearson_partial and by k_pearson_final_spec: a merge by the last block to arrive cost a device-scope fence per block (28.8 against
// 19.5 us per launch of 124 pairs, profiles/r5_experiments/22_*).
struct AsxSpecWs {
    double *part;          // [pairs][nb][4] the blocks' shares of Sx, Sxx, Sy, Syy, added in block order by whoever reads them
    double *hdr;           // [pairs][ASX_SPEC_HDR]
    unsigned long long *mode_count; // [ASX_PM_NMODES] cumulative
    double tol;            // a pair leaves the spectral form when its error bound exceeds this (1e-5: north_star's tolerance)
    int nb;                // blocks of k_pearson_prep per pair (set by the launcher: 1 or ASX_PREP_BLOCKS)
    uint32_t N;            // sample_len (set by the launcher)
};

void asx_launch_fwd_cols(const AsxDev &P, const float *src, const float *smp, float2 *zxa,
                         float2 *zya, const AsxPeakWs &W, int npairs, hipStream_t s);
void asx_launch_rows(const AsxDev &P, const float2 *zxa, const float2 *zya, float2 *ga,
                     const AsxPeakWs &W, int npairs, hipStream_t s);
void asx_launch_inv_cols(const AsxDev &P, const float2 *ga, const AsxPeakWs &W, float *r_out,
                         int npairs, hipStream_t s);
// rlayout.hip: the real-column decomposition (production lengths); false = no kernel compiled in for this plan
bool asx_launch_rows_r(const AsxDev &P, const float2 *cx, const float2 *cy, float2 *q, const AsxPeakWs &W, int npairs,
                       hipStream_t s);
bool asx_launch_fwd_cols_r(const AsxDev &P, const float *src, const float *smp, float2 *cx, float2 *cy, const AsxPeakWs &W,
                           int npairs, hipStream_t s);
bool asx_launch_inv_cols_r(const AsxDev &P, const float2 *q, const AsxPeakWs &W, float *r_out, int npairs, hipStream_t s);
bool asx_rlayout_available(const AsxDev &P); // all three kernels compiled in for this plan's schedules
int asx_rlayout_band_rows(const AsxDev &P);
void asx_launch_finalize(const AsxDev &P, const AsxPeakWs &W, AsxSeg *seg, int npairs, hipStream_t s, uint32_t pair_base = 0);
void asx_launch_refine_f32(const AsxDev &P, const float *src, const float *smp, const AsxPeakWs &W,
                           AsxSeg *seg, int npairs, hipStream_t s, int dot_blocks = ASX_DOT_BLOCKS, bool pick = true);
                           // pick = false: the exact values only; the caller's next kernel applies the rule (k_pearson_prep)
void asx_launch_refine_f64(const AsxDev &P, const double *src, const double *smp, const AsxPeakWs &W,
                           AsxSeg *seg, int npairs, hipStream_t s, int dot_blocks = ASX_DOT_BLOCKS);
void asx_launch_pearson_f32(const float *src, const float *smp, size_t src_pitch, size_t smp_pitch,
                            uint32_t basis_len, const AsxSeg *seg, double *psums, int64_t *lag,
                            double *coef, int32_t *ret, int npairs, hipStream_t s);
void asx_launch_pearson_f64(const double *src, const double *smp, size_t src_pitch, size_t smp_pitch,
                            uint32_t basis_len, const AsxSeg *seg, double *psums, int64_t *lag,
                            double *coef, int32_t *ret, int npairs, hipStream_t s);
// the partial-sum kernel alone (the spectral form runs it on its own segment list, pearson_spectral.hip)
void asx_launch_pearson_partial_spec_f32(const float *src, const float *smp, size_t src_pitch, size_t smp_pitch, uint32_t basis_len,
                                         const AsxSeg *seg, const AsxSpecWs &S, double *psums, int npairs, hipStream_t s);
// pearson_spectral.hip: float32 inputs, real-column plans (W.band and W.tile_peak filled by this group's transform kernels)
void asx_launch_pearson_spectral_f32(const AsxDev &P, const float *src, const float *smp, const AsxPeakWs &W, const AsxSpecWs &S,
                                     AsxSeg *seg, double *psums, int64_t *lag, double *coef, int32_t *ret, int npairs,
                                     hipStream_t s);
void asx_launch_results_to_ms(const int64_t *lag, const double *coef, const int32_t *ret, size_t batch,
                              double min_confidence, double sample_rate, int64_t *lag_ms, int32_t *accept,
                              hipStream_t s);
void asx_launch_cvt_f64_f32(const double *in, float *out, size_t n, hipStream_t s);
unsigned asx_pearson_blocks(uint32_t basis_len); // partial blocks per pair: psums holds 6 doubles per block and pair
// second look, DC removal: stats[0] = mean of source[0..2N), stats[1] = sum of sample[0..N), stats[2] = scale * stats[0] * stats[1]
// (scale = F: the device's r is F times the plain sum of products); out[i] = (float)(source[i] - stats[0]); stats holds
// ASX_DC_STATS_DOUBLES doubles
void asx_launch_dc_remove_f32(const float *src, const float *smp, uint32_t N, double scale, double *stats, float *out, hipStream_t s);
void asx_launch_dc_remove_f64(const double *src, const double *smp, uint32_t N, double scale, double *stats, float *out, hipStream_t s);
void asx_launch_synth(uint64_t seed, uint64_t first_pair, size_t count, uint32_t N,
                      int noise_shift, float *src, float *smp, int64_t *true_lag, hipStream_t s);
int asx_pick_threads(const AsxStages &st, int groups, int min_threads, size_t lds_bytes);
size_t asx_lds_bytes_cols(const AsxDev &P);
size_t asx_lds_bytes_rows(const AsxDev &P);
