/* include/audiosync/xcorr_hip.h — the C-ABI of the MI355X (gfx950) layer.
 *
 * This is the drop-in boundary for ONE path of vidify/old-audiosync: the FFT
 * cross-correlation of src/cross_correlation.c.  Plain C, plain pointers and
 * sizes; no C++/torch types.  The library behind it (libaudiosync_hip.so) is
 * hand-written HIP for gfx950 and has no CPU fallback: every entry point
 * fails (-1 / NULL, message in asx_last_error()) when no HIP device works.
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *
 *   asx_xcorr_f64            the body of cross_correlation()
 *                            src/cross_correlation.c:133-307, called from
 *                            src/audiosync.c:246 and tests/test_cross_correlation.c:25..109.
 *                            include/audiosync/cross_correlation.h:24-25 is the signature
 *                            the host-side wrapper (host/cross_correlation.c) keeps.
 *   asx_pearson_f64          pearson_coefficient(), src/cross_correlation.c:74-116,
 *                            include/audiosync/cross_correlation.h:10-11
 *   asx_plan_create/destroy  the per-call fftw_plan_dft_r2c_1d / _c2r_1d / fftw_alloc_*
 *                            / fftw_free of src/cross_correlation.c:33-36,159,187-201,
 *                            237-239,300-304, hoisted into a reusable object
 *   asx_xcorr_batch_f32      many independent cross_correlation() calls on float32 data
 *   asx_xcorr_batch_f32_dev  (BASELINE.json north_star: batched many-pair variant);
 *                            no reference equivalent beyond a loop over :133-307
 *   asx_synth_pairs_dev      synthetic 48 kHz mono float32 pairs for benchmarks
 *                            (stands in for the producers src/ffmpeg_pipe.c:68-81)
 *
 * Return convention everywhere: 0 = ok, -1 = error (asx_last_error() says why).
 * Per-pair result convention (the reference's, src/cross_correlation.c:140,276,298):
 *   ret[i] = 0 on success, -1 when the Pearson coefficient is NaN (lag[i] and
 *   coef[i] are still written, exactly as the reference leaves them).
 */
#ifndef AUDIOSYNC_XCORR_HIP_H
#define AUDIOSYNC_XCORR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct asx_plan asx_plan;

/* ---- library / device ------------------------------------------------- */

/* Number of usable HIP devices (0 when there is none or the runtime fails). */
int asx_device_count(void);
/* The calling thread's current HIP device (-1 when the runtime fails): what device < 0 means below. */
int asx_current_device(void);
/* Last error message of the calling thread ("" if none). Never NULL. */
const char *asx_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int asx_abi_version(void);

/* ---- plans ------------------------------------------------------------ */

/* A plan fixes sample_len (N frames; source is 2N frames) and owns twiddle
 * tables, index tables and HBM workspaces for up to `max_batch` pairs per
 * launch group (larger batches are processed in groups).  device < 0 means
 * "the current HIP device".  Returns NULL on error.
 * Supported N: any N >= 1 whose transform length splits into two factors that
 * fit the LDS kernels (N up to about 4,000,000).  Lengths 2N that are not
 * {2,3,5}-smooth are embedded in a longer smooth transform (same r[k]). */
asx_plan *asx_plan_create(size_t sample_len, size_t max_batch, int device);
/* Same, with the transform split chosen by the caller: "M1xM2xT" (M1*M2 = F/2, T = tile
 * columns, a power of two <= 64) forces it; "measure" times the planner's best candidates
 * on the device (synthetic pairs, a fraction of a second) and keeps the fastest -- what
 * FFTW_MEASURE is to the FFTW_ESTIMATE of src/cross_correlation.c:187-201; NULL, "" or
 * "auto" = the planner's cost model (tuned table for the reference's six lengths).
 * $ASX_SPLIT supplies the value when the argument is NULL or "". */
asx_plan *asx_plan_create_ex(size_t sample_len, size_t max_batch, int device, const char *split);
/* "measure" plans only, once, at their first asx_xcorr_batch_f32_dev call of at least min(group, 8) pairs: the forward column
 * kernel is timed against the caller's buffers on two allocations of its output workspaces (those of the plan's first stream
 * lane) and the faster set is kept (the kernel runs 2-5 % apart with the physical placement of the buffers it streams
 * together; offsets inside an allocation change nothing).  That call allocates, SYNCHRONISES with the device (events on the
 * caller's stream) and frees: a few milliseconds and, transiently, a second set of workspaces; a call made while its stream is
 * being captured into a graph does not tune (the next one outside a capture does).  asx_plan_placement: the two times in ms (0 = not measured)
 * and which set was kept (0 = the first, 1 = the second, -1 = not measured).  Plans of every other mode never do this. */
int asx_plan_placement(asx_plan *plan, double ms[2], int *kept);
void asx_plan_destroy(asx_plan *plan);

/* Peak search exactness (replaces max_abs_index(), src/cross_correlation.c:52-67, a float64 scan).
 * The transforms run in float32; every lag whose float32 value is within the float32 error bound of
 * the float32 maximum is re-evaluated exactly and the reference's rule is applied to the exact values,
 * for up to asx_plan_peak_capacity() such lags per pair (2048..16384 by sample_len; all 2N lags for
 * short tracks).  A pair with more near-ties than that (a signal periodic in that many lags, an offset
 * of hundreds of deviations in both tracks) is marked on the device, counted -- asx_plan_peak_overflows():
 * *count = number of such pairs since the plan was created -- and EVERY entry point, the device-resident
 * asx_xcorr_batch_f32_dev included, takes a second look at it before it returns: the pair's transforms
 * again on the mean-removed source, into lists that hold all 2N lags, so that no candidate limit remains
 * and the lag is the reference's float64 argmax by construction (asx_plan_peak_repairs() counts these;
 * slow: candidates x N multiply-adds).  The price: one host synchronisation of the call's stream per call
 * (per window of >= 1024 pairs of a very long batch), behind the last launch group; the second look itself
 * only runs when the list of marked pairs is not empty.
 * asx_plan_set_exact(plan, 0) trades that for a fully asynchronous asx_xcorr_batch_f32_dev: no host
 * synchronisation, and a marked pair comes back with ret[i] = 1 ("inexact: lag[i] is the float32 argmax,
 * submit the pair again on an exact plan") -- never silently.  ret[i] = 1 cannot occur in the default mode.
 * Both getters synchronise the plan's streams. */
int asx_plan_peak_overflows(asx_plan *plan, uint64_t *count);
int asx_plan_peak_repairs(asx_plan *plan, uint64_t *count);
/* on != 0 (the default): see above.  Synchronises the plan's streams. */
int asx_plan_set_exact(asx_plan *plan, int on);
size_t asx_plan_peak_capacity(const asx_plan *plan);

/* The Pearson coefficient of src/cross_correlation.c:74-116 (pearson_coefficient(), :272) in the batched float32 entry
 * points.  Two forms, the same formula:
 *   direct    the reference's reduction over both segments (one streaming pass with the accuracy of its two);
 *   spectral  (the default on plans for the reference's six lengths, asx_plan_layout() == 1) the cross term is r[peak] --
 *             exactly the sum the transforms have just computed (minus, for a negative lag, the |lag| products that did not
 *             wrap around) -- and the four window sums come from per-band sums the forward pass keeps of the samples it
 *             loads anyway; the inputs are not read a second time.  r[peak] carries the float32 transforms' error; its
 *             bound (the one that guards the lag) and that of the float32 band sums are turned into a bound on the
 *             coefficient's error PER PAIR, and a pair whose bound exceeds 1e-5 (quiet windows of a loud track, large
 *             offsets, short segments) takes the direct form by itself.  So: |coefficient - reference's| <= 1e-5 either
 *             way (the bound is first order in the rounding errors it adds up; measured worst case over the parity and
 *             fuzz runs: 5e-7); the spectral form's value is not bit-identical to the direct form's.
 *             ONE OBSERVABLE DIFFERENCE: identical (or exactly negated) segments give exactly +-1.0 only in the direct form
 *             (the reference's own test asserts `coefficient == 1.0`, tests/test_cross_correlation.c:29); the spectral form
 *             returns a value clamped to [-1, 1] within 1e-5 of +-1.0 (tests/test_gpu_pearson_spectral.py::
 *             test_identical_segments_on_the_spectral_path).  asx_plan_set_pearson(plan, 0) restores the reference's bits.
 * cross_correlation(double*) / asx_xcorr_f64 / asx_stream_xcorr always use the direct form on the caller's doubles.
 * asx_plan_pearson_modes: pairs so far that took {spectral, spectral + wrap-around correction, direct} under the spectral
 * setting (synchronises the DEVICE: batches submitted on a caller's stream are counted too). */
int asx_plan_set_pearson(asx_plan *plan, int spectral);
int asx_plan_pearson_modes(asx_plan *plan, uint64_t counts[3]);

/* Introspection (used by tests, bench and DESIGN.md's numbers). */
size_t asx_plan_sample_len(const asx_plan *plan);
size_t asx_plan_fft_len(const asx_plan *plan);        /* F, real transform length */
int asx_plan_split(const asx_plan *plan, int *m1, int *m2, int *tile_cols);
/* Which decomposition the plan runs: 1 = the real-column kernels (csrc/rlayout.hip: r2c / c2r column transforms with the
 * untangling inside the tile, one independent row problem per k1; the reference's six lengths), 0 = the packed-sample
 * kernels (csrc/xcorr_kernels.hip: any length; forced by $ASX_LAYOUT=packed), -1 = NULL plan.  Same results either way. */
int asx_plan_layout(const asx_plan *plan);
int asx_plan_threads(const asx_plan *plan, int *threads_cols, int *threads_rows); /* block sizes */
size_t asx_plan_group(const asx_plan *plan);          /* pairs per launch group */
size_t asx_plan_workspace_bytes(const asx_plan *plan);

/* ---- the hot path ------------------------------------------------------ */

/* One pair, host double buffers: the reference's own calling convention.
 * source: 2N doubles (read only), sample: N doubles.  Copies to the device,
 * converts to float32 there, runs the transforms in float32 and the Pearson
 * reduction in float64 on the ORIGINAL doubles.  Returns 0, or -1 exactly
 * where the reference does (device/allocation failure: outputs untouched;
 * NaN coefficient: outputs written). */
int asx_xcorr_f64(asx_plan *plan, const double *source, const double *sample, long *lag,
                  double *coefficient);
/* When EVERY double of both buffers is exactly a float32 -- true of whatever ffmpeg decodes from 16-bit or float audio,
 * although the reference asks it for f64le (src/capture/linux_capture.c:370) -- asx_xcorr_f64 moves 4 bytes per frame across PCIe instead of 8: the check and the conversion run on a small host thread pool
 * ($ASX_HOST_THREADS, default 12 or the cgroup's CPU quota) into page-locked staging, overlapped with the uploads, and the float64 passes read the
 * float32 copy widened on the device: the same values, the same operations, the same bits.  Anything else (a NaN, a value
 * with more than 24 significant bits) takes the 8-byte route.  $ASX_NARROW=0 switches the check off.
 * asx_plan_narrowed_calls(): how many asx_xcorr_f64 calls on this plan went the 4-byte way. */
int asx_plan_narrowed_calls(asx_plan *plan, uint64_t *count);

/* `batch` pairs, host float32 buffers laid out pair after pair:
 * source[batch][2N], sample[batch][N].  lag/coef/ret: `batch` entries each. */
int asx_xcorr_batch_f32(asx_plan *plan, const float *source, const float *sample, size_t batch,
                        int64_t *lag, double *coef, int32_t *ret);

/* Same with everything already resident in this plan's device memory space.
 * All pointers are DEVICE pointers; `stream` is a hipStream_t (NULL = the
 * plan's own stream).  Results are valid after the stream is synchronised (the
 * call itself waits once for its kernels to look at the list of overflowed pairs,
 * see asx_plan_set_exact; what it enqueues after that is asynchronous).
 * source pairs are 2N floats apart, sample pairs N floats.
 * A plan's workspaces are shared by all its calls: use ONE stream at a time per plan
 * (calls on different streams must be ordered by the caller, e.g. with events);
 * concurrent work belongs on separate plans. */
int asx_xcorr_batch_f32_dev(asx_plan *plan, const float *d_source, const float *d_sample,
                            size_t batch, int64_t *d_lag, double *d_coef, int32_t *d_ret,
                            void *stream);

/* The batched variant over several GPUs of one node from ONE process (BASELINE.json north_star; no
 * reference equivalent): plans[i] was created on device i (any devices; all the same sample_len); the
 * batch is block-partitioned over the plans, each block runs concurrently on its device, results come
 * back in pair order.  Pairs are independent: nothing is exchanged between devices.  (bench.py uses
 * one process per GPU and RCCL for the result gather instead; this is the entry point for C hosts.) */
int asx_xcorr_batch_multi(asx_plan *const *plans, int nplans, const float *source, const float *sample,
                          size_t batch, int64_t *lag, double *coef, int32_t *ret);

/* The same with DEVICE-RESIDENT shards and the result gather done by RCCL inside the library (one
 * host thread and one stream per device in a single process, ncclCommInitAll, one ncclAllGather per batch):
 *
 *   asx_shard_range(total, nshards, i, &start, &count)   block partition of a batch: the first total % nshards
 *                       shards hold one pair more (the rule bench.py's sharding.py uses between processes)
 *   asx_comm_create(plans, nplans)   one RCCL communicator over the plans' devices (all different, one plan each,
 *                       all the same sample_len); librccl.so.1 is dlopen()ed here, not linked: hosts that never
 *                       shard do not load it.  NULL on failure (asx_last_error()).
 *   asx_xcorr_batch_multi_dev(comm, d_source, d_sample, counts, width, d_gathered)
 *                       shard i = counts[i] <= width pairs resident on plans[i]'s device (d_source[i]: counts[i] * 2N
 *                       floats, d_sample[i]: counts[i] * N).  Every device runs its shard on its plan's stream from
 *                       its own host thread, then the shards' result records -- asx_result_bytes(width) bytes each
 *                       (20 * width rounded up to a multiple of 8, so that every record starts 8-byte aligned):
 *                       int64 lag[width] | double coef[width] | int32 ret[width], entries past counts[i] zero --
 *                       are all-gathered over xGMI: d_gathered[i] (device i, nplans * asx_result_bytes(width)
 *                       bytes) receives the record of every shard, in shard order.  Returns after all streams
 *                       have been synchronised.  The data path itself exchanges nothing: pairs are independent.
 *   asx_comm_destroy(comm) */
typedef struct asx_comm asx_comm;
int asx_shard_range(size_t total, int nshards, int shard, size_t *start, size_t *count);
size_t asx_result_bytes(size_t width);
asx_comm *asx_comm_create(asx_plan *const *plans, int nplans);
void asx_comm_destroy(asx_comm *comm);
int asx_xcorr_batch_multi_dev(asx_comm *comm, const float *const *d_source, const float *const *d_sample,
                              const size_t *counts, size_t width, void *const *d_gathered);

/* Debug/parity aid: run ONE device-resident pair and also return the raw
 * correlation r[0..2N) (device pointer, 2N floats; scaled by F/(2N) relative
 * to the reference when the length had to be embedded).  Like every entry point it
 * takes the second look at a pair whose near-tie list overflowed (exact mode, the
 * default) or marks it with ret = 1 (asx_plan_set_exact(plan, 0)), and leaves nothing
 * on the plan's overflow list either way; the coefficient is the direct form's. */
int asx_xcorr_debug_r_dev(asx_plan *plan, const float *d_source, const float *d_sample,
                          float *d_r, int64_t *d_lag, double *d_coef, int32_t *d_ret,
                          void *stream);

/* pearson_coefficient() on two equal-length host double ranges. Writes the
 * coefficient (NaN for a constant range, like the reference). */
int asx_pearson_f64(const double *source_seg, const double *sample_seg, size_t n, int device,
                    double *coefficient);

/* ---- result consumers ---------------------------------------------------- */

/* What audiosync_run does with a result (src/audiosync.c:254-256,
 * include/audiosync/audiosync.h:21,24), for a whole batch on the device:
 *   accept[i]  = ret[i] == 0 && coef[i] >= min_confidence      (1 / 0)
 *   lag_ms[i]  = round(lag[i] * 1000 / sample_rate)            (C round(): halves away from zero)
 * All pointers are device pointers; d_accept may be NULL. Asynchronous on `stream`. */
int asx_results_to_ms_dev(const int64_t *d_lag, const double *d_coef, const int32_t *d_ret,
                          size_t batch, double min_confidence, double sample_rate,
                          int64_t *d_lag_ms, int32_t *d_accept, void *stream);

/* ---- growing-window (streaming) mode ------------------------------------ */

/* The reference re-runs the whole correlation on growing prefixes of the two
 * tracks (3, 6, 10, 15, 20, 30 s; src/audiosync.c:50-57,226-259) and rebuilds
 * plans and buffers every time.  A stream keeps both tracks resident in HBM:
 * only NEW frames are uploaded (as the producers' f64le doubles,
 * src/capture/linux_capture.c:370, and converted to float32 on the device),
 * and one plan per prefix length is built once and reused. */
typedef struct asx_stream asx_stream;

/* Capacity: sample track max_sample_len frames, source track 2*max_sample_len. */
asx_stream *asx_stream_create(size_t max_sample_len, int device);
void asx_stream_destroy(asx_stream *stream);
/* Append frames to the tracks (either count may be 0). -1 if capacity would be exceeded. */
int asx_stream_append_f64(asx_stream *stream, const double *source_frames, size_t n_source,
                          const double *sample_frames, size_t n_sample);
/* Frames appended so far. */
int asx_stream_lengths(const asx_stream *stream, size_t *n_source, size_t *n_sample);
/* Forget all frames (plans stay). */
int asx_stream_reset(asx_stream *stream);
/* cross_correlation() on the prefixes source[0,2*sample_len), sample[0,sample_len) that are
 * already resident.  Same return convention as asx_xcorr_f64.  -1 (outputs untouched) if
 * fewer frames than that have been appended. */
int asx_stream_xcorr(asx_stream *stream, size_t sample_len, long *lag, double *coefficient);

/* ---- synthetic inputs and timing -------------------------------------- */

/* Fill device buffers with pairs [first_pair, first_pair+count) of the
 * deterministic generator specified in oracle/xcorr_oracle.h (bit-identical
 * to oracle_synth_pair).  d_true_lag may be NULL. */
int asx_synth_pairs_dev(uint64_t seed, uint64_t first_pair, size_t count, size_t sample_len,
                        int noise_shift, float *d_source, float *d_sample, int64_t *d_true_lag,
                        void *stream);

/* Milliseconds spent by recent asx_xcorr_batch_f32_dev calls on `plan`, per kernel family, measured
 * with HIP events on the stream the kernels ran on.  asx_plan_set_profiling(plan, depth): depth > 0
 * keeps the events of the last `depth` calls (so that consecutive steps can be timed with no host
 * synchronisation between them), 0 switches profiling off.  asx_plan_timings_ms(plan, calls_back, out):
 * the call `calls_back` calls ago (0 = the latest); asx_plan_last_timings_ms = calls_back 0.
 * out[0..5] = fwd_cols, rows, inv_cols, finalize (+ exact re-evaluation), pearson, total. */
int asx_plan_set_profiling(asx_plan *plan, int depth);
int asx_plan_timings_ms(asx_plan *plan, int calls_back, float out[6]);
int asx_plan_last_timings_ms(asx_plan *plan, float out[6]);

/* Raw device memory helpers so a C host (no torch) can stage buffers. */
void *asx_device_malloc(size_t bytes, int device);
int asx_device_free(void *ptr);
/* Page-locked host memory: frames appended from it (asx_stream_append_f64) or passed to the host-array entry points travel
 * by DMA without the runtime's bounce buffer.  The track buffers of audiosync_run() are allocated with it -- what
 * fftw_alloc_real was to the reference's source buffer (src/audiosync.c:189): the allocation its backend wants. */
void *asx_host_malloc(size_t bytes);
int asx_host_free(void *ptr);
int asx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int asx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int asx_stream_sync(asx_plan *plan, void *stream);

#ifdef __cplusplus
}
#endif
#endif
