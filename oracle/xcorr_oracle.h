/* oracle/xcorr_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, double precision) of the reference hot path
 * src/cross_correlation.c, function by function.  It is the checker for the
 * HIP path; it is never the thing shipped or measured as product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED for L <= 2000 by the reference's own 12 known-answer
 * tests (tests/test_cross_correlation.c:21-113, tests/test_pearson_coefficient.c:20-58,
 * carried as data in tests/golden/reference_kat.json).  The reference itself
 * cannot be built in this image (FFTW3 absent, and writing a stand-in fftw3.h
 * is not allowed), so at production sizes the DFT is additionally cross-checked
 * against numpy.fft (pocketfft, float64) in tests/test_oracle.py.
 */
#ifndef XCORR_ORACLE_H
#define XCORR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/cross_correlation.c:52-67 */
size_t oracle_max_abs_index(const double *arr, size_t len);

/* src/cross_correlation.c:74-116 */
double oracle_pearson_coefficient(const double *source_start, const double *source_end,
                                  const double *sample_start, const double *sample_end);

/* src/cross_correlation.c:133-307.  Returns 0, or -1 (allocation failure or NaN
 * coefficient; in the NaN case *lag and *coefficient have been written). */
int oracle_cross_correlation(const double *source, const double *sample, size_t sample_len,
                             long *lag, double *coefficient);

/* Same, but also hands back the raw correlation r[0..2N) (may be NULL) and
 * the peak margin (largest key / second largest key; >1 means unambiguous). */
int oracle_cross_correlation_ex(const double *source, const double *sample, size_t sample_len,
                                long *lag, double *coefficient, double *results_out,
                                double *peak_margin);

/* float32 inputs widened exactly to double (the parity rule of SURVEY.md "three facts" #1). */
int oracle_cross_correlation_f32(const float *source, const float *sample, size_t sample_len,
                                 long *lag, double *coefficient);

/* bench.py's cpu_baseline leg (cpu_baseline.c): one call with the reference's own cost model -- four
 * allocations, two threads that each plan + execute a forward transform, plan + execute the inverse,
 * everything released again (src/cross_correlation.c:26-46,159-239,300-304).  The transform backend is
 * real FFTW3 when libfftw3.so.3 can be dlopen()ed on the node, else this directory's own DFT. */
int oracle_cross_correlation_faithful(const double *source, const double *sample, size_t sample_len,
                                      long *lag, double *coefficient);
const char *oracle_baseline_backend(void); /* "fftw3" or "port" */
/* bench.py's node-throughput leg: an independent per-core worker with the same arithmetic and backend -- ONE thread,
 * plans and buffers made once (oracle_worker_create) and kept between calls; oracle_worker_run returns what the
 * reference's call would (0, or -1 for a NaN coefficient; -2 on an internal failure). */
typedef struct oracle_worker oracle_worker;
oracle_worker *oracle_worker_create(size_t sample_len);
int oracle_worker_run(oracle_worker *w, const double *source, const double *sample, long *lag, double *coefficient);
void oracle_worker_destroy(oracle_worker *w);

/* Deterministic synthetic pair generator, integer-exact so the
 * HIP generator (csrc/synth.hip) reproduces it bit for bit:
 *   u(i)      = 24-bit uniform in [-1,1) from splitmix64(key_sig + i),  i in [0,3N)
 *   source[j] = u(N + j),                                   j in [0,2N)
 *   sample[n] = 0.5*u(N + n + lag) + 2^-noise_shift * z(n),  n in [0,N)
 *   z(n)      = (sum of four 22-bit uniforms) * 2^-22 - 2    (variance 1/3)
 *   lag       = hash % (2*span+1) - span,  span = 3N/4  (both signs occur)
 * noise_shift: 3 => +12 dB SNR, 1 => 0 dB, 0 => -6 dB, -1 => -12 dB. */
void oracle_synth_pair(uint64_t seed, uint64_t pair, size_t sample_len, int noise_shift,
                       float *source, float *sample, int64_t *true_lag);

#ifdef __cplusplus
}
#endif
#endif
