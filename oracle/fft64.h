/* oracle/fft64.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Double-precision DFT of arbitrary length, written from the published
 * definition of the transforms the reference obtains from FFTW3
 * (third-party, not vendored under /root/reference, version unpinned:
 * cmake/FindFFTW.cmake:14, setup.py:12):
 *
 *   r2c  (fftw_plan_dft_r2c_1d, src/cross_correlation.c:34)
 *        X[k] = sum_{j<L} x[j] * exp(-2*pi*i*j*k/L),  k = 0 .. L/2, unnormalised
 *   c2r  (fftw_plan_dft_c2r_1d, src/cross_correlation.c:237)
 *        r[j] = sum_{k<L} X~[k] * exp(+2*pi*i*j*k/L), X~ the Hermitian
 *        extension of X[0..L/2]; unnormalised; Im X[0] (and Im X[L/2] for
 *        even L) ignored.
 *
 * Nothing here is shipped or measured as product; only tests/, the smoke
 * check and bench.py's cpu_baseline leg may link it.
 */
#ifndef ORACLE_FFT64_H
#define ORACLE_FFT64_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } ocpx;

typedef struct offt_plan offt_plan;

/* complex DFT plan of length n (any n >= 1). sign = -1 forward, +1 inverse. */
offt_plan *offt_plan_create(size_t n);
void offt_plan_destroy(offt_plan *p);
/* out-of-place; in and out must not alias; `in` is preserved. */
void offt_execute(const offt_plan *p, const ocpx *in, ocpx *out, int sign);
/* the same with the caller's scratch of n elements; -1 for lengths that need Bluestein (use offt_execute) */
int offt_execute_ws(const offt_plan *p, const ocpx *in, ocpx *out, int sign, ocpx *scratch);

/* real transforms with FFTW r2c / c2r conventions (see header comment).
 * L >= 1.  rfft writes L/2+1 bins.  irfft reads L/2+1 bins, writes L reals. */
int offt_rfft(size_t L, const double *x, ocpx *X);
int offt_irfft(size_t L, const ocpx *X, double *r);

#ifdef __cplusplus
}
#endif
#endif
