/* include/audiosync/cross_correlation.h — MI355X build.
 *
 * The two exported functions of the hot path, with the reference's exact
 * signatures (reference: include/audiosync/cross_correlation.h:10-11,24-25) so
 * that existing callers (src/audiosync.c:246, tests/test_cross_correlation.c,
 * tests/test_pearson_coefficient.c) compile and link unchanged.  Both are
 * implemented in old-audiosync_amd/host/cross_correlation.c on top of the
 * gfx950 layer declared in audiosync/xcorr_hip.h; neither has a CPU fallback.
 */
#ifndef AUDIOSYNC_CROSS_CORRELATION_H
#define AUDIOSYNC_CROSS_CORRELATION_H

#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Lag of `data2` (length frames, zero-padded internally) inside `data1`
 * (2*length frames, used as is): peak of the circular correlation
 * ifft(fft(data1) * conj(fft(data2))), wrapped to [-length, length), plus the
 * Pearson coefficient of the overlapping segments as a confidence in [-1, 1].
 * Returns 0, or -1 on failure / NaN coefficient (then the outputs still hold
 * the wrapped lag and the NaN).  Inputs are rounded to float32 for the
 * transforms; the coefficient is accumulated in float64 from the doubles. */
int cross_correlation(double *data1, double *data2, const size_t length,
                      long *displacement, double *coefficient);

/* Sample Pearson correlation of [source_start, source_end) against the
 * equally long range starting at sample_start.  NaN when either range is
 * constant.  The ranges must not be empty. */
double pearson_coefficient(double *source_start, const double *source_end,
                           double *sample_start, const double *sample_end);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOSYNC_CROSS_CORRELATION_H */
