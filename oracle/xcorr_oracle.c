/* oracle/xcorr_oracle.c — TEST INFRASTRUCTURE ONLY (see xcorr_oracle.h). */
#include "xcorr_oracle.h"
#include "fft64.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/cross_correlation.c:52-67 — the running maximum starts at the SIGNED
 * value of element 0 (:56); later elements compete with fabs (:59); strict '>'
 * (:60) keeps the earliest index among equals; a NaN never compares greater. */
size_t oracle_max_abs_index(const double *arr, size_t len)
{
    double best = arr[0];
    size_t where = 0;
    for (size_t i = 1; i < len; i++) {
        double a = fabs(arr[i]);
        if (a > best) { best = a; where = i; }
    }
    return where;
}

/* src/cross_correlation.c:74-116 — two passes: the means (:82-95), then the
 * three centred sums and diffprod / sqrt(d1sq * d2sq) (:98-115). */
double oracle_pearson_coefficient(const double *source_start, const double *source_end,
                                  const double *sample_start, const double *sample_end)
{
    const ptrdiff_t n1 = source_end - source_start;
    const ptrdiff_t n2 = sample_end - sample_start;
    double s1 = 0.0, s2 = 0.0;
    for (ptrdiff_t i = 0; i < n1; i++) { s1 += source_start[i]; s2 += sample_start[i]; }
    const double m1 = s1 / (double)n1;
    const double m2 = s2 / (double)n2;
    double pr = 0.0, q1 = 0.0, q2 = 0.0;
    for (ptrdiff_t i = 0; i < n1; i++) {
        const double d1 = source_start[i] - m1;
        const double d2 = sample_start[i] - m2;
        pr += d1 * d2;
        q1 += d1 * d1;
        q2 += d2 * d2;
    }
    return pr / sqrt(q1 * q2);
}

int oracle_cross_correlation_ex(const double *source, const double *input_sample, size_t N,
                                long *lag, double *coefficient, double *results_out,
                                double *peak_margin)
{
    int ret = -1;
    const size_t L = 2 * N;      /* :141 */
    const size_t H = L / 2 + 1;  /* :142 */
    double *sample = malloc(sizeof(double) * L);
    double *results = malloc(sizeof(double) * L);
    ocpx *X = malloc(sizeof(ocpx) * H);
    ocpx *Y = malloc(sizeof(ocpx) * H);
    if (!sample || !results || !X || !Y) goto finish;

    /* :164-166 zero-padded copy of the sample; the source is used as is */
    memcpy(sample, input_sample, sizeof(double) * N);
    memset(sample + N, 0, sizeof(double) * (L - N));

    /* :204-229 two forward r2c transforms of length L */
    if (offt_rfft(L, source, X) != 0) goto finish;
    if (offt_rfft(L, sample, Y) != 0) goto finish;

    /* :232-233  arr1[i] *= conj(arr2[i]) */
    for (size_t k = 0; k < H; k++) {
        const double re = X[k].re * Y[k].re + X[k].im * Y[k].im;
        const double im = X[k].im * Y[k].re - X[k].re * Y[k].im;
        X[k].re = re; X[k].im = im;
    }

    /* :237-239 unnormalised c2r */
    if (offt_irfft(L, X, results) != 0) goto finish;
    if (results_out) memcpy(results_out, results, sizeof(double) * L);

    /* :242 */
    size_t peak = oracle_max_abs_index(results, L);
    if (peak_margin) {
        double best = peak == 0 ? results[0] : fabs(results[peak]);
        double second = -INFINITY;
        for (size_t i = 0; i < L; i++) {
            if (i == peak) continue;
            double key = i == 0 ? results[0] : fabs(results[i]);
            if (key > second) second = key;
        }
        *peak_margin = second > 0.0 ? best / second : INFINITY;
    }

    /* :256-271 wrap to a signed lag and choose the overlapping segments */
    const double *s0, *s1, *t0, *t1;
    long l = (long)peak;
    if (l >= (long)N) {
        l = (l % (long)N) - (long)N;
        s0 = source;         s1 = source + l + (long)N;
        t0 = sample - l;     t1 = sample + N;
    } else {
        s0 = source + l;     s1 = source + l + (long)N;
        t0 = sample;         t1 = sample + N;
    }
    *lag = l;
    *coefficient = oracle_pearson_coefficient(s0, s1, t0, t1); /* :272 */

    /* :276 */
    if (*coefficient != *coefficient) goto finish;
    ret = 0;

finish:
    free(sample); free(results); free(X); free(Y);
    return ret;
}

int oracle_cross_correlation(const double *source, const double *sample, size_t N, long *lag,
                             double *coefficient)
{
    return oracle_cross_correlation_ex(source, sample, N, lag, coefficient, NULL, NULL);
}

int oracle_cross_correlation_f32(const float *source, const float *sample, size_t N, long *lag,
                                 double *coefficient)
{
    double *s = malloc(sizeof(double) * 2 * N);
    double *t = malloc(sizeof(double) * N);
    int ret = -1;
    if (s && t) {
        for (size_t i = 0; i < 2 * N; i++) s[i] = (double)source[i];
        for (size_t i = 0; i < N; i++) t[i] = (double)sample[i];
        ret = oracle_cross_correlation(s, t, N, lag, coefficient);
    }
    free(s); free(t);
    return ret;
}

/* ---- synthetic pairs -------------------------------------------------- */

static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline uint64_t stream_key(uint64_t seed, uint64_t pair, uint64_t stream)
{
    return mix64(mix64(seed) + 0x632BE59BD9B4E019ull * (pair + 1) + stream);
}

static inline float u24(uint64_t h)
{
    /* top 24 bits -> integer in [-2^23, 2^23) -> exact float in [-1,1) */
    const int32_t v = (int32_t)(h >> 40) - (1 << 23);
    return (float)v * (1.0f / 8388608.0f);
}

static inline float z22(uint64_t key, uint64_t n)
{
    const uint64_t h1 = mix64(key + 2 * n), h2 = mix64(key + 2 * n + 1);
    const int32_t a = (int32_t)(h1 >> 42), b = (int32_t)((h1 >> 20) & 0x3FFFFF);
    const int32_t c = (int32_t)(h2 >> 42), d = (int32_t)((h2 >> 20) & 0x3FFFFF);
    const int32_t v = a + b + c + d - (1 << 23); /* [-2^23, 2^23) */
    return (float)v * (1.0f / 4194304.0f);       /* [-2, 2) */
}

void oracle_synth_pair(uint64_t seed, uint64_t pair, size_t N, int noise_shift, float *source,
                       float *sample, int64_t *true_lag)
{
    const uint64_t ks = stream_key(seed, pair, 1), kn = stream_key(seed, pair, 2);
    const uint64_t kl = stream_key(seed, pair, 3);
    const int64_t span = (int64_t)(3 * N / 4);
    const int64_t lag = (int64_t)(mix64(kl) % (uint64_t)(2 * span + 1)) - span;
    const float amp = ldexpf(1.0f, -noise_shift);
    for (size_t j = 0; j < 2 * N; j++) source[j] = u24(mix64(ks + N + j));
    for (size_t n = 0; n < N; n++) {
        const float sig = 0.5f * u24(mix64(ks + (uint64_t)((int64_t)N + (int64_t)n + lag)));
        sample[n] = sig + amp * z22(kn, n);
    }
    if (true_lag) *true_lag = lag;
}
