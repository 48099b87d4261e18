/* tests/c/asx_stub.c — TEST INFRASTRUCTURE ONLY: a CPU stand-in for the asx_* entry points that the host C files call
 * (include/audiosync/xcorr_hip.h), so that the host C code -- the plan cache with its pin / doom reference counts
 * (host/cross_correlation.c:46-118), the producers and the interval loop (host/audiosync.c) -- can run under
 * AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a box without a GPU (GPU sanitizers are not
 * available on this pool; the reference's Debug build is -fsanitize=undefined,address, CMakeLists.txt:15-16).
 * It is linked ONLY into the sanitizer test programs of this directory, never into libaudiosync*.so: the product has no
 * CPU fallback (tests/test_abi.py::test_no_oracle_or_fallback_in_product_sources).
 *
 * Arithmetic: the definition, O(N^2): r[k] = sum_n source[(n+k) mod 2N] * sample[n], the reference's peak rule
 * (src/cross_correlation.c:52-67), lag wrap (:256-271) and two-pass Pearson (:74-116).  A plan is a heap object that
 * every call reads at its start and its end, with a short sleep in between: a plan freed under a running call is a
 * heap-use-after-free ASan reports, an unlocked access a race TSan reports. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <audiosync/xcorr_hip.h>

struct asx_plan {
    size_t n;
    unsigned magic;
    unsigned long calls;
    pthread_mutex_t lock;   /* the real plan serialises its own use too (asx_api.hip: asx_plan::lock) */
};
#define STUB_MAGIC 0xA5C0FFEEu

const char *asx_last_error(void) { return "asx_stub"; }
int asx_abi_version(void) { return 2; }
int asx_device_count(void) { return 1; }
int asx_current_device(void) { return 0; }

asx_plan *asx_plan_create(size_t sample_len, size_t max_batch, int device)
{
    (void)max_batch; (void)device;
    if (sample_len == 0) return NULL;
    asx_plan *p = malloc(sizeof(*p));
    if (!p) return NULL;
    p->n = sample_len; p->magic = STUB_MAGIC; p->calls = 0;
    pthread_mutex_init(&p->lock, NULL);
    usleep(300); /* building a plan takes a while: widens the window in which two callers want the same length */
    return p;
}

void asx_plan_destroy(asx_plan *p)
{
    if (!p) return;
    if (p->magic != STUB_MAGIC) { fprintf(stderr, "asx_stub: destroying a dead plan\n"); abort(); }
    p->magic = 0;
    pthread_mutex_destroy(&p->lock);
    free(p);
}

static double pearson(const double *a, const double *b, size_t n)
{
    double ma = 0, mb = 0;
    for (size_t i = 0; i < n; i++) { ma += a[i]; mb += b[i]; }
    ma /= (double)n; mb /= (double)n;
    double sab = 0, saa = 0, sbb = 0;
    for (size_t i = 0; i < n; i++) {
        const double da = a[i] - ma, db = b[i] - mb;
        sab += da * db; saa += da * da; sbb += db * db;
    }
    return sab / sqrt(saa * sbb);
}

void *asx_host_malloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
int asx_host_free(void *ptr) { free(ptr); return 0; }

int asx_pearson_f64(const double *a, const double *b, size_t n, int device, double *out)
{
    (void)device;
    if (!a || !b || !out) return -1;
    *out = pearson(a, b, n);
    return 0;
}

static int xcorr(const double *source, const double *sample, size_t N, long *lag, double *coef)
{
    const size_t L = 2 * N;
    size_t best = 0;
    double best_key = 0;
    for (size_t k = 0; k < L; k++) {
        double r = 0;
        for (size_t n = 0; n < N; n++) r += source[(n + k) % L] * sample[n];
        const double key = k == 0 ? r : fabs(r);   /* src/cross_correlation.c:56,59 */
        if (k == 0 || key > best_key) { best_key = key; best = k; }
    }
    long l = (long)best;
    const double *s0, *t0;
    size_t len;
    if (l >= (long)N) { l = (l % (long)N) - (long)N; s0 = source; t0 = sample - l; len = (size_t)((long)N + l); }
    else { s0 = source + l; t0 = sample; len = N; }
    *lag = l;
    *coef = len ? pearson(s0, t0, len) : NAN;
    return (*coef == *coef) ? 0 : -1;
}

int asx_xcorr_f64(asx_plan *p, const double *source, const double *sample, long *lag, double *coefficient)
{
    if (!p || !source || !sample || !lag || !coefficient) return -1;
    pthread_mutex_lock(&p->lock);
    if (p->magic != STUB_MAGIC) { fprintf(stderr, "asx_stub: call on a dead plan\n"); abort(); }
    const size_t n = p->n;
    p->calls++;
    const int rc = xcorr(source, sample, n, lag, coefficient);
    usleep(100);
    if (p->magic != STUB_MAGIC || p->n != n) { fprintf(stderr, "asx_stub: plan changed under a call\n"); abort(); }
    pthread_mutex_unlock(&p->lock);
    return rc;
}

/* growing-window streams (host/audiosync.c): the doubles are kept, every prefix is correlated from scratch */
struct asx_stream {
    size_t cap, n_src, n_smp;
    double *src, *smp;
};

asx_stream *asx_stream_create(size_t max_sample_len, int device)
{
    (void)device;
    asx_stream *s = calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->cap = max_sample_len;
    s->src = malloc(sizeof(double) * 2 * max_sample_len);
    s->smp = malloc(sizeof(double) * max_sample_len);
    if (!s->src || !s->smp) { free(s->src); free(s->smp); free(s); return NULL; }
    return s;
}
void asx_stream_destroy(asx_stream *s) { if (s) { free(s->src); free(s->smp); free(s); } }
int asx_stream_lengths(const asx_stream *s, size_t *a, size_t *b) { if (!s) return -1; if (a) *a = s->n_src; if (b) *b = s->n_smp; return 0; }
int asx_stream_reset(asx_stream *s) { if (!s) return -1; s->n_src = s->n_smp = 0; return 0; }
int asx_stream_append_f64(asx_stream *s, const double *src, size_t ns, const double *smp, size_t nt)
{
    if (!s || s->n_src + ns > 2 * s->cap || s->n_smp + nt > s->cap) return -1;
    if (ns) memcpy(s->src + s->n_src, src, ns * sizeof(double));
    if (nt) memcpy(s->smp + s->n_smp, smp, nt * sizeof(double));
    s->n_src += ns; s->n_smp += nt;
    return 0;
}
int asx_stream_xcorr(asx_stream *s, size_t n, long *lag, double *coef)
{
    if (!s || s->n_smp < n || s->n_src < 2 * n) return -1;
    return xcorr(s->src, s->smp, n, lag, coef);
}
