/* oracle/cpu_baseline.c — TEST INFRASTRUCTURE ONLY: the CPU side of bench.py's `cpu_baseline` leg.
 *
 * One call = the reference's own cost model for cross_correlation()
 * (/root/reference/src/cross_correlation.c:133-307), nothing hoisted out of the call:
 *   four aligned allocations and the zero-padded copy of the sample        :159-166,187-201
 *   two pthreads, each: plan a forward r2c under a mutex, execute, destroy :26-46,204-229
 *   conj-multiply                                                          :232-233
 *   plan + execute + destroy the c2r                                       :237-239
 *   max_abs_index, lag wrap, pearson_coefficient, NaN gate, four frees     :242-306
 *
 * Transform backend, chosen at run time and reported by oracle_baseline_backend():
 *   "fftw3"  libfftw3.so.3 found with dlopen (what the reference links: cmake/FindFFTW.cmake:14,
 *            setup.py:12) -- the seven FFTW entry points the reference calls are bound by name;
 *   "port"   this directory's own float64 DFT (fft64.c), when no FFTW is installed on the node.
 * Nothing here is shipped or measured as product.
 *
 * Two cost models:
 *   oracle_cross_correlation_faithful   one call exactly as the reference makes it (above): the latency figure;
 *   oracle_worker_*                     what an independent per-core worker of a batch job would do with the same
 *                                       arithmetic: ONE thread, plans (FFTW_ESTIMATE, or the port's twiddle tables) and
 *                                       buffers made once and kept between calls -- the node-throughput figure
 *                                       ("one independent worker per core").
 */
#include "fft64.h"
#include "xcorr_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef double fftw_cpx[2];
typedef void *fftw_plan_t;
static struct {
    int probed, ok;
    void *(*malloc_)(size_t);
    void (*free_)(void *);
    fftw_plan_t (*plan_r2c)(int, double *, fftw_cpx *, unsigned);
    fftw_plan_t (*plan_c2r)(int, fftw_cpx *, double *, unsigned);
    void (*execute)(const fftw_plan_t);
    void (*destroy)(fftw_plan_t);
} fw;
static pthread_mutex_t plan_mutex = PTHREAD_MUTEX_INITIALIZER; /* cc_mutex, src/cross_correlation.c:15 */
#define FFTW_ESTIMATE_FLAG (1U << 6)

static void probe(void)
{
    pthread_mutex_lock(&plan_mutex);
    if (!fw.probed) {
        fw.probed = 1;
        void *h = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libfftw3.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            fw.malloc_ = (void *(*)(size_t))dlsym(h, "fftw_malloc");
            fw.free_ = (void (*)(void *))dlsym(h, "fftw_free");
            fw.plan_r2c = (fftw_plan_t(*)(int, double *, fftw_cpx *, unsigned))dlsym(h, "fftw_plan_dft_r2c_1d");
            fw.plan_c2r = (fftw_plan_t(*)(int, fftw_cpx *, double *, unsigned))dlsym(h, "fftw_plan_dft_c2r_1d");
            fw.execute = (void (*)(const fftw_plan_t))dlsym(h, "fftw_execute");
            fw.destroy = (void (*)(fftw_plan_t))dlsym(h, "fftw_destroy_plan");
            fw.ok = fw.malloc_ && fw.free_ && fw.plan_r2c && fw.plan_c2r && fw.execute && fw.destroy;
        }
    }
    pthread_mutex_unlock(&plan_mutex);
}

const char *oracle_baseline_backend(void)
{
    probe();
    return fw.ok ? "fftw3" : "port";
}

struct fwd_job {          /* struct fftw_data, src/cross_correlation.c:18-22 */
    size_t L;
    double *real;
    void *cpx;
    int rc;
};

static void *fwd_thread(void *arg) /* fft(), src/cross_correlation.c:26-46 */
{
    struct fwd_job *j = arg;
    if (fw.ok) {
        pthread_mutex_lock(&plan_mutex);
        fftw_plan_t p = fw.plan_r2c((int)j->L, j->real, (fftw_cpx *)j->cpx, FFTW_ESTIMATE_FLAG);
        pthread_mutex_unlock(&plan_mutex);
        fw.execute(p);
        pthread_mutex_lock(&plan_mutex);
        fw.destroy(p);
        pthread_mutex_unlock(&plan_mutex);
        j->rc = 0;
    } else {
        j->rc = offt_rfft(j->L, j->real, (ocpx *)j->cpx); /* plans (twiddles, schedule) inside, per call */
    }
    return NULL;
}

static void *xmalloc(size_t bytes) { return fw.ok ? fw.malloc_(bytes) : malloc(bytes); }
static void xfree(void *p) { if (fw.ok) fw.free_(p); else free(p); }

int oracle_cross_correlation_faithful(const double *source, const double *input_sample, size_t N,
                                      long *lag, double *coefficient)
{
    probe();
    int ret = -1;
    const size_t L = 2 * N, H = L / 2 + 1;
    double *sample = xmalloc(sizeof(double) * L);
    ocpx *arr1 = xmalloc(sizeof(ocpx) * H), *arr2 = xmalloc(sizeof(ocpx) * H);
    double *results = xmalloc(sizeof(double) * L);
    if (!sample || !arr1 || !arr2 || !results) goto finish;
    memcpy(sample, input_sample, sizeof(double) * N);
    memset(sample + N, 0, sizeof(double) * (L - N));

    struct fwd_job a = { L, (double *)source, arr1, -1 }, b = { L, sample, arr2, -1 };
    pthread_t ta, tb;
    if (pthread_create(&ta, NULL, fwd_thread, &a) != 0) goto finish;
    if (pthread_create(&tb, NULL, fwd_thread, &b) != 0) { pthread_join(ta, NULL); goto finish; }
    pthread_join(ta, NULL);
    pthread_join(tb, NULL);
    if (a.rc != 0 || b.rc != 0) goto finish;

    for (size_t k = 0; k < H; k++) {
        const double re = arr1[k].re * arr2[k].re + arr1[k].im * arr2[k].im;
        const double im = arr1[k].im * arr2[k].re - arr1[k].re * arr2[k].im;
        arr1[k].re = re; arr1[k].im = im;
    }
    if (fw.ok) {
        fftw_plan_t p = fw.plan_c2r((int)L, (fftw_cpx *)arr1, results, FFTW_ESTIMATE_FLAG);
        fw.execute(p);
        fw.destroy(p);
    } else if (offt_irfft(L, arr1, results) != 0) {
        goto finish;
    }
    const size_t peak = oracle_max_abs_index(results, L);
    const double *s0, *s1, *t0, *t1;
    long l = (long)peak;
    if (l >= (long)N) {
        l = (l % (long)N) - (long)N;
        s0 = source; s1 = source + l + (long)N; t0 = sample - l; t1 = sample + N;
    } else {
        s0 = source + l; s1 = source + l + (long)N; t0 = sample; t1 = sample + N;
    }
    *lag = l;
    *coefficient = oracle_pearson_coefficient(s0, s1, t0, t1);
    if (*coefficient == *coefficient) ret = 0;
finish:
    xfree(sample); xfree(arr1); xfree(arr2); xfree(results);
    return ret;
}

/* ---- per-core worker: plans and buffers kept between calls ------------------------------------------- */
struct oracle_worker {
    size_t N;
    /* fftw3 backend */
    fftw_plan_t pa, pb, pc;
    double *fsrc, *fsmp, *fres;
    fftw_cpx *fa, *fb;
    /* port backend */
    offt_plan *plan;
    ocpx *wtab, *z, *Z, *X, *Y, *scratch;
    double *results, *sample;
};
void oracle_worker_destroy(oracle_worker *w)
{
    if (!w) return;
    if (fw.ok) {
        pthread_mutex_lock(&plan_mutex);
        if (w->pa) fw.destroy(w->pa);
        if (w->pb) fw.destroy(w->pb);
        if (w->pc) fw.destroy(w->pc);
        pthread_mutex_unlock(&plan_mutex);
        if (w->fsrc) fw.free_(w->fsrc);
        if (w->fsmp) fw.free_(w->fsmp);
        if (w->fres) fw.free_(w->fres);
        if (w->fa) fw.free_(w->fa);
        if (w->fb) fw.free_(w->fb);
    }
    offt_plan_destroy(w->plan);
    free(w->wtab); free(w->z); free(w->Z); free(w->X); free(w->Y); free(w->scratch); free(w->results); free(w->sample);
    free(w);
}

oracle_worker *oracle_worker_create(size_t N)
{
    probe();
    if (N == 0) return NULL;
    oracle_worker *w = calloc(1, sizeof(*w));
    if (!w) return NULL;
    w->N = N;
    const size_t L = 2 * N, H = N + 1;
    if (fw.ok) {
        w->fsrc = fw.malloc_(sizeof(double) * L); w->fsmp = fw.malloc_(sizeof(double) * L); w->fres = fw.malloc_(sizeof(double) * L);
        w->fa = fw.malloc_(sizeof(fftw_cpx) * H); w->fb = fw.malloc_(sizeof(fftw_cpx) * H);
        if (!w->fsrc || !w->fsmp || !w->fres || !w->fa || !w->fb) { oracle_worker_destroy(w); return NULL; }
        pthread_mutex_lock(&plan_mutex);
        w->pa = fw.plan_r2c((int)L, w->fsrc, w->fa, FFTW_ESTIMATE_FLAG);
        w->pb = fw.plan_r2c((int)L, w->fsmp, w->fb, FFTW_ESTIMATE_FLAG);
        w->pc = fw.plan_c2r((int)L, w->fa, w->fres, FFTW_ESTIMATE_FLAG);
        pthread_mutex_unlock(&plan_mutex);
        if (!w->pa || !w->pb || !w->pc) { oracle_worker_destroy(w); return NULL; }
        return w;
    }
    const size_t M = N; /* L = 2N is even: packed real transform of length M */
    w->plan = offt_plan_create(M);
    w->wtab = malloc(sizeof(ocpx) * M);
    w->z = malloc(sizeof(ocpx) * M); w->Z = malloc(sizeof(ocpx) * M); w->scratch = malloc(sizeof(ocpx) * M);
    w->X = malloc(sizeof(ocpx) * H); w->Y = malloc(sizeof(ocpx) * H);
    w->results = malloc(sizeof(double) * L); w->sample = malloc(sizeof(double) * L);
    if (!w->plan || !w->wtab || !w->z || !w->Z || !w->scratch || !w->X || !w->Y || !w->results || !w->sample) {
        oracle_worker_destroy(w);
        return NULL;
    }
    for (size_t k = 0; k < M; k++) { /* w_L^k, the real-transform post/pre twiddles */
        const double a = 2.0 * M_PI * (double)k / (double)L;
        w->wtab[k].re = cos(a); w->wtab[k].im = -sin(a);
    }
    return w;
}

/* packed real transform of length L = 2M with the worker's plan and tables (the arithmetic of offt_rfft) */
static int worker_rfft(oracle_worker *w, const double *x, size_t nvalid, ocpx *X)
{
    const size_t M = w->N;
    for (size_t j = 0; j < M; j++) {
        w->z[j].re = 2 * j < nvalid ? x[2 * j] : 0.0;
        w->z[j].im = 2 * j + 1 < nvalid ? x[2 * j + 1] : 0.0;
    }
    if (offt_execute_ws(w->plan, w->z, w->Z, -1, w->scratch) != 0) return -1;
    const ocpx *Z = w->Z;
    X[0].re = Z[0].re + Z[0].im; X[0].im = 0.0;
    X[M].re = Z[0].re - Z[0].im; X[M].im = 0.0;
    for (size_t k = 1; k < M; k++) {
        const ocpx a = Z[k], b = { Z[M - k].re, -Z[M - k].im };
        const ocpx E = { 0.5 * (a.re + b.re), 0.5 * (a.im + b.im) };
        const ocpx D = { 0.5 * (a.re - b.re), 0.5 * (a.im - b.im) };
        const ocpx O = { D.im, -D.re };
        const ocpx t = w->wtab[k];
        X[k].re = E.re + (t.re * O.re - t.im * O.im);
        X[k].im = E.im + (t.re * O.im + t.im * O.re);
    }
    return 0;
}

int oracle_worker_run(oracle_worker *w, const double *source, const double *input_sample, long *lag, double *coefficient)
{
    const size_t N = w->N, L = 2 * N, H = N + 1;
    const double *results;
    const double *sample;
    if (fw.ok) {
        memcpy(w->fsrc, source, sizeof(double) * L);
        memcpy(w->fsmp, input_sample, sizeof(double) * N);
        memset(w->fsmp + N, 0, sizeof(double) * N);
        fw.execute(w->pa);
        fw.execute(w->pb);
        for (size_t k = 0; k < H; k++) {
            const double re = w->fa[k][0] * w->fb[k][0] + w->fa[k][1] * w->fb[k][1];
            const double im = w->fa[k][1] * w->fb[k][0] - w->fa[k][0] * w->fb[k][1];
            w->fa[k][0] = re; w->fa[k][1] = im;
        }
        fw.execute(w->pc);
        results = w->fres;
        sample = w->fsmp;
    } else {
        const size_t M = N;
        memcpy(w->sample, input_sample, sizeof(double) * N);
        if (worker_rfft(w, source, L, w->X) != 0 || worker_rfft(w, input_sample, N, w->Y) != 0) return -2;
        for (size_t k = 0; k < H; k++) {
            const double re = w->X[k].re * w->Y[k].re + w->X[k].im * w->Y[k].im;
            const double im = w->X[k].im * w->Y[k].re - w->X[k].re * w->Y[k].im;
            w->X[k].re = re; w->X[k].im = im;
        }
        /* the arithmetic of offt_irfft */
        ocpx *G = w->z, *g = w->Z;
        const ocpx *X = w->X;
        G[0].re = X[0].re + X[M].re;
        G[0].im = X[0].re - X[M].re;
        for (size_t k = 1; k < M; k++) {
            const ocpx a = X[k], b = { X[M - k].re, -X[M - k].im };
            const ocpx S = { a.re + b.re, a.im + b.im };
            const ocpx D = { a.re - b.re, a.im - b.im };
            const ocpx t = { w->wtab[k].re, -w->wtab[k].im };
            const ocpx u = { t.re * D.re - t.im * D.im, t.re * D.im + t.im * D.re };
            G[k].re = S.re - u.im;
            G[k].im = S.im + u.re;
        }
        if (offt_execute_ws(w->plan, G, g, +1, w->scratch) != 0) return -2;
        for (size_t j = 0; j < M; j++) { w->results[2 * j] = g[j].re; w->results[2 * j + 1] = g[j].im; }
        results = w->results;
        sample = w->sample;
    }
    const size_t peak = oracle_max_abs_index((double *)results, L);
    const double *s0, *s1, *t0, *t1;
    long l = (long)peak;
    if (l >= (long)N) {
        l = (l % (long)N) - (long)N;
        s0 = source; s1 = source + l + (long)N; t0 = sample - l; t1 = sample + N;
    } else {
        s0 = source + l; s1 = source + l + (long)N; t0 = sample; t1 = sample + N;
    }
    *lag = l;
    *coefficient = oracle_pearson_coefficient((double *)s0, s1, (double *)t0, t1);
    return (*coefficient == *coefficient) ? 0 : -1;
}
