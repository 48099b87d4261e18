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
