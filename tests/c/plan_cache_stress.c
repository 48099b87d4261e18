/* tests/c/plan_cache_stress.c — the plan cache of host/cross_correlation.c:46-118 under contention, for the sanitizer
 * builds (ASan + UBSan, TSan; linked against tests/c/asx_stub.c, no GPU): 8 threads call cross_correlation() with 9
 * different sample lengths (one more than the cache has slots: evictions while other callers hold plans), a ninth thread
 * calls audiosync_release_plans() again and again (plans doomed while in use), every answer is checked.
 * usage: plan_cache_stress [iterations per thread] ; exit code 0 = every call returned the planted delay. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>

void audiosync_release_plans(void);

#define NLEN 9
#define NTHREADS 8
static const size_t lengths[NLEN] = { 16, 20, 24, 30, 36, 40, 48, 50, 60 };
static double *sources[NLEN], *samples[NLEN];
static long delays[NLEN];
static int iterations = 40;
static int stop_releaser; /* atomic accesses only */
static int failures;
static pthread_mutex_t fail_lock = PTHREAD_MUTEX_INITIALIZER;

static unsigned next_rand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void *worker(void *arg)
{
    unsigned seed = 12345u + 977u * (unsigned)(size_t)arg;
    for (int it = 0; it < iterations; it++) {
        const int i = (int)(next_rand(&seed) % NLEN);
        long lag = -777;
        double coef = -7.0;
        const int ret = cross_correlation(sources[i], samples[i], lengths[i], &lag, &coef);
        if (ret != 0 || lag != delays[i] || !(coef > 0.99)) {
            pthread_mutex_lock(&fail_lock);
            failures++;
            fprintf(stderr, "FAIL length %zu: ret %d lag %ld (want %ld) coef %f\n", lengths[i], ret, lag, delays[i], coef);
            pthread_mutex_unlock(&fail_lock);
        }
    }
    return NULL;
}

static void *releaser(void *arg)
{
    (void)arg;
    while (!__atomic_load_n(&stop_releaser, __ATOMIC_ACQUIRE)) {
        audiosync_release_plans();
        usleep(700);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc > 1) iterations = atoi(argv[1]);
    unsigned seed = 99;
    for (int i = 0; i < NLEN; i++) {
        const size_t n = lengths[i];
        sources[i] = malloc(sizeof(double) * 2 * n);
        samples[i] = malloc(sizeof(double) * n);
        for (size_t j = 0; j < 2 * n; j++) sources[i][j] = (double)(next_rand(&seed) % 2001) / 1000.0 - 1.0;
        delays[i] = (long)(n / 4) + i;
        for (size_t j = 0; j < n; j++) samples[i][j] = 0.5 * sources[i][j + (size_t)delays[i]];
    }
    pthread_t th[NTHREADS], rel;
    pthread_create(&rel, NULL, releaser, NULL);
    for (size_t t = 0; t < NTHREADS; t++) pthread_create(&th[t], NULL, worker, (void *)t);
    for (int t = 0; t < NTHREADS; t++) pthread_join(th[t], NULL);
    __atomic_store_n(&stop_releaser, 1, __ATOMIC_RELEASE);
    pthread_join(rel, NULL);
    audiosync_release_plans();
    for (int i = 0; i < NLEN; i++) { free(sources[i]); free(samples[i]); }
    printf("%d calls, %d failures\n", NTHREADS * iterations, failures);
    return failures ? 1 : 0;
}
