/* host/cross_correlation.c — the reference's hot-path API on the MI355X layer.
 *
 * Plain C.  cross_correlation() and pearson_coefficient() keep the signatures
 * and the error behaviour of the reference (src/cross_correlation.c:74-75,
 * 133-135; contract in SURVEY.md section 8b) and forward to the gfx950 C-ABI
 * (audiosync/xcorr_hip.h).  What the reference does per call -- FFTW plans
 * under cc_mutex (:33-36), four aligned allocations (:159,187-201) -- becomes a
 * small cache of asx_plan objects keyed by sample_len, guarded by one mutex, so
 * that the six interval lengths of src/audiosync.c:50-57 each build their
 * tables and HBM workspaces once.
 *
 * There is no CPU path here: without a working GPU both functions fail the way
 * the reference fails on an allocation error (-1 / NaN, message on stderr).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>
#include <audiosync/xcorr_hip.h>

#define PLAN_CACHE_SLOTS 8

/* One cache entry.  `users` counts the callers that are inside asx_xcorr_f64 with this plan: an
 * entry is only ever destroyed when nobody uses it, so a concurrent caller with a ninth length
 * (or audiosync_release_plans) cannot free a plan under another thread.  A plan belongs to the HIP
 * device that was current when it was built, so the device is part of the key. */
struct cached_plan {
    size_t sample_len;
    int device;
    asx_plan *plan;
    unsigned users;
    int doomed;            /* evicted or released while in use: destroyed by its last user */
    unsigned long last_use;
};

/* Concurrent callers are legal in the reference (SURVEY.md 8b "Threading"); the
 * cache is shared, each plan serialises its own use internally. */
static pthread_mutex_t cache_mutex = PTHREAD_MUTEX_INITIALIZER;
static struct cached_plan cache[PLAN_CACHE_SLOTS];
static unsigned long use_clock;

/* Returns a pinned entry (users incremented) or NULL.  *transient is set when the cache was full of
 * plans in use: the caller then owns a private plan and destroys it itself. */
static struct cached_plan *plan_acquire(size_t sample_len, asx_plan **transient)
{
    struct cached_plan *found = NULL;
    const int device = asx_current_device();
    *transient = NULL;
    pthread_mutex_lock(&cache_mutex);
    int victim = -1;
    for (int i = 0; i < PLAN_CACHE_SLOTS; i++) {
        struct cached_plan *c = &cache[i];
        if (c->plan && !c->doomed && c->sample_len == sample_len && c->device == device) {
            found = c;
            break;
        }
        if (c->plan && c->users > 0) continue;                /* pinned (doomed or not): not a victim */
        if (victim < 0) victim = i;
        else if (!c->plan && cache[victim].plan) victim = i;  /* prefer an empty slot */
        else if (c->plan && cache[victim].plan && c->last_use < cache[victim].last_use) victim = i;
    }
    if (!found) {
        /* Building a plan (tables, HBM workspaces) under the cache lock keeps two first callers with
         * the same length from building it twice; it happens once per length. */
        asx_plan *fresh = asx_plan_create(sample_len, 1, device);
        if (fresh && victim >= 0) {
            struct cached_plan *c = &cache[victim];
            if (c->plan) asx_plan_destroy(c->plan);           /* users == 0: nobody holds it */
            c->plan = fresh;
            c->sample_len = sample_len;
            c->device = device;
            c->users = 0;
            c->doomed = 0;
            found = c;
        } else if (fresh) {
            *transient = fresh;                               /* every slot is in use right now */
        }
    }
    if (found) {
        found->users++;
        found->last_use = ++use_clock;
    }
    pthread_mutex_unlock(&cache_mutex);
    return found;
}

static void plan_release(struct cached_plan *c)
{
    asx_plan *dead = NULL;
    pthread_mutex_lock(&cache_mutex);
    if (--c->users == 0 && c->doomed) {
        dead = c->plan;
        c->plan = NULL;
        c->doomed = 0;
    }
    pthread_mutex_unlock(&cache_mutex);
    if (dead) asx_plan_destroy(dead);
}

/* Drops every cached plan (frees the HBM workspaces).  Safe at any time: a plan that is in use is
 * destroyed by the call that is using it, when that call returns. */
void audiosync_release_plans(void)
{
    pthread_mutex_lock(&cache_mutex);
    for (int i = 0; i < PLAN_CACHE_SLOTS; i++) {
        if (!cache[i].plan) continue;
        if (cache[i].users > 0) {
            cache[i].doomed = 1;
        } else {
            asx_plan_destroy(cache[i].plan);
            cache[i].plan = NULL;
        }
    }
    pthread_mutex_unlock(&cache_mutex);
}

int cross_correlation(double *source, double *sample, const size_t sample_len, long *lag,
                      double *coefficient)
{
    DEBUG_ASSERT(source); DEBUG_ASSERT(sample);
    DEBUG_ASSERT(lag); DEBUG_ASSERT(coefficient);
    DEBUG_ASSERT(sample_len > 0);

    asx_plan *transient = NULL;
    struct cached_plan *entry = plan_acquire(sample_len, &transient);
    asx_plan *plan = entry ? entry->plan : transient;
    if (plan == NULL) {
        /* same class of failure as a failed fftw_alloc_* in the reference: -1, outputs untouched */
        fprintf(stderr, "audiosync: no GPU plan for %zu frames: %s\n", sample_len, asx_last_error());
        return -1;
    }
    const int ret = asx_xcorr_f64(plan, source, sample, lag, coefficient);
    if (entry) plan_release(entry);
    else asx_plan_destroy(transient);
    if (ret == 0) {
        LOG("%ld frames of delay with a confidence of %f", *lag, *coefficient);
    }
    return ret;
}

double pearson_coefficient(double *source_start, const double *source_end, double *sample_start,
                           const double *sample_end)
{
    DEBUG_ASSERT(source_start); DEBUG_ASSERT(source_end);
    DEBUG_ASSERT(source_end - source_start > 0);
    DEBUG_ASSERT(sample_start); DEBUG_ASSERT(sample_end);
    DEBUG_ASSERT(sample_end - sample_start > 0);
    UNUSED(sample_end);

    double value = NAN;
    const size_t n = (size_t)(source_end - source_start);
    if (asx_pearson_f64(source_start, sample_start, n, -1, &value) != 0) {
        fprintf(stderr, "audiosync: pearson_coefficient failed on the GPU: %s\n", asx_last_error());
        return NAN;
    }
    return value;
}

/* CSV instead of gnuplot: the segments the coefficient is computed on (see audiosync.h). */
long audiosync_dump_segments_csv(const char *path, const double *source, const double *sample,
                                 size_t sample_len, long lag)
{
    if (!path || !source || !sample || sample_len == 0) return -1;
    if (lag >= (long) sample_len || lag < -(long) sample_len) return -1;
    /* src/cross_correlation.c:256-271 */
    const double *s0 = lag < 0 ? source : source + lag;
    const double *t0 = lag < 0 ? sample - lag : sample;
    const size_t n = lag < 0 ? (size_t) ((long) sample_len + lag) : sample_len;
    FILE *f = fopen(path, "w");
    if (f == NULL) {
        perror("audiosync: dump fopen failed");
        return -1;
    }
    fprintf(f, "index,source,sample\n");
    for (size_t i = 0; i < n; i++) fprintf(f, "%zu,%.17g,%.17g\n", i, s0[i], t0[i]);
    fclose(f);
    return (long) n;
}
