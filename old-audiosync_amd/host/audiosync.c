/* host/audiosync.c — job state, control API and the growing-window loop.
 *
 * Mirrors the observable behaviour of the reference orchestrator
 * (src/audiosync.c:36-138 for the globals/control functions, :166-284 for
 * audiosync_run) with ONE difference: the two producers are not ffmpeg
 * children (src/ffmpeg_pipe.c, out of scope) but threads copying from
 * caller-provided memory (audiosync_set_feed) in the same 4096-frame steps and
 * with the same signalling protocol (src/ffmpeg_pipe.c:63-149): signal
 * interval_done when an interval boundary is crossed, honour PAUSED_ST /
 * ABORT_ST between steps, zero-fill a short track and signal once more.
 * The consumer side -- wait for both producers, correlate the prefixes, first
 * interval with coefficient >= MIN_CONFIDENCE wins, lag converted to
 * milliseconds -- follows src/audiosync.c:226-259.  Where the reference calls
 * cross_correlation() (:246) and so re-plans, re-allocates and re-reads both whole
 * prefixes every interval, this loop keeps the tracks resident on the GPU
 * (asx_stream): each interval uploads only the frames that arrived since the
 * last one and reuses the plan of its length.  Result and error behaviour per
 * interval are those of cross_correlation().
 */
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <poll.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>
#include <audiosync/xcorr_hip.h>

volatile global_status_t global_status = IDLE_ST;
volatile int global_debug = 0;
pthread_mutex_t mutex = PTHREAD_MUTEX_INITIALIZER;
pthread_cond_t interval_done = PTHREAD_COND_INITIALIZER;
pthread_cond_t read_continue = PTHREAD_COND_INITIALIZER;

/* 3, 6, 10, 15, 20, 30 seconds of sample; the source side is always twice that
 * (src/audiosync.c:50-70). */
#define N_INTERVALS 6
static const size_t interv_sample[N_INTERVALS] = {
    3 * SAMPLE_RATE, 6 * SAMPLE_RATE, 10 * SAMPLE_RATE,
    15 * SAMPLE_RATE, 20 * SAMPLE_RATE, 30 * SAMPLE_RATE,
};
static const size_t interv_source[N_INTERVALS] = {
    2 * 3 * SAMPLE_RATE, 2 * 6 * SAMPLE_RATE, 2 * 10 * SAMPLE_RATE,
    2 * 15 * SAMPLE_RATE, 2 * 20 * SAMPLE_RATE, 2 * 30 * SAMPLE_RATE,
};
#define LEN_SAMPLE ((size_t) 30 * SAMPLE_RATE)
#define LEN_SOURCE ((size_t) 2 * 30 * SAMPLE_RATE)
#define FEED_STEP 4096 /* frames per producer step, src/ffmpeg_pipe.c BUFSIZE */

void audiosync_abort()
{
    pthread_mutex_lock(&mutex);
    global_status = ABORT_ST;
    pthread_cond_broadcast(&interval_done);
    pthread_cond_broadcast(&read_continue);
    pthread_mutex_unlock(&mutex);
}

void audiosync_pause()
{
    pthread_mutex_lock(&mutex);
    global_status = PAUSED_ST;
    pthread_mutex_unlock(&mutex);
}

void audiosync_resume()
{
    pthread_mutex_lock(&mutex);
    global_status = RUNNING_ST;
    pthread_cond_broadcast(&read_continue);
    pthread_mutex_unlock(&mutex);
}

global_status_t audiosync_status()
{
    pthread_mutex_lock(&mutex);
    global_status_t now = global_status;
    pthread_mutex_unlock(&mutex);
    return now;
}

int audiosync_get_debug()
{
    pthread_mutex_lock(&mutex);
    int now = global_debug;
    pthread_mutex_unlock(&mutex);
    return now;
}

void audiosync_set_debug(int do_debug)
{
    pthread_mutex_lock(&mutex);
    global_debug = do_debug;
    pthread_mutex_unlock(&mutex);
}

char *status_to_string(global_status_t status)
{
    switch (status) {
    case IDLE_ST:    return "idle";
    case RUNNING_ST: return "running";
    case PAUSED_ST:  return "paused";
    case ABORT_ST:   return "aborting";
    default:         return "unknown";
    }
}

int audiosync_setup(const char *stream_name)
{
    UNUSED(stream_name);
    LOG("audiosync_setup: PulseAudio capture is out of scope of the MI355X build");
    return -1;
}

/* ---- producers: from caller memory, or f64le frames from a file / FIFO ------- */
struct feed {
    const double *data;   /* in-memory track (audiosync_set_feed), or NULL */
    size_t len;
    char *path;           /* file or FIFO of f64le mono frames (audiosync_set_feed_files), or NULL */
};
static struct feed feed_source, feed_sample;
static unsigned feed_frames_per_ms;

static void feed_clear(struct feed *f)
{
    free(f->path);
    f->data = NULL; f->len = 0; f->path = NULL;
}

/* Both setters refuse while a run is in progress (-1): the producers of that run are reading the feed. */
int audiosync_set_feed(const double *source, size_t source_len, const double *sample,
                       size_t sample_len, unsigned frames_per_ms)
{
    pthread_mutex_lock(&mutex);
    if (global_status != IDLE_ST) {
        pthread_mutex_unlock(&mutex);
        return -1;
    }
    feed_clear(&feed_source); feed_clear(&feed_sample);
    feed_source.data = source; feed_source.len = source_len;
    feed_sample.data = sample; feed_sample.len = sample_len;
    feed_frames_per_ms = frames_per_ms;
    pthread_mutex_unlock(&mutex);
    return 0;
}

/* The wire format of the reference's producers: what `ffmpeg ... -f f64le -ac 1 -ar 48000 pipe:1`
 * writes (src/capture/linux_capture.c:370, src/download/linux_download.c:41) and src/ffmpeg_pipe.c:68-81
 * reads in chunks.  Each path may be a regular file or a FIFO an ffmpeg process is writing to; the
 * producers read it in the same 4096-frame steps with the same signalling.  Returns -1 if a path is NULL. */
int audiosync_set_feed_files(const char *source_path, const char *sample_path)
{
    if (source_path == NULL || sample_path == NULL) return -1;
    char *a = strdup(source_path), *b = strdup(sample_path);
    if (a == NULL || b == NULL) {
        perror("audiosync: strdup for the feed paths failed");
        free(a); free(b);
        return -1;
    }
    pthread_mutex_lock(&mutex);
    if (global_status != IDLE_ST) {
        pthread_mutex_unlock(&mutex);
        free(a); free(b);
        return -1;
    }
    feed_clear(&feed_source); feed_clear(&feed_sample);
    feed_source.path = a;
    feed_sample.path = b;
    feed_frames_per_ms = 0;
    pthread_mutex_unlock(&mutex);
    return 0;
}

struct producer_args {
    struct ffmpeg_data *out;
    struct feed in;
    unsigned frames_per_ms;
};

static int aborting(void)
{
    pthread_mutex_lock(&mutex);
    const int a = (global_status == ABORT_ST);
    pthread_mutex_unlock(&mutex);
    return a;
}

/* Reads up to `want` bytes from a descriptor opened O_NONBLOCK; returns the bytes read (short only at end of file),
 * -1 on error, -2 when the job was aborted while waiting.  A FIFO whose writer has not started yet, or stalls, must not
 * hold the run hostage: poll() with a timeout, the status checked every round.  End of file = a zero-byte read while
 * poll() reports the descriptor readable or hung up (a FIFO nobody has opened for writing yet reports neither). */
static ssize_t read_chunk(int fd, void *dst, size_t want)
{
    size_t got = 0;
    while (got < want) {
        struct pollfd pf = { .fd = fd, .events = POLLIN, .revents = 0 };
        const int pr = poll(&pf, 1, 100);
        if (pr < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (aborting()) return -2;
        if (pr == 0) continue;
        if (pf.revents & POLLNVAL) { errno = EBADF; return -1; }
        const ssize_t r = read(fd, (char *) dst + got, want - got);
        if (r < 0) {
            if (errno == EINTR || errno == EAGAIN) continue;
            return -1;
        }
        if (r == 0) {
            /* poll() returned at once and there is nothing to read: end of file (POLLIN / POLLHUP), or an error
             * condition on the descriptor (POLLERR alone) -- which would otherwise bring poll() back immediately, round
             * after round, at 100 % CPU */
            if (pf.revents & (POLLIN | POLLHUP)) break;
            if (pf.revents & POLLERR) { errno = EIO; return -1; }
            continue;
        }
        got += (size_t) r;
    }
    /* whole frames only: a writer that dies inside a frame leaves a tail that is not a double */
    return (ssize_t) (got - got % sizeof(double));
}

static void *producer(void *arg)
{
    struct producer_args *pa = arg;
    struct ffmpeg_data *d = pa->out;
    size_t interval = 0;
    int fd = -1;
    size_t avail = pa->in.data ? (pa->in.len < d->total_len ? pa->in.len : d->total_len) : 0;
    if (pa->in.path) {
        fd = open(pa->in.path, O_RDONLY | O_NONBLOCK); /* a FIFO without a writer yet opens at once */
        if (fd < 0) {
            /* like a failed ffmpeg child (src/ffmpeg_pipe.c:59-62): the job is aborted */
            perror("audiosync: open of the feed file failed");
            audiosync_abort();
            return NULL;
        }
        avail = d->total_len;
    }

    while (d->len < avail) {
        size_t step = avail - d->len < FEED_STEP ? avail - d->len : FEED_STEP;
        if (fd >= 0) {
            /* src/ffmpeg_pipe.c:68-81: one chunk of f64le frames; end of file ends the track */
            const ssize_t got = read_chunk(fd, d->buf + d->len, step * sizeof(*d->buf));
            if (got == -2) { /* aborted while waiting for the writer */
                close(fd);
                return NULL;
            }
            if (got < 0) {
                perror("audiosync: read of the feed file failed");
                close(fd);
                audiosync_abort();
                return NULL;
            }
            step = (size_t) got / sizeof(*d->buf);
            if (step == 0) break;
        } else {
        memcpy(d->buf + d->len, pa->in.data + d->len, step * sizeof(*d->buf));
        }
        if (pa->frames_per_ms) {
            struct timespec ts = { 0, (long)(1000000.0 * step / pa->frames_per_ms) };
            nanosleep(&ts, NULL);
        }
        pthread_mutex_lock(&mutex);
        d->len += step;
        if (interval < d->n_intervals && d->len >= d->intervals[interval]) {
            pthread_cond_signal(&interval_done);
            interval++;
        }
        /* pause / abort are observed between steps, like src/ffmpeg_pipe.c:99-134 */
        while (global_status == PAUSED_ST) pthread_cond_wait(&read_continue, &mutex);
        const int stop = (global_status == ABORT_ST);
        pthread_mutex_unlock(&mutex);
        if (stop) {
            if (fd >= 0) close(fd);
            return NULL;
        }
    }
    if (fd >= 0) close(fd);
    /* a short track: the tail reads as silence (src/ffmpeg_pipe.c:139-149) */
    if (d->len < d->total_len) {
        memset(d->buf + d->len, 0, (d->total_len - d->len) * sizeof(*d->buf));
        pthread_mutex_lock(&mutex);
        d->len = d->total_len;
        pthread_cond_signal(&interval_done);
        pthread_mutex_unlock(&mutex);
    }
    return NULL;
}

int audiosync_run(const char *yt_title, long *lag)
{
    DEBUG_ASSERT(yt_title); DEBUG_ASSERT(lag);
    pthread_mutex_lock(&mutex);
    DEBUG_ASSERT(global_status == IDLE_ST);
    global_status = RUNNING_ST;
    pthread_mutex_unlock(&mutex);
    int ret = -1;
    double confidence;
    pthread_t cap_th, down_th;
    int cap_started = 0, down_started = 0;
    /* page-locked: every interval uploads the frames that are new since the previous one (the reference allocates its
     * source with fftw_alloc_real, src/audiosync.c:189: the allocation its transform backend wants) */
    double *sample = asx_host_malloc(LEN_SAMPLE * sizeof(*sample));
    double *source = asx_host_malloc(LEN_SOURCE * sizeof(*source));
    struct ffmpeg_data cap_args = {
        .title = "", .buf = sample, .len = 0, .total_len = LEN_SAMPLE,
        .intervals = interv_sample, .n_intervals = N_INTERVALS,
    };
    struct ffmpeg_data down_args = {
        .title = yt_title, .buf = source, .len = 0, .total_len = LEN_SOURCE,
        .intervals = interv_source, .n_intervals = N_INTERVALS,
    };
    struct producer_args cap_pa, down_pa;
    int own_paths = 0;
    asx_stream *stream = NULL;

    if (sample == NULL || source == NULL) {
        perror("audiosync: track buffer malloc failed");
        goto finish;
    }
    pthread_mutex_lock(&mutex);
    cap_pa.out = &cap_args; cap_pa.in = feed_sample; cap_pa.frames_per_ms = feed_frames_per_ms;
    down_pa.out = &down_args; down_pa.in = feed_source; down_pa.frames_per_ms = feed_frames_per_ms;
    /* the run keeps its own copies of the paths (the setters refuse while it is in progress, this is the belt) */
    if (cap_pa.in.path) cap_pa.in.path = strdup(cap_pa.in.path);
    if (down_pa.in.path) down_pa.in.path = strdup(down_pa.in.path);
    /* (what the setters guard under the mutex is also read under it) */
    const int lost_path = (feed_sample.path && !cap_pa.in.path) || (feed_source.path && !down_pa.in.path);
    pthread_mutex_unlock(&mutex);
    own_paths = 1;
    if (lost_path) {
        perror("audiosync: strdup for the feed paths failed");
        goto finish;
    }
    if ((cap_pa.in.data == NULL && cap_pa.in.path == NULL) || (down_pa.in.data == NULL && down_pa.in.path == NULL)) {
        /* nothing to record or download: the reference's producers fail and abort the job */
        LOG("no feed configured (audiosync_set_feed / audiosync_set_feed_files); aborting");
        goto finish;
    }
    if (pthread_create(&cap_th, NULL, &producer, &cap_pa) != 0) {
        perror("audiosync: pthread_create for cap_th failed");
        goto finish;
    }
    cap_started = 1;
    if (pthread_create(&down_th, NULL, &producer, &down_pa) != 0) {
        perror("audiosync: pthread_create for down_th failed");
        goto finish;
    }
    down_started = 1;

    /* both tracks live on the GPU for the whole run; freed at finish */
    stream = asx_stream_create(LEN_SAMPLE, -1);
    if (stream == NULL) {
        fprintf(stderr, "audiosync: no GPU stream: %s\n", asx_last_error());
        goto finish;
    }

    LOG("starting interval loop");
    for (size_t i = 0; i < N_INTERVALS; i++) {
        pthread_mutex_lock(&mutex);
        while ((cap_args.len < interv_sample[i] || down_args.len < interv_source[i])
               && global_status != ABORT_ST) {
            pthread_cond_wait(&interval_done, &mutex);
        }
        const int aborted = (global_status == ABORT_ST);
        const size_t have_cap = cap_args.len, have_down = down_args.len;
        pthread_mutex_unlock(&mutex);
        if (aborted) break;

        LOG("next interval (%ld): cap=%ld down=%ld", (long) i, (long) have_cap, (long) have_down);

        /* upload what is new since the previous interval, then correlate the prefixes */
        size_t up_src = 0, up_smp = 0;
        asx_stream_lengths(stream, &up_src, &up_smp);
        if (asx_stream_append_f64(stream, source + up_src, interv_source[i] - up_src,
                                  sample + up_smp, interv_sample[i] - up_smp) < 0)
            continue;
        if (asx_stream_xcorr(stream, interv_sample[i], lag, &confidence) < 0)
            continue;
        LOG("%ld frames of delay with a confidence of %f", *lag, confidence);
        if (confidence >= MIN_CONFIDENCE) {
            *lag = round((double) (*lag) * FRAMES_TO_MS);
            ret = 0;
            break;
        }
    }

finish:
    if (stream) asx_stream_destroy(stream);
    audiosync_abort();
    if (cap_started) pthread_join(cap_th, NULL);
    if (down_started) pthread_join(down_th, NULL);
    asx_host_free(sample);
    asx_host_free(source);
    if (own_paths) { free(cap_pa.in.path); free(down_pa.in.path); }
    pthread_mutex_lock(&mutex);
    global_status = IDLE_ST;
    pthread_mutex_unlock(&mutex);
    LOG("finished run");
    return ret;
}
