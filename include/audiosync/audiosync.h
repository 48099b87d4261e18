/* include/audiosync/audiosync.h — MI355X build of the audiosync public header.
 *
 * Source-compatible with the parts of the reference header
 * (reference: include/audiosync/audiosync.h:12-119) that the hot path and its
 * callers use: the audio-format constants, DEBUG_ASSERT / UNUSED / LOG, the
 * status enum and control functions, struct ffmpeg_data and audiosync_run().
 * What sits behind audiosync_run() here is the interval loop of
 * src/audiosync.c:226-259 fed from memory (see audiosync_set_feed below); the
 * ffmpeg / PulseAudio / youtube-dl producers are out of scope of this build.
 */
#ifndef AUDIOSYNC_AUDIOSYNC_H
#define AUDIOSYNC_AUDIOSYNC_H

#include <assert.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- audio format shared by both tracks (48 kHz mono) -------------------- */
#define SAMPLE_RATE 48000
#define SAMPLE_RATE_STR "48000"
#define NUM_CHANNELS 1
#define NUM_CHANNELS_STR "1"
#define MAX_SECONDS_STR "30"                        /* last interval, seconds */
#define FRAMES_TO_MS (1000.0 / (double) SAMPLE_RATE) /* frames -> milliseconds */
#define MIN_CONFIDENCE 0.95                          /* accepted coefficient   */

/* ---- small helpers --------------------------------------------------------- */
#define UNUSED(x) (void)(x)

#ifdef NDEBUG
# define DEBUG_ASSERT(x) do {} while(0)
#else
# define DEBUG_ASSERT(x) assert(x)
#endif

/* stderr logging, switched at run time by audiosync_set_debug() */
#define DEBUG_COLOR "\x1B[36m"
#define END_COLOR "\x1B[0m"
extern volatile int global_debug;
#define LOG(str, ...) \
    do { \
        if (global_debug) { \
            fprintf(stderr, DEBUG_COLOR "audiosync: " END_COLOR str "\n", \
                    ##__VA_ARGS__); \
        } \
    } while (0);
extern int audiosync_get_debug();
extern void audiosync_set_debug(int do_debug);

/* ---- job state ------------------------------------------------------------- */
typedef enum {
    IDLE_ST,     /* nothing running                          */
    RUNNING_ST,  /* audiosync_run() in progress              */
    PAUSED_ST,   /* producers paused                         */
    ABORT_ST     /* stop requested, run is winding down      */
} global_status_t;
extern volatile global_status_t global_status;
extern char *status_to_string(global_status_t status);

extern pthread_mutex_t mutex;          /* guards the globals above and the producers' len */
extern pthread_cond_t interval_done;   /* a producer completed one of its intervals       */
extern pthread_cond_t read_continue;   /* resume after a pause                            */

extern void audiosync_abort();
extern void audiosync_pause();
extern void audiosync_resume();
extern global_status_t audiosync_status();

/* ---- producers ------------------------------------------------------------- */
/* What a producer thread fills and signals through (same members as the reference). */
struct ffmpeg_data {
    const char *title;         /* what to fetch (download side only)       */
    double *buf;               /* destination buffer                       */
    size_t len;                /* frames written so far                    */
    const size_t total_len;    /* capacity of buf                          */
    const size_t *intervals;   /* frame counts at which to signal          */
    const size_t n_intervals;  /* how many of those                        */
};

/* Optional PulseAudio sink setup in the reference.  This build has no audio
 * capture: it always returns -1 (and logs that it is out of scope). */
extern int audiosync_setup(const char *stream_name);

/* Runs the growing-window loop: waits until both producers have delivered the
 * next interval, calls cross_correlation() on the prefixes, stops at the first
 * interval whose coefficient reaches MIN_CONFIDENCE and returns the lag in
 * milliseconds through *lag.  0 on success, -1 otherwise. */
extern int audiosync_run(const char *yt_title, long int *lag);

/* ---- extension of this build: in-memory producers --------------------------- */
/* Provide the two tracks audiosync_run() will "record" and "download":
 * source has 2*30*SAMPLE_RATE doubles worth of capacity, sample 30*SAMPLE_RATE;
 * shorter buffers are zero-filled like src/ffmpeg_pipe.c:139-149 does.
 * frames_per_ms throttles delivery (0 = as fast as possible).  The buffers must
 * stay valid until audiosync_run() returns.  Returns 0, or -1 while a run is in
 * progress (its producers are reading the feed: call it in IDLE_ST only). */
extern int audiosync_set_feed(const double *source, size_t source_len,
                              const double *sample, size_t sample_len,
                              unsigned frames_per_ms);

/* The same from two files or FIFOs of f64le mono frames at SAMPLE_RATE -- the wire format the
 * reference's producers read from their ffmpeg children (`-f f64le`, src/capture/linux_capture.c:370,
 * src/download/linux_download.c:41; chunked reads of src/ffmpeg_pipe.c:68-81).  source_path is the
 * downloaded track (up to 2*30 s are read), sample_path the recorded one (up to 30 s); a short file is
 * zero-filled.  A path that cannot be opened aborts the run like a failed ffmpeg child; a FIFO whose writer
 * has not started, or stalls, is waited for with the status checked every 100 ms, so audiosync_abort() ends
 * the run.  Returns 0, or -1 for a NULL path or while a run is in progress. */
extern int audiosync_set_feed_files(const char *source_path, const char *sample_path);

/* Debug aid replacing the reference's compile-time PLOT/gnuplot dumps
 * (src/cross_correlation.c:168-184,280-296): writes the two segments that
 * pearson_coefficient() compares for `lag` (src/cross_correlation.c:256-271) as
 * CSV lines "index,source,sample".  Returns the number of lines, -1 on error. */
extern long audiosync_dump_segments_csv(const char *path, const double *source,
                                        const double *sample, size_t sample_len, long lag);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOSYNC_AUDIOSYNC_H */
