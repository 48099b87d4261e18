/* tests/c/run_abort.c — the producers and the state machine of host/audiosync.c under the sanitizer builds (linked against
 * tests/c/asx_stub.c, no GPU): a run fed slowly from memory is paused, resumed and aborted before its first interval is
 * complete; a run fed from a FIFO nobody writes to is aborted (the producers poll with a timeout, host/audiosync.c
 * read_chunk); the feed setters refuse while a run is in progress.  usage: run_abort <dir for the fifo>; exit code 0 = ok. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <audiosync/audiosync.h>

static void *runner(void *arg)
{
    long lag = 0;
    *(int *)arg = audiosync_run("sanitizer", &lag);
    return NULL;
}

static int expect(int cond, const char *what)
{
    if (!cond) fprintf(stderr, "FAIL: %s\n", what);
    return cond ? 0 : 1;
}

int main(int argc, char **argv)
{
    int bad = 0, ret = 0;
    const size_t ns = 30 * SAMPLE_RATE;
    double *source = calloc(2 * ns, sizeof(double)), *sample = calloc(ns, sizeof(double));
    for (size_t i = 0; i < 2 * ns; i++) source[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
    for (size_t i = 0; i < ns; i++) sample[i] = 0.5 * source[i + 480];
    pthread_t th;

    /* 1: memory feed, 20 frames per millisecond: the first interval (144 000 frames) would need 7 s */
    bad += expect(audiosync_set_feed(source, 2 * ns, sample, ns, 20) == 0, "set_feed while idle");
    pthread_create(&th, NULL, runner, &ret);
    usleep(150 * 1000);
    bad += expect(audiosync_status() == RUNNING_ST, "running");
    bad += expect(audiosync_set_feed(source, 2 * ns, sample, ns, 0) == -1, "set_feed refused during a run");
    audiosync_pause();
    bad += expect(audiosync_status() == PAUSED_ST, "paused");
    usleep(50 * 1000);
    audiosync_resume();
    bad += expect(audiosync_status() == RUNNING_ST, "resumed");
    audiosync_abort();
    pthread_join(th, NULL);
    bad += expect(ret == -1 && audiosync_status() == IDLE_ST, "aborted run returns -1, idle");

    /* 2: the sample from a FIFO whose writer never starts */
    char fifo[512], file[512];
    snprintf(fifo, sizeof fifo, "%s/nobody.fifo", argc > 1 ? argv[1] : "/tmp");
    snprintf(file, sizeof file, "%s/source.f64le", argc > 1 ? argv[1] : "/tmp");
    unlink(fifo);
    bad += expect(mkfifo(fifo, 0600) == 0, "mkfifo");
    FILE *f = fopen(file, "wb");
    bad += expect(f && fwrite(source, sizeof(double), 48000, f) == 48000, "write source file");
    if (f) fclose(f);
    bad += expect(audiosync_set_feed_files(file, fifo) == 0, "set_feed_files while idle");
    pthread_create(&th, NULL, runner, &ret);
    usleep(300 * 1000);
    bad += expect(audiosync_status() == RUNNING_ST, "running on the stalled fifo");
    bad += expect(audiosync_set_feed_files(file, file) == -1, "set_feed_files refused during a run");
    audiosync_abort();
    pthread_join(th, NULL);
    bad += expect(ret == -1 && audiosync_status() == IDLE_ST, "aborted fifo run returns -1, idle");
    unlink(fifo); unlink(file);
    free(source); free(sample);
    printf("%d failures\n", bad);
    return bad ? 1 : 0;
}
