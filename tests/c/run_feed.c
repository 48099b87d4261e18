/* tests/c/run_feed.c — audiosync_run() with in-memory producers: the
 * growing-window loop of src/audiosync.c:226-259 must stop at the first interval whose
 * coefficient reaches MIN_CONFIDENCE and report the planted delay in milliseconds.
 * usage: run_feed <delay_frames> <noise_amplitude> ; prints "ret lag_ms status" */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <audiosync/audiosync.h>

static unsigned long long state = 88172645463325252ull;
static double noise(void)
{
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (double)(state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

int main(int argc, char **argv)
{
    const long delay = argc > 1 ? atol(argv[1]) : 12345;
    const double amp = argc > 2 ? atof(argv[2]) : 0.01;
    const size_t ns = 30 * SAMPLE_RATE, nsrc = 2 * ns;
    double *source = malloc(sizeof(double) * nsrc), *sample = malloc(sizeof(double) * ns);
    for (size_t i = 0; i < nsrc; i++) source[i] = noise();
    for (size_t i = 0; i < ns; i++) {
        const long j = (long)i + delay;
        sample[i] = (j >= 0 && (size_t)j < nsrc ? 0.5 * source[j] : 0.0) + amp * noise();
    }
    audiosync_set_debug(argc > 3);
    audiosync_set_feed(source, nsrc, sample, ns, 0);
    long lag_ms = -1;
    const int ret = audiosync_run("in-memory", &lag_ms);
    printf("%d %ld %s\n", ret, lag_ms, status_to_string(audiosync_status()));
    free(source); free(sample);
    return 0;
}
