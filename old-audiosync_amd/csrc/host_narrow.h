// csrc/host_narrow.h -- "is every double of this buffer exactly a float32?", answered WHILE converting it (host side, no HIP).
// The reference's buffers are double because ffmpeg is asked for f64le (src/capture/linux_capture.c:370,
// src/download/linux_download.c:41), but what it decodes is 16-bit or float audio: every frame is exactly representable
// in float32.  When that holds for a whole call, cross_correlation(double*) uploads 4 bytes per frame instead of 8 and runs
// the Pearson reduction on the float32 copy widened on the device -- bit-identical to using the doubles.
#pragma once
#include <stddef.h>

// Converts src[0..n) to dst[0..n) chunk by chunk on a small pool of worker threads (created on first use, kept for the life
// of the process; $ASX_HOST_THREADS, default 12 or the cgroup's CPU quota).  `ready(first, count, user)` is called on the CALLER's thread, in order, for
// every finished run of chunks (so uploads overlap the conversion of later chunks); it is never called for elements at or
// after an inexact value.  Returns 1 when every value was exactly a float32 (NaNs count as inexact: they keep the 8-byte
// path), 0 otherwise (dst is then partly written and of no use).  Thread-safe: concurrent calls take turns.
typedef void (*asx_narrow_ready_fn)(size_t first, size_t count, void *user);
int asx_narrow_exact(const double *src, float *dst, size_t n, asx_narrow_ready_fn ready, void *user);
