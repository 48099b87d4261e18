// csrc/host_narrow.cpp -- see host_narrow.h.  Plain C++ threads; no HIP.
#include "host_narrow.h"

#include <unistd.h>

#include <atomic>
#include <stdint.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {
constexpr size_t CHUNK = 128 * 1024; // doubles per chunk: 1 MB in, 0.5 MB out
constexpr size_t MIN_RUN = 4;        // chunks per ready() call, unless the buffer ends first

// exact <=> widening the narrowed value gives the double back (false for NaN, for values with more than 24 significant
// bits, and for magnitudes float32 cannot hold)
inline bool narrow_scalar(const double *s, float *d, size_t n)
{
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        const float f = (float)s[i];
        d[i] = f;
        bad |= ((double)f != s[i]);
    }
    return bad == 0;
}
#if defined(__x86_64__)
// eight doubles per step; the floats leave with streaming stores (the page-locked destination is only ever read by the DMA
// engine: no read-for-ownership of its lines).  Chosen at run time: the library itself is built for baseline x86-64.
__attribute__((target("avx2"))) bool narrow_avx2(const double *s, float *d, size_t n)
{
    size_t i = 0;
    while (i < n && ((uintptr_t)(d + i) & 31u)) { // up to the destination's 32-byte alignment
        const float f = (float)s[i];
        d[i] = f;
        if ((double)f != s[i]) return false;
        i++;
    }
    __m256d bad = _mm256_setzero_pd();
    for (; i + 8 <= n; i += 8) {
        const __m256d a = _mm256_loadu_pd(s + i), b = _mm256_loadu_pd(s + i + 4);
        const __m128 fa = _mm256_cvtpd_ps(a), fb = _mm256_cvtpd_ps(b);
        bad = _mm256_or_pd(bad, _mm256_cmp_pd(_mm256_cvtps_pd(fa), a, _CMP_NEQ_UQ)); // unordered (NaN) counts as different
        bad = _mm256_or_pd(bad, _mm256_cmp_pd(_mm256_cvtps_pd(fb), b, _CMP_NEQ_UQ));
        _mm256_stream_ps(d + i, _mm256_set_m128(fb, fa));
    }
    _mm_sfence();
    bool ok = _mm256_movemask_pd(bad) == 0;
    for (; i < n; i++) {
        const float f = (float)s[i];
        d[i] = f;
        ok = ok && ((double)f == s[i]);
    }
    return ok;
}
#endif
inline bool narrow_chunk(const double *s, float *d, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return narrow_avx2(s, d, n);
#endif
    return narrow_scalar(s, d, n);
}

struct Pool {
    std::mutex job_lock;              // one job at a time
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> workers;
    bool stop = false;
    unsigned long long gen = 0;       // bumped when a job is posted
    int active = 0;                   // workers still inside the current job
    // the current job
    const double *src = nullptr;
    float *dst = nullptr;
    size_t n = 0, nchunks = 0;
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::unique_ptr<std::atomic<int>[]> done; // per chunk: 0 pending, 1 exact, 2 inexact / skipped
    size_t done_cap = 0;

    Pool()
    {
        // 12 by default (measured on the MI355X host: 4 / 6 / 8 / 12 / 16 threads -> 0.66 / 0.61 / 0.60 / 0.56 / 0.58 ms per
        // cross_correlation(double*) at N = 1 440 000), fewer when the machine or the cgroup quota is smaller
        int nt = 12;
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc && (unsigned)nt > hc) nt = (int)hc;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long q = 0, per = 0;
            if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0 && q / per >= 1 && q / per < nt) nt = (int)(q / per);
            fclose(f);
        }
        if (const char *e = getenv("ASX_HOST_THREADS")) nt = atoi(e);
        if (nt < 1) nt = 1;
        if (nt > 64) nt = 64; // whatever the environment says
        owner = getpid();
        for (int i = 0; i < nt; i++) workers.emplace_back([this] { run(); });
    }
    // Threads do not survive fork(): in a child of the process that built the pool (Python multiprocessing with the fork start
    // method, for one) nobody would ever take a posted job.  There the caller converts every chunk itself (work_one below).
    pid_t owner = 0;
    bool alive() const { return getpid() == owner; }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv.notify_all();
        for (std::thread &t : workers) t.join();
    }
    // takes the next unclaimed chunk of the current job, if there is one
    bool work_one()
    {
        const size_t c = next.fetch_add(1);
        if (c >= nchunks) return false;
        int state = 2;
        if (!bad.load(std::memory_order_relaxed)) {
            const size_t off = c * CHUNK, len = off + CHUNK <= n ? CHUNK : n - off;
            if (narrow_chunk(src + off, dst + off, len)) state = 1;
            else bad.store(true);
        }
        done[c].store(state, std::memory_order_release);
        return true;
    }
    void work()
    {
        while (work_one()) {}
    }
    void run()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            work();
            {
                std::lock_guard<std::mutex> g(m);
                active--;
            }
            cv_done.notify_all();
        }
    }
};
Pool &pool()
{
    static Pool p; // joined at exit
    return p;
}
} // namespace

int asx_narrow_exact(const double *src, float *dst, size_t n, asx_narrow_ready_fn ready, void *user)
{
    if (n == 0) return 1;
    Pool &P = pool();
    std::lock_guard<std::mutex> job(P.job_lock);
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    if (nchunks > P.done_cap) { P.done.reset(new std::atomic<int>[nchunks]); P.done_cap = nchunks; }
    for (size_t c = 0; c < nchunks; c++) P.done[c].store(0, std::memory_order_relaxed);
    P.src = src; P.dst = dst; P.n = n; P.nchunks = nchunks;
    P.next.store(0); P.bad.store(false);
    const bool alive = P.alive();
    if (alive) {
        {
            std::lock_guard<std::mutex> g(P.m);
            P.active = (int)P.workers.size();
            P.gen++;
        }
        P.cv.notify_all();
    } else {
        P.active = 0; // a forked child: no worker threads here
    }
    // The caller hands finished chunks on, in order, while the workers convert the later ones.  While the chunk it needs is
    // still pending it converts chunks itself instead of spinning: under a one-CPU quota a spinning caller competes with the
    // very workers it waits for, and in a forked child it is the only thread there is.
    auto wait_chunk = [&](size_t i) {
        int st;
        while ((st = P.done[i].load(std::memory_order_acquire)) == 0)
            if (!P.work_one()) std::this_thread::yield();
        return st;
    };
    bool ok = true;
    size_t c = 0;
    while (c < nchunks && ok) {
        int st = wait_chunk(c);
        if (st != 1) { ok = false; break; }
        // at least MIN_RUN chunks per hand-over (every hipMemcpyAsync costs the caller microseconds), plus whatever else
        // is finished already
        size_t e = c + 1;
        while (e < nchunks && e - c < MIN_RUN) {
            st = wait_chunk(e);
            if (st != 1) break;
            e++;
        }
        while (e < nchunks && P.done[e].load(std::memory_order_acquire) == 1) e++;
        if (ready) {
            const size_t first = c * CHUNK, last = e * CHUNK < n ? e * CHUNK : n;
            ready(first, last - first, user);
        }
        c = e;
    }
    // every worker has left the job before the next one may be posted (and before src / dst may go away)
    if (alive) {
        std::unique_lock<std::mutex> g(P.m);
        P.cv_done.wait(g, [&] { return P.active == 0; });
    }
    return ok && !P.bad.load() ? 1 : 0;
}
