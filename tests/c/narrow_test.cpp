// tests/c/narrow_test.cpp -- csrc/host_narrow.cpp under ThreadSanitizer / AddressSanitizer: exact buffers convert and report
// every element through ready() exactly once and in order, an inexact value (or a NaN) anywhere makes the call return 0 and
// stops the reports before it, concurrent callers take turns.
#include "host_narrow.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)
struct Seen { size_t next = 0; bool ordered = true; };
static void on_ready(size_t first, size_t count, void *u) { Seen *s = (Seen *)u; if (first != s->next) s->ordered = false; s->next = first + count; }

static void one(size_t n, long bad_at, bool nan)
{
    std::vector<double> src(n);
    std::vector<float> dst(n, -1.f);
    for (size_t i = 0; i < n; i++) src[i] = (double)(float)(std::sin(0.001 * (double)i) * 1000.0);
    if (bad_at >= 0) src[(size_t)bad_at] = nan ? NAN : 0.1; // 0.1 is not a float32
    Seen seen;
    const int r = asx_narrow_exact(src.data(), dst.data(), n, on_ready, &seen);
    CHECK(seen.ordered);
    if (bad_at < 0) {
        CHECK(r == 1 && seen.next == n);
        for (size_t i = 0; i < n; i++) if ((double)dst[i] != src[i]) { CHECK(!"converted value differs"); break; }
    } else {
        CHECK(r == 0 && seen.next <= (size_t)bad_at);
    }
}

int main(int argc, char **)
{
    int cases = 0;
    for (size_t n : { (size_t)1, (size_t)1000, (size_t)131072, (size_t)131073, (size_t)1000003, (size_t)4320000 }) {
        one(n, -1, false); cases++;
        one(n, 0, false); cases++;
        one(n, (long)(n - 1), true); cases++;
        one(n, (long)(n / 2), false); cases++;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back([t] { for (int k = 0; k < 6; k++) one(300000 + 1000 * (size_t)t, k % 2 ? -1 : 12345 + t, false); });
    for (auto &x : th) x.join();
    cases += 24;
    // a forked child (the pool's threads do not exist there: ADVICE r4): the caller converts every chunk itself and returns.
    // (Not under ThreadSanitizer, which does not follow a fork of a threaded process: argv[1] = "nofork".)
    if (argc < 2) {
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) {
            alarm(60); // a hang is a failure, not a stuck test
            failures = 0;
            one(300000, -1, false);
            one(300000, 1234, false);
            _exit(failures ? 1 : 0);
        }
        int status = -1;
        CHECK(pid > 0 && waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
        cases += 2;
    }
    printf("%d cases, %d failures\n", cases, failures);
    return failures ? 1 : 0;
}
