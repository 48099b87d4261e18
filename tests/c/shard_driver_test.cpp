// tests/c/shard_driver_test.cpp -- csrc/shard_driver.cpp (the control flow of asx_xcorr_batch_multi_dev) on host-memory
// stand-ins for the device operations, under ThreadSanitizer / AddressSanitizer.  Every "device" is a worker thread with a
// FIFO of closures (= a stream): run_shard enqueues a slow "kernel" that writes the shard's record, the gather enqueues
// the copies behind it, sync drains the FIFO.  Checks: block partition of uneven batches, record layout
// (int64 lag[width] | double coef[width] | int32 ret[width], entries past the count zero), every shard's gathered
// buffer complete, a failing shard (its error comes back, every thread is joined, every stream is drained BEFORE the
// call returns, the records survive), a failing record allocation (no stale width), width changes.
#include "shard_driver.h"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread t;
    Stream() : t([this] { loop(); }) {}
    ~Stream() { { std::lock_guard<std::mutex> g(m); stop = true; } cv.notify_all(); t.join(); }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front()); q.pop_front(); busy = true;
            }
            f();
            { std::lock_guard<std::mutex> g(m); busy = false; }
            cv.notify_all();
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> g(m); q.push_back(std::move(f)); } cv.notify_all(); }
    void drain() { std::unique_lock<std::mutex> g(m); cv.wait(g, [this] { return q.empty() && !busy; }); }
};

struct Ctx {
    int n;
    std::vector<Stream *> streams;
    int fail_shard = -1, fail_alloc = -1;
    std::atomic<int> allocs{0}, frees{0}, in_flight{0};
    std::vector<size_t> starts;
    // "kernel of shard r finished in round e": what the collective's ordering between ranks stands on
    std::mutex em;
    std::condition_variable ecv;
    std::vector<int> done_epoch;
    int epoch = 0;
    void mark(int r, int e) { { std::lock_guard<std::mutex> g(em); done_epoch[(size_t)r] = e; } ecv.notify_all(); }
    void wait_for(int r, int e) { std::unique_lock<std::mutex> g(em); ecv.wait(g, [&] { return done_epoch[(size_t)r] >= e; }); }
};

static int op_alloc(void *c, int i, size_t bytes, void **out, std::string *err)
{
    Ctx *x = (Ctx *)c;
    if (i == x->fail_alloc) { *out = nullptr; *err = "out of memory (injected)"; return -1; }
    *out = calloc(1, bytes ? bytes : 1);
    x->allocs++;
    return 0;
}
static void op_free(void *c, int, void *r) { free(r); ((Ctx *)c)->frees++; }
static int op_run(void *c, int i, void *record, size_t width, size_t count, const float *src, const float *, std::string *err)
{
    Ctx *x = (Ctx *)c;
    if (i == x->fail_shard) { *err = "kernel launch failed (injected)"; return -1; }
    const size_t start = x->starts[(size_t)i];
    const int e = x->epoch;
    x->in_flight++;
    x->streams[(size_t)i]->push([=] {
        std::this_thread::sleep_for(std::chrono::milliseconds(2 + 3 * (i % 3))); // shards finish out of order
        char *base = (char *)record;
        memset(base, 0, asx_shard_record_bytes(width));
        int64_t *lag = (int64_t *)base;
        double *coef = (double *)(base + width * sizeof(int64_t));
        int32_t *ret = (int32_t *)(base + width * (sizeof(int64_t) + sizeof(double)));
        for (size_t k = 0; k < count; k++) {
            const size_t id = start + k;
            lag[k] = 3 * (int64_t)id - 7 + (int64_t)src[k]; // src[k] = 0: the input is read while the "kernel" runs
            coef[k] = (double)id / 1024.0;
            ret[k] = -(int32_t)(id % 2);
        }
        x->in_flight--;
        x->mark(i, e);
    });
    return 0;
}
static int op_gather(void *c, int n, void *const *records, void *const *gathered, size_t rec, std::string *)
{
    Ctx *x = (Ctx *)c;
    // shard i's stream copies EVERY record into its gathered buffer, each source record behind its producer: a stand-in for
    // the collective's ordering (the real one waits for all ranks' kernels through the communicator)
    const int e = x->epoch;
    for (int i = 0; i < n; i++)
        x->streams[(size_t)i]->push([=] {
            for (int r = 0; r < n; r++) {
                x->wait_for(r, e);
                memcpy((char *)gathered[i] + (size_t)r * rec, records[r], rec);
            }
        });
    return 0;
}
static int op_sync(void *c, int i, std::string *) { ((Ctx *)c)->streams[(size_t)i]->drain(); return 0; }

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static void shard_range(size_t total, int n, int i, size_t *start, size_t *count)
{
    const size_t base = total / (size_t)n, extra = total % (size_t)n; // the rule of asx_shard_range / sharding.shard_range
    *count = base + ((size_t)i < extra ? 1 : 0);
    *start = (size_t)i * base + ((size_t)i < extra ? (size_t)i : extra);
}

static void one_case(int n, size_t total, int fail_shard, int fail_alloc)
{
    Ctx x; x.n = n; x.fail_shard = fail_shard; x.fail_alloc = fail_alloc;
    for (int i = 0; i < n; i++) x.streams.push_back(new Stream());
    x.done_epoch.assign((size_t)n, 0);
    AsxShardOps ops{ &x, op_alloc, op_free, op_run, op_gather, op_sync };
    AsxShardState st;
    for (int round = 0; round < 3; round++) {
        const size_t tot = total + (size_t)round * 5;       // the width changes between rounds
        const size_t width = (tot + (size_t)n - 1) / (size_t)n;
        std::vector<size_t> counts((size_t)n);
        x.starts.assign((size_t)n, 0);
        std::vector<std::vector<float>> src((size_t)n);
        std::vector<const float *> ps((size_t)n), pm((size_t)n);
        std::vector<void *> gathered((size_t)n);
        const size_t rec = asx_shard_record_bytes(width);
        for (int i = 0; i < n; i++) {
            shard_range(tot, n, i, &x.starts[(size_t)i], &counts[(size_t)i]);
            src[(size_t)i].assign(counts[(size_t)i] + 1, 0.f);
            ps[(size_t)i] = pm[(size_t)i] = src[(size_t)i].data();
            gathered[(size_t)i] = malloc((size_t)n * rec);
            memset(gathered[(size_t)i], 0xEE, (size_t)n * rec);
        }
        std::string err;
        x.epoch++;
        const int rc = asx_shard_drive(ops, st, n, ps.data(), pm.data(), counts.data(), width, gathered.data(), &err);
        CHECK(x.in_flight.load() == 0); // nothing runs behind the call's back, error or not
        if (x.fail_alloc >= 0) {
            CHECK(rc != 0 && err.find("result record") != std::string::npos && st.width == 0);
        } else if (x.fail_shard >= 0) {
            CHECK(rc != 0 && err.find("injected") != std::string::npos);
            char want[32]; snprintf(want, sizeof want, "shard %d", x.fail_shard);
            CHECK(err.find(want) != std::string::npos);
            CHECK(st.width == width); // the records are fine: the next call may use them
        } else {
            CHECK(rc == 0);
            for (int i = 0; i < n && rc == 0; i++)
                for (int r = 0; r < n; r++) {
                    const char *base = (const char *)gathered[(size_t)i] + (size_t)r * rec;
                    const int64_t *lag = (const int64_t *)base;
                    const double *coef = (const double *)(base + width * sizeof(int64_t));
                    const int32_t *ret = (const int32_t *)(base + width * (sizeof(int64_t) + sizeof(double)));
                    size_t s_r, c_r; shard_range(tot, n, r, &s_r, &c_r);
                    for (size_t k = 0; k < width; k++) {
                        const size_t id = s_r + k;
                        if (k < c_r) CHECK(lag[k] == 3 * (int64_t)id - 7 && coef[k] == (double)id / 1024.0 && ret[k] == -(int32_t)(id % 2));
                        else CHECK(lag[k] == 0 && coef[k] == 0.0 && ret[k] == 0);
                    }
                }
        }
        // the caller frees its buffers right after the call, as a caller that got an error would
        for (int i = 0; i < n; i++) free(gathered[(size_t)i]);
        if (round == 1) x.fail_shard = -1, x.fail_alloc = -1; // the third round succeeds on whatever the failures left
        if (round == 2 && (fail_shard >= 0 || fail_alloc >= 0)) CHECK(rc == 0);
    }
    asx_shard_release(ops, st);
    CHECK(x.allocs.load() == x.frees.load());
    for (Stream *s : x.streams) delete s;
}

int main()
{
    int cases = 0;
    for (int n : { 1, 2, 3, 8 })
        for (size_t total : { (size_t)n, (size_t)(5 * n + 1), (size_t)(7 * n + n - 1), (size_t)3 }) {
            if (total < 1) continue;
            one_case(n, total, -1, -1); cases++;
        }
    for (int n : { 2, 3, 8 }) {
        one_case(n, 5 * (size_t)n + 2, n - 1, -1); cases++; // the last shard fails
        one_case(n, 5 * (size_t)n + 2, 0, -1); cases++;     // the first shard fails while the others run
        one_case(n, 5 * (size_t)n + 2, -1, n / 2); cases++; // a record allocation fails half-way
    }
    printf("%d cases, %d failures\n", cases, failures);
    return failures ? 1 : 0;
}
