// csrc/asx_api.hip — host side of the HIP layer: plans, workspaces, the C-ABI
// declared in include/audiosync/xcorr_hip.h.  No CPU fallback anywhere: when
// HIP is not usable every entry point fails and says so.
#include "audiosync/xcorr_hip.h"

#include "asx_internal.h"
#include "plan_math.h"
#include "shard_driver.h"
#include "host_narrow.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    // same channel and prefix as the reference's perror("audiosync: ...") calls
    fprintf(stderr, "audiosync: %s\n", buf);
    return -1;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define HIP_TRY_NULL(expr)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return nullptr;                                                                   \
        }                                                                                     \
    } while (0)

extern "C" const char *asx_last_error(void) { return g_err.c_str(); }
extern "C" int asx_abi_version(void) { return 2; }

extern "C" int asx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct DevGuard {
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

extern "C" int asx_current_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct asx_plan {
    int device = 0;
    AsxHostPlan host;
    AsxDev dev{};
    size_t group = 1;          // pairs per launch group
    size_t stamp_blocks = 0;   // diagnostic stamp buffer: blocks x 8 slots
    size_t ws_bytes = 0;
    hipStream_t stream = nullptr;
    std::mutex lock;

    // device tables
    std::vector<void *> allocs; // everything to hipFree on destroy
    // "Lanes": each owns the workspaces of one launch group and a stream.  With two lanes
    // (ASX_LANES=2) consecutive groups of a batch alternate lanes so that kernels of different
    // groups may overlap; measured gain 0..4 %, less than simply doubling the group, so one
    // lane is the default.
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        float2 *zxa = nullptr, *zya = nullptr, *ga = nullptr;
        AsxPeakWs pk{};
        AsxSeg *seg = nullptr;
        double *psums = nullptr;
        AsxSpecWs spec{};      // spectral Pearson (pearson_spectral.hip): work list, window sums, mode counters
    } lanes[2];
    int nlanes = 1;   // ASX_LANES=2 enables the second lane (measured: +0..4 %)
    hipEvent_t fork = nullptr;
    // Second look for pairs whose near-tie list overflowed (lazy; see second_look): lists that hold
    // every lag of ONE pair
    struct BigPeak {
        AsxCand *cand = nullptr;
        uint32_t *refine_idx = nullptr, *cand_n = nullptr, *refine_n = nullptr;
        double *refine_val = nullptr;
        unsigned long long *overflows = nullptr;
        float *src_dc = nullptr;     // the pair's source minus its mean (float32, 2N)
        double *stats = nullptr;     // {mean of source, sum of sample, their product = the shift of r}
        size_t cap = 0;
    } big;
    // pairs whose near-tie list overflowed since the list was last emptied (k_finalize appends, resolve_overflows reads)
    uint32_t *over_list = nullptr, *over_n = nullptr;
    volatile uint32_t *h_over_n = nullptr; // page-locked host mirror of *over_n (device-visible: AsxPeakWs::over_host)
    size_t over_cap = 0;
    std::vector<uint32_t> h_over;
    unsigned long long repaired = 0;   // pairs that took the second look
    bool exact = true;                 // asx_plan_set_exact: every entry point takes the second look (default)
    bool spectral = false;             // float32 groups take the spectral Pearson form (asx_plan_set_pearson; real-column plans)
    unsigned long long *mode_count = nullptr; // [ASX_PM_NMODES], cumulative over the plan's life
    // "measure" plans only: at the first device-resident batch the forward column kernel is timed against the caller's buffers
    // on TWO allocations of its output workspaces and the faster set is kept (tune_placement)
    bool tune_placement = false, placement_done = false;
    double placement_ms[2] = { 0.0, 0.0 };
    int placement_kept = -1;
    // staging for the host-pointer entry points (lazy)
    float *st_src = nullptr, *st_smp = nullptr;
    int64_t *st_lag = nullptr;
    double *st_coef = nullptr;
    int32_t *st_ret = nullptr;
    double *st_src64 = nullptr, *st_smp64 = nullptr;
    float *pin32 = nullptr;            // page-locked staging of the double ABI's narrowed frames (3N floats, lazy)
    unsigned long long narrowed = 0;   // asx_xcorr_f64 calls whose frames crossed PCIe as float32

    // Profiling: HIP events around every kernel family, on the stream the kernels run on.  A ring of
    // the last `prof_depth` batch calls is kept so that consecutive steps can be timed without a
    // host synchronisation between them (6 events per launch group, every group of a call).
    bool profiling = false;
    size_t prof_depth = 1, prof_calls = 0, prof_ring = 0;
    std::vector<std::vector<hipEvent_t>> evr; // ring slot -> 6 events per launch group of that call, grown as needed
    std::vector<size_t> prof_groups; // groups recorded by the call in ring slot i
    size_t ev_groups = 0;            // groups recorded by the call in progress / the latest call
};

template <typename T> static int dev_alloc(asx_plan *p, T **out, size_t count)
{
    void *ptr = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(&ptr, bytes));
    p->allocs.push_back(ptr);
    p->ws_bytes += bytes;
    *out = static_cast<T *>(ptr);
    return 0;
}

template <typename T> static int dev_upload(asx_plan *p, const T **out, const std::vector<T> &v)
{
    T *d = nullptr;
    if (dev_alloc(p, &d, v.size()) != 0) return -1;
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return 0;
}

static int plan_init(asx_plan *p, size_t N, size_t max_batch, const char *split)
{
    std::string err = asx_host_plan_build(N, split, &p->host);
    if (!err.empty()) return fail("plan for sample_len=%zu: %s", N, err.c_str());
    const AsxHostPlan &h = p->host;
    AsxDev &d = p->dev;
    d.N = (uint32_t)N;
    d.F = h.F; d.M = h.M; d.M1 = h.M1; d.M2 = h.M2; d.T = h.T; d.logT = h.logT; d.ntiles = h.ntiles;
    d.src_valid = h.src_valid;
    d.src_period = (uint32_t)(2 * N);
    d.nout = (uint32_t)(2 * N);
    // the device's r is the unnormalised inverse transform: F times the plain sum of products
    d.bound_scale = 2.0f * ASX_BOUND_C * 5.9604645e-8f * log2f((float)h.F) * (float)h.F;
    d.st1 = h.st1; d.st2 = h.st2;
    d.threads_cols = asx_pick_threads(h.st1, h.T / 2, 64, asx_lds_bytes_cols(d));
    d.threads_rows = asx_pick_threads(h.st2, 2, (h.M2 + ASX_ROW_STEPS - 1) / ASX_ROW_STEPS, asx_lds_bytes_rows(d));
    if (const char *e = getenv("ASX_THREADS_COLS")) d.threads_cols = atoi(e);
    if (const char *e = getenv("ASX_THREADS_ROWS")) d.threads_rows = atoi(e);
    if (dev_upload(p, &d.tw1, h.tw1) || dev_upload(p, &d.tw2, h.tw2) || dev_upload(p, &d.tw2s, h.tw2s) || dev_upload(p, &d.tw_lo, h.tw_lo) ||
        dev_upload(p, &d.tw_hi, h.tw_hi) ||
        dev_upload(p, &d.k1_of_pos1, h.k1_of_pos1) || dev_upload(p, &d.pos1_of_k1, h.pos1_of_k1) ||
        dev_upload(p, &d.pos2_of_k2, h.pos2_of_k2) || dev_upload(p, &d.row_tasks, h.row_tasks))
        return -1;

    // Which decomposition: the real-column kernels (rlayout.hip) when the plan allows them and they are compiled in for its
    // schedules (the reference's six lengths); ASX_LAYOUT=packed forces the packed-sample kernels (A/B runs, the run-time-schedule
    // kernels of ASX_GENERIC)
    d.rlayout = 0;
    d.band_rows = d.nbands = 0;
    d.col_pairs = nullptr;
    d.col_tw = nullptr;
    {
        const char *lay = getenv("ASX_LAYOUT");
        const bool packed = (lay && !strcmp(lay, "packed")) || getenv("ASX_GENERIC");
        if (h.rlayout && !packed) {
            if (dev_upload(p, &d.col_pairs, h.col_pairs) || dev_upload(p, &d.col_tw, h.col_tw)) return -1;
            d.rlayout = asx_rlayout_available(d) ? 1 : 0;
            d.band_rows = d.rlayout ? asx_rlayout_band_rows(d) : 0;
            d.nbands = d.band_rows ? 2 * h.M1 / d.band_rows : 0;
        }
    }
    // group size: keep the three inter-kernel intermediates (24*M bytes per pair) of one
    // group around the size of the 256 MiB Infinity Cache so the next kernel re-reads them on die
    size_t ws_mb = 4096;
    if (const char *e = getenv("ASX_WS_MB")) ws_mb = (size_t)atol(e) > 0 ? (size_t)atol(e) : ws_mb;
    size_t per_pair = (size_t)3 * h.M * sizeof(float2);
    size_t g = (ws_mb << 20) / per_pair / (getenv("ASX_LANES") && atoi(getenv("ASX_LANES")) == 2 ? 2 : 1);
    if (g < 1) g = 1;
    if (g > 65535) g = 65535; // grid.y / grid.z limit
    if (max_batch < 1) max_batch = 1;
    if (g > max_batch) g = max_batch;
    p->group = g;
    if (const char *e = getenv("ASX_LANES")) p->nlanes = atoi(e) == 2 ? 2 : 1;
    // candidate capacity per pair of the peak refinement (asx_internal.h): all 2N lags for short
    // tracks, else N/64 clamped to [ASX_CAND_MIN, ASX_CAND_MAX]
    size_t cap = std::min<size_t>(std::max<size_t>(N / 64, ASX_CAND_MIN), ASX_CAND_MAX);
    cap = std::min<size_t>(cap, 2 * N);
    for (int l = 0; l < p->nlanes; l++) {
        asx_plan::Lane &ln = p->lanes[l];
        ln.pk.cap = (uint32_t)cap;
        // (M1 + 1) rows per pair and spectrum: the real-column kernels (rlayout.hip) keep rows k1 = 0 .. M1
        const size_t mz = ((size_t)h.M1 + 1) * (size_t)h.M2;
        if (dev_alloc(p, &ln.zxa, g * mz) || dev_alloc(p, &ln.zya, g * mz) || dev_alloc(p, &ln.ga, g * mz) ||
            dev_alloc(p, &ln.pk.nrm_part, g * 2 * (size_t)h.ntiles) || dev_alloc(p, &ln.pk.bound2, g) ||
            dev_alloc(p, &ln.pk.pairmax, g) || dev_alloc(p, &ln.pk.cand_n, g) ||
            dev_alloc(p, &ln.pk.cand, g * cap) || dev_alloc(p, &ln.pk.refine_n, g) ||
            dev_alloc(p, &ln.pk.refine_idx, g * cap) || dev_alloc(p, &ln.pk.refine_val, g * cap) ||
            dev_alloc(p, &ln.pk.overflows, 1) ||
            dev_alloc(p, &ln.seg, g) || dev_alloc(p, &ln.psums, g * (size_t)asx_pearson_blocks((uint32_t)N) * 6))
            return -1;
        HIP_TRY(hipMemset(ln.pk.overflows, 0, sizeof(unsigned long long)));
        ln.pk.band = nullptr; ln.pk.tile_peak = nullptr;
        if (d.rlayout) {
            // spectral Pearson: band sums of both tracks (the sample fills the first half of its bands), one signed peak value
            // per inverse tile, the work list and the window sums of the group's pairs
            const size_t nbands = (size_t)d.nbands;
            if (!p->mode_count) {
                if (dev_alloc(p, &p->mode_count, ASX_PM_NMODES)) return -1;
                HIP_TRY(hipMemset(p->mode_count, 0, ASX_PM_NMODES * sizeof(unsigned long long)));
            }
            if (dev_alloc(p, &ln.pk.band, g * 2 * nbands * (size_t)h.ntiles) || dev_alloc(p, &ln.pk.tile_peak, g * (size_t)(h.M2 / h.T)) ||
                dev_alloc(p, &ln.spec.part, g * ASX_PREP_BLOCKS_MAX * 4) || dev_alloc(p, &ln.spec.hdr, g * ASX_SPEC_HDR))
                return -1;
            ln.spec.mode_count = p->mode_count;
            ln.spec.tol = 1e-5;
        }
        HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    }
    // the list of overflowed pairs is one per plan (the lanes append with atomics); a "window" of a device-resident
    // batch between two looks at it never holds more pairs than this
    p->over_cap = std::max<size_t>(g * (size_t)p->nlanes, 1024);
    if (dev_alloc(p, &p->over_list, p->over_cap) || dev_alloc(p, &p->over_n, 1)) return -1;
    HIP_TRY(hipMemset(p->over_n, 0, sizeof(uint32_t)));
    uint32_t *d_over_host = nullptr;
    {
        void *h = nullptr, *dptr = nullptr;
        HIP_TRY(hipHostMalloc(&h, sizeof(uint32_t), hipHostMallocMapped));
        p->h_over_n = (volatile uint32_t *)h;
        *p->h_over_n = 0;
        HIP_TRY(hipHostGetDevicePointer(&dptr, h, 0));
        d_over_host = (uint32_t *)dptr;
    }
    for (int l = 0; l < p->nlanes; l++) {
        p->lanes[l].pk.over_list = p->over_list;
        p->lanes[l].pk.over_n = p->over_n;
        p->lanes[l].pk.over_host = d_over_host;
        p->lanes[l].pk.over_cap = (uint32_t)p->over_cap;
    }
    p->spectral = d.rlayout != 0;
    if (const char *e = getenv("ASX_PEARSON")) p->spectral = p->spectral && strcmp(e, "direct") != 0; // A/B of the two forms
    if (const char *e = getenv("ASX_EXACT")) p->exact = atoi(e) != 0; // initial value of asx_plan_set_exact (A/B of its cost)
    HIP_TRY(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
    d.stamps = nullptr;
    d.stamp_kernel = 0;
    if (const char *e = getenv("ASX_STAMPS")) {
        unsigned long long *st = nullptr;
        p->stamp_blocks = g * (size_t)std::max(h.M1 + 1, 2 * h.ntiles);
        if (dev_alloc(p, &st, p->stamp_blocks * 8)) return -1;
        HIP_TRY(hipMemset(st, 0, p->stamp_blocks * 8 * sizeof(unsigned long long)));
        d.stamps = st;
        d.stamp_kernel = !strcmp(e, "fwd") ? 1 : !strcmp(e, "inv") ? 2 : 0;
    }
    {
        AsxDev *dcopy = nullptr;
        if (dev_alloc(p, &dcopy, 1)) return -1;
        d.self_dev = dcopy;
        HIP_TRY(hipMemcpy(dcopy, &d, sizeof(AsxDev), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipStreamCreate(&p->stream));
    // the tables and the memsets above went through the null stream; the lanes' streams are non-blocking
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

// Measured mode (what FFTW_MEASURE is to FFTW_ESTIMATE): the planner's cost model is only a model --
// for lengths outside its tuned table its pick was up to 27 % slower than the best split -- so time
// its cheapest candidates (plus its own pick) on the device, on synthetic pairs, and return the
// fastest.  About 20 plans are built and run; a fraction of a second per sample length.
static std::string measure_best_split(size_t N, size_t max_batch, int device)
{
    std::vector<std::string> cands = asx_host_plan_candidates(N, 16);
    {
        AsxHostPlan h;
        if (asx_host_plan_build(N, "", &h).empty()) {
            char buf[64];
            snprintf(buf, sizeof buf, "%dx%dx%d", h.M1, h.M2, h.T);
            if (std::find(cands.begin(), cands.end(), std::string(buf)) == cands.end()) cands.insert(cands.begin(), buf);
        }
    }
    if (cands.empty()) return "";
    size_t probe = ((size_t)512 << 20) / (12 * N); // at most 512 MB of float32 inputs
    probe = std::max<size_t>(probe, 4);
    probe = std::min<size_t>(probe, 1024);
    probe = std::max<size_t>(1, std::min(probe, max_batch));
    float *d_src = nullptr, *d_smp = nullptr;
    int64_t *d_lag = nullptr;
    double *d_coef = nullptr;
    int32_t *d_ret = nullptr;
    std::string best;
    float best_ms = 0.f;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc(&d_src, probe * 2 * N * sizeof(float)) == hipSuccess && hipMalloc(&d_smp, probe * N * sizeof(float)) == hipSuccess &&
        hipMalloc(&d_lag, probe * sizeof(int64_t)) == hipSuccess && hipMalloc(&d_coef, probe * sizeof(double)) == hipSuccess &&
        hipMalloc(&d_ret, probe * sizeof(int32_t)) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess &&
        asx_synth_pairs_dev(12345, 0, probe, N, 1, d_src, d_smp, nullptr, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess) {
        for (const std::string &c : cands) {
            asx_plan *p = asx_plan_create_ex(N, probe, device, c.c_str());
            if (!p) continue;
            bool ok = asx_xcorr_batch_f32_dev(p, d_src, d_smp, probe, d_lag, d_coef, d_ret, nullptr) == 0; // warm-up
            float ms = 0.f;
            if (ok) {
                ok = hipEventRecord(e0, p->stream) == hipSuccess;
                for (int r = 0; r < 2 && ok; r++) ok = asx_xcorr_batch_f32_dev(p, d_src, d_smp, probe, d_lag, d_coef, d_ret, nullptr) == 0;
                ok = ok && hipEventRecord(e1, p->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                     hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            }
            asx_plan_destroy(p);
            if (ok && (best.empty() || ms < best_ms)) { best = c; best_ms = ms; }
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_src); (void)hipFree(d_smp); (void)hipFree(d_lag); (void)hipFree(d_coef); (void)hipFree(d_ret);
    (void)hipGetLastError();
    return best;
}

extern "C" asx_plan *asx_plan_create_ex(size_t sample_len, size_t max_batch, int device, const char *split)
{
    if (!split || !*split) split = getenv("ASX_SPLIT");
    if (split && !strcmp(split, "measure")) {
        int prev = 0, ndev0 = 0;
        if (hipGetDeviceCount(&ndev0) != hipSuccess || ndev0 == 0) {
            fail("no usable HIP device; this library has no CPU fallback");
            return nullptr;
        }
        HIP_TRY_NULL(hipGetDevice(&prev));
        const int dev = device < 0 ? prev : device;
        if (dev >= ndev0) { fail("device %d out of range (%d devices)", dev, ndev0); return nullptr; }
        HIP_TRY_NULL(hipSetDevice(dev));
        const std::string best = measure_best_split(sample_len, max_batch ? max_batch : 1, dev);
        (void)hipSetDevice(prev);
        // an empty result (nothing could be timed) falls back to the planner's own choice
        asx_plan *mp = asx_plan_create_ex(sample_len, max_batch, device, best.empty() ? "auto" : best.c_str());
        if (mp) mp->tune_placement = true;
        return mp;
    }
    if (split && !strcmp(split, "auto")) split = "";
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        fail("no usable HIP device (%s); this library has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0) HIP_TRY_NULL(hipGetDevice(&device));
    if (device >= ndev) {
        fail("device %d out of range (%d devices)", device, ndev);
        return nullptr;
    }
    int prev = 0;
    HIP_TRY_NULL(hipGetDevice(&prev));
    HIP_TRY_NULL(hipSetDevice(device));
    asx_plan *p = new asx_plan();
    p->device = device;
    int rc = plan_init(p, sample_len, max_batch, split);
    (void)hipSetDevice(prev);
    if (rc != 0) {
        asx_plan_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" asx_plan *asx_plan_create(size_t sample_len, size_t max_batch, int device)
{
    return asx_plan_create_ex(sample_len, max_batch, device, nullptr);
}

extern "C" void asx_plan_destroy(asx_plan *p)
{
    if (!p) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    if (p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
    for (auto &ln : p->lanes) {
        if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
        if (ln.done) (void)hipEventDestroy(ln.done);
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    for (auto &ring : p->evr)
        for (hipEvent_t ev : ring) (void)hipEventDestroy(ev);
    for (void *a : p->allocs) (void)hipFree(a);
    if (p->pin32) (void)hipHostFree(p->pin32);
    if (p->h_over_n) (void)hipHostFree((void *)p->h_over_n);
    (void)hipSetDevice(prev);
    delete p;
}

extern "C" size_t asx_plan_sample_len(const asx_plan *p) { return p ? p->host.N : 0; }
extern "C" size_t asx_plan_fft_len(const asx_plan *p) { return p ? p->host.F : 0; }
extern "C" size_t asx_plan_group(const asx_plan *p) { return p ? p->group : 0; }
extern "C" size_t asx_plan_workspace_bytes(const asx_plan *p) { return p ? p->ws_bytes : 0; }
// diagnostic: copy back the k_rows phase clocks of the last group (needs ASX_STAMPS env + -DASX_STAMPS build)
extern "C" long asx_plan_debug_stamps(asx_plan *p, unsigned long long *out, size_t cap)
{
    if (!p || !p->dev.stamps) return -1;
    size_t n = p->stamp_blocks * 8;
    if (n > cap) n = cap;
    if (hipMemcpy(out, p->dev.stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)n;
}

extern "C" int asx_plan_peak_overflows(asx_plan *p, uint64_t *count)
{
    if (!p || !count) return fail("asx_plan_peak_overflows: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    unsigned long long total = 0;
    for (int l = 0; l < p->nlanes; l++) {
        unsigned long long v = 0;
        HIP_TRY(hipStreamSynchronize(p->lanes[l].stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(&v, p->lanes[l].pk.overflows, sizeof(v), hipMemcpyDeviceToHost));
        total += v;
    }
    *count = total;
    return 0;
}

extern "C" int asx_plan_set_exact(asx_plan *p, int on)
{
    if (!p) return fail("asx_plan_set_exact: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    // whatever the asynchronous mode left on the list belongs to calls that have returned: start empty.  The whole
    // device is drained, not only the plan's own streams: an asynchronous batch may still be in flight on a stream the
    // CALLER supplied (asx_xcorr_batch_f32_dev's `stream`), and its k_finalize adds to the count this resets.
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(p->over_n, 0, sizeof(uint32_t)));
    *p->h_over_n = 0;
    p->exact = on != 0;
    return 0;
}

extern "C" int asx_plan_set_pearson(asx_plan *p, int spectral)
{
    if (!p) return fail("asx_plan_set_pearson: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    if (spectral && !p->dev.rlayout) return fail("asx_plan_set_pearson: the spectral form needs a real-column plan (the reference's six lengths)");
    p->spectral = spectral != 0;
    return 0;
}

extern "C" int asx_plan_pearson_modes(asx_plan *p, uint64_t counts[3])
{
    if (!p || !counts) return fail("asx_plan_pearson_modes: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    counts[0] = counts[1] = counts[2] = 0;
    if (!p->mode_count) return 0;
    // the whole device, as asx_plan_set_exact does: a batch on a caller-supplied stream (asx_xcorr_batch_f32_dev) is on none of the
    // plan's own streams, and its pairs would be missing from the counts
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[ASX_PM_NMODES];
    HIP_TRY(hipMemcpy(v, p->mode_count, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) counts[i] = v[i];
    return 0;
}

extern "C" int asx_plan_peak_repairs(asx_plan *p, uint64_t *count)
{
    if (!p || !count) return fail("asx_plan_peak_repairs: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    *count = p->repaired;
    return 0;
}

// diagnostic (not in the public header): the peak-search state of pair `pair` of the last group on lane 0
extern "C" int asx_plan_debug_peak(asx_plan *p, size_t pair, float *bound2, uint32_t *cand_n, uint32_t *refine_n,
                                   unsigned long long *pairmax, double *vals, uint32_t *idxs, size_t cap)
{
    if (!p || pair >= p->group) return -1;
    DevGuard dg(p->device);
    const AsxPeakWs &W = p->lanes[0].pk;
    (void)hipDeviceSynchronize();
    if (hipMemcpy(bound2, W.bound2 + pair, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(cand_n, W.cand_n + pair, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(refine_n, W.refine_n + pair, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(pairmax, W.pairmax + pair, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    size_t n = std::min<size_t>(cap, W.cap);
    if (vals && hipMemcpy(vals, W.refine_val + pair * (size_t)W.cap, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (idxs && hipMemcpy(idxs, W.refine_idx + pair * (size_t)W.cap, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

extern "C" size_t asx_plan_peak_capacity(const asx_plan *p) { return p ? p->lanes[0].pk.cap : 0; }

extern "C" int asx_plan_threads(const asx_plan *p, int *cols, int *rows)
{
    if (!p) return -1;
    if (cols) *cols = p->dev.threads_cols;
    if (rows) *rows = p->dev.threads_rows;
    return 0;
}

extern "C" int asx_plan_layout(const asx_plan *p) { return p ? p->dev.rlayout : -1; }

extern "C" int asx_plan_narrowed_calls(asx_plan *p, uint64_t *count)
{
    if (!p || !count) return fail("asx_plan_narrowed_calls: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    *count = p->narrowed;
    return 0;
}

extern "C" int asx_plan_split(const asx_plan *p, int *m1, int *m2, int *tile_cols)
{
    if (!p) return -1;
    if (m1) *m1 = p->host.M1;
    if (m2) *m2 = p->host.M2;
    if (tile_cols) *tile_cols = p->host.T;
    return 0;
}

// ---------------------------------------------------------------------------
// running groups
// ---------------------------------------------------------------------------
static void prof_begin_call(asx_plan *p)
{
    p->ev_groups = 0;
    if (!p->profiling) return;
    p->prof_ring = p->prof_calls % p->prof_depth;
    if (p->prof_groups.size() < p->prof_depth) p->prof_groups.resize(p->prof_depth, 0);
    if (p->evr.size() < p->prof_depth) p->evr.resize(p->prof_depth);
    p->prof_groups[p->prof_ring] = 0;
}
static void prof_end_call(asx_plan *p, size_t groups)
{
    if (!p->profiling) return;
    p->ev_groups = groups;
    p->prof_groups[p->prof_ring] = groups;
    p->prof_calls++;
}

// every launch group of a call is recorded: the ring slot's event list grows with the call
static int prof_mark(asx_plan *p, hipStream_t s, size_t slot)
{
    if (!p->profiling) return 0;
    std::vector<hipEvent_t> &ev = p->evr[p->prof_ring];
    while (ev.size() <= slot) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        ev.push_back(e);
    }
    HIP_TRY(hipEventRecord(ev[slot], s));
    return 0;
}

// one group: g <= plan->group pairs, inputs device resident.  TIn selects the Pearson input type.
template <typename TIn>
static int run_group(asx_plan *p, const float *d_src, const float *d_smp, const TIn *p_src,
                     const TIn *p_smp, size_t g, int64_t *d_lag, double *d_coef, int32_t *d_ret,
                     float *d_r, hipStream_t s, size_t group_index, int lane = 0, uint32_t pair_base = 0, bool listed = true,
                     bool float32_call = true)
{
    const AsxDev &P = p->dev;
    asx_plan::Lane &W = p->lanes[lane];
    // `listed`: the caller reads the plan's overflow list behind this group (resolve_overflows).  A caller that does not
    // (the asynchronous mode) must not leave entries on it either: their indices count from ITS buffers, and the next
    // synchronous call would take a second look at its own staging buffers with them.  Marked and counted
    // (ASX_SEG_INEXACT -> ret = 1, asx_plan_peak_overflows) they still are.
    AsxPeakWs fin = W.pk;
    if (!listed) { fin.over_list = nullptr; fin.over_n = nullptr; fin.over_host = nullptr; fin.over_cap = 0; }
    // The spectral Pearson form (pearson_spectral.hip): groups of float32 inputs of a float32 ENTRY POINT on a real-column plan.
    // The double ABI keeps the float64 reduction over the caller's values even when its frames crossed PCIe as float32.
    const bool spectral = p->spectral && sizeof(TIn) == sizeof(float) && float32_call && W.pk.band;
    AsxPeakWs tk = W.pk; // what the transform kernels see
    if (!spectral) { tk.band = nullptr; tk.tile_peak = nullptr; }
    const size_t e0 = group_index * 6;
    if (prof_mark(p, s, e0 + 0)) return -1;
    asx_launch_fwd_cols(P, d_src, d_smp, W.zxa, W.zya, tk, (int)g, s);
    if (prof_mark(p, s, e0 + 1)) return -1;
    float2 *q = W.ga;
    asx_launch_rows(P, W.zxa, W.zya, q, tk, (int)g, s);
    if (prof_mark(p, s, e0 + 2)) return -1;
    asx_launch_inv_cols(P, q, tk, d_r, (int)g, s);
    if (prof_mark(p, s, e0 + 3)) return -1;
    asx_launch_finalize(P, fin, W.seg, (int)g, s, pair_base);
    // Blocks per pair of the exact re-evaluation: a candidate is one whole block's work whatever the grid, so the count only
    // sets how many candidates of a pair are in flight.  Nearly every block of a batch finds no candidate and exits: with 1024
    // pairs, 128 blocks each were 131 072 empty blocks, 25 us of a 2 ms step.
    const int dot_blocks = (int)std::min<size_t>(ASX_DOT_BLOCKS, std::max<size_t>(8, 16384 / g));
    if (sizeof(TIn) == sizeof(float)) // (the spectral form's first kernel applies the rule to the exact values itself: one launch less)
        asx_launch_refine_f32(P, (const float *)p_src, (const float *)p_smp, W.pk, W.seg, (int)g, s, dot_blocks, !spectral);
    else
        asx_launch_refine_f64(P, (const double *)p_src, (const double *)p_smp, W.pk, W.seg, (int)g, s, dot_blocks);
    if (prof_mark(p, s, e0 + 4)) return -1;
    if (spectral)
        asx_launch_pearson_spectral_f32(P, (const float *)p_src, (const float *)p_smp, tk, W.spec, W.seg, W.psums, d_lag, d_coef,
                                        d_ret, (int)g, s);
    else if (sizeof(TIn) == sizeof(float))
        asx_launch_pearson_f32((const float *)p_src, (const float *)p_smp, 2 * (size_t)P.N, P.N, P.N,
                               W.seg, W.psums, d_lag, d_coef, d_ret, (int)g, s);
    else
        asx_launch_pearson_f64((const double *)p_src, (const double *)p_smp, 2 * (size_t)P.N, P.N, P.N,
                               W.seg, W.psums, d_lag, d_coef, d_ret, (int)g, s);
    if (prof_mark(p, s, e0 + 5)) return -1;
    HIP_TRY(hipGetLastError());
    return 0;
}

// Second look at the pairs whose near-tie list overflowed (more lags inside the float32 error window than the
// per-pair list holds: a signal periodic in a few frames at a production length has hundreds of thousands of exactly
// tied peaks; an offset of hundreds of deviations in BOTH tracks makes every lag a near-tie).  k_finalize put their
// indices on the plan's list; every entry point reads it -- ONE host synchronisation per call (per window of
// `over_cap` pairs in a long device-resident batch), the price of a lag that is the reference's float64 argmax by
// construction (src/cross_correlation.c:52-67 has no candidate limit) -- and, only when the list is not empty, runs
// each listed pair again:
//   (i)  its transforms on (source - mean of source): the float32 error bound scales with both norms, and
//        r[k] = r'[k] + mean * sum(sample) exactly, a constant that k_inv_cols adds back when it forms the keys
//        (AsxPeakWs::shift, peak_key_shifted): with the offset gone from the source's norm the window shrinks by the
//        ratio offset / deviation;
//   (ii) into lists that hold all 2N lags; every listed lag is re-evaluated exactly (candidates x N double-double
//        multiply-adds: about 0.2 s for 400 000 candidates at N = 1 440 000) and the Pearson pass is redone for the
//        lag that wins.
// It runs on `s` behind everything the call has launched (the lanes have been joined), in the workspace slot 0 of
// lane 0 and the plan's one set of big lists: nothing else is in flight on this plan (one stream at a time per plan).
// The pointers are the bases the list's indices count from: pair i of the list is f_src + i * 2N etc.
template <typename TIn>
static int second_look(asx_plan *p, size_t i, const float *f_smp, const TIn *p_src, const TIn *p_smp, int64_t *d_lag,
                       double *d_coef, int32_t *d_ret, hipStream_t s)
{
    const AsxDev &P = p->dev;
    asx_plan::Lane &W = p->lanes[0];
    const size_t N = p->host.N;
    asx_plan::BigPeak &B = p->big;
    if (!B.cand) {
        // into a local first: a failed allocation must not leave half a set behind for the next call
        // (what was allocated stays on the plan's list and is freed with the plan)
        asx_plan::BigPeak T;
        T.cap = 2 * N;
        if (dev_alloc(p, &T.cand, T.cap) || dev_alloc(p, &T.refine_idx, T.cap) || dev_alloc(p, &T.refine_val, T.cap) ||
            dev_alloc(p, &T.cand_n, 1) || dev_alloc(p, &T.refine_n, 1) || dev_alloc(p, &T.overflows, 1) ||
            dev_alloc(p, &T.src_dc, 2 * N) || dev_alloc(p, &T.stats, ASX_DC_STATS_DOUBLES))
            return -1;
        HIP_TRY(hipMemsetAsync(T.overflows, 0, sizeof(unsigned long long), s));
        B = T;
    }
    AsxPeakWs K = W.pk;                   // slot 0's norms, bound and running maximum (k_rows recomputes all three) ...
    K.cand_n = B.cand_n; K.cand = B.cand; // ... with lists for all 2N lags: this look cannot overflow
    K.refine_n = B.refine_n; K.refine_idx = B.refine_idx; K.refine_val = B.refine_val;
    K.overflows = B.overflows;
    K.over_list = nullptr; K.over_n = nullptr; K.over_host = nullptr; K.over_cap = 0;
    K.band = nullptr; K.tile_peak = nullptr; // the second look ends with the direct Pearson reduction
    K.cap = (uint32_t)B.cap;
    HIP_TRY(hipMemsetAsync(B.cand_n, 0, sizeof(uint32_t), s));
    // r = r' + stats[2]
    if (sizeof(TIn) == sizeof(float))
        asx_launch_dc_remove_f32((const float *)p_src + i * 2 * N, (const float *)p_smp + i * N, P.N, (double)P.F, B.stats, B.src_dc, s);
    else
        asx_launch_dc_remove_f64((const double *)p_src + i * 2 * N, (const double *)p_smp + i * N, P.N, (double)P.F, B.stats, B.src_dc, s);
    K.shift = B.stats + 2;
    asx_launch_fwd_cols(P, B.src_dc, f_smp + i * N, W.zxa, W.zya, K, 1, s);
    asx_launch_rows(P, W.zxa, W.zya, W.ga, K, 1, s);
    asx_launch_inv_cols(P, W.ga, K, nullptr, 1, s);
    asx_launch_finalize(P, K, W.seg, 1, s);
    if (sizeof(TIn) == sizeof(float))
        asx_launch_refine_f32(P, (const float *)p_src + i * 2 * N, (const float *)p_smp + i * N, K, W.seg, 1, s, 2048);
    else
        asx_launch_refine_f64(P, (const double *)p_src + i * 2 * N, (const double *)p_smp + i * N, K, W.seg, 1, s, 2048);
    if (sizeof(TIn) == sizeof(float))
        asx_launch_pearson_f32((const float *)p_src + i * 2 * N, (const float *)p_smp + i * N, 2 * N, N, P.N,
                               W.seg, W.psums, d_lag ? d_lag + i : nullptr, d_coef + i, d_ret ? d_ret + i : nullptr, 1, s);
    else
        asx_launch_pearson_f64((const double *)p_src + i * 2 * N, (const double *)p_smp + i * N, 2 * N, N, P.N,
                               W.seg, W.psums, d_lag ? d_lag + i : nullptr, d_coef + i, d_ret ? d_ret + i : nullptr, 1, s);
    HIP_TRY(hipGetLastError());
    p->repaired++;
    return 0;
}

// Returns -1 on error, else the number of pairs that took the second look.  Waits for everything the caller has enqueued
// on `s` (its result copies included): ONE synchronisation; the count is read from the page-locked mirror k_finalize adds to
// only when a pair overflows -- no device-to-host copy sits between the last kernel and the host.
template <typename TIn>
static int resolve_overflows(asx_plan *p, const float *f_smp, const TIn *p_src, const TIn *p_smp, int64_t *d_lag,
                             double *d_coef, int32_t *d_ret, hipStream_t s)
{
    HIP_TRY(hipStreamSynchronize(s));
    if (p->lanes[0].pk.cap >= 2 * p->host.N) return 0; // the ordinary list already holds every lag
    const uint32_t n = *p->h_over_n;
    if (n == 0) return 0;
    if (n > p->over_cap) {
        // cannot happen (a window is at most over_cap pairs); if it ever does, the list must not stay poisoned
        (void)hipMemsetAsync(p->over_n, 0, sizeof(uint32_t), s);
        (void)hipStreamSynchronize(s);
        *p->h_over_n = 0;
        return fail("internal: %u overflowed pairs in a window of %zu", n, p->over_cap);
    }
    p->h_over.resize(n);
    HIP_TRY(hipMemcpyAsync(p->h_over.data(), p->over_list, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(p->over_n, 0, sizeof(uint32_t), s));
    HIP_TRY(hipStreamSynchronize(s));
    *p->h_over_n = 0;
    for (uint32_t k = 0; k < n; k++)
        if (second_look<TIn>(p, p->h_over[k], f_smp, p_src, p_smp, d_lag, d_coef, d_ret, s)) return -1;
    return (int)n;
}

// The forward column kernel runs 2-5 % faster or slower with the PHYSICAL placement of the buffers it streams together (the
// caller's inputs read, C_x / C_y written): strictly alternating with every re-allocation of either side, whatever the
// allocator, and untouched by offsets inside an allocation (EXPERIMENTS.md, round 4 items 27-28).  Nothing the library lays
// out can steer that -- but a plan made with split = "measure" (FFTW_MEASURE's role) may spend a few milliseconds once: at its
// first device-resident batch it allocates C_x / C_y a second time, times k_fwd_cols against the caller's buffers on both sets
// and keeps the faster.  The default planning mode never does this.
static int tune_placement(asx_plan *p, const float *d_src, const float *d_smp, size_t g, hipStream_t s)
{
    // (It allocates, synchronises and frees: never inside a stream capture -- a capturing caller keeps the first set, and the next
    // un-captured batch of >= min(group, 8) pairs tunes.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) return 0;
    p->placement_done = true;
    const AsxDev &P = p->dev;
    asx_plan::Lane &W = p->lanes[0];
    const size_t mz = ((size_t)p->host.M1 + 1) * (size_t)p->host.M2 * p->group;
    // whatever leaves this function early gives back the second set and the events (a failing HIP call used to leak them)
    struct Scratch {
        float2 *alt[2] = { nullptr, nullptr };
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool keep_alt = false;
        ~Scratch()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (!keep_alt) for (float2 *a : alt) if (a) (void)hipFree(a);
        }
    } sc;
    float2 *(&alt)[2] = sc.alt;
    if (hipMalloc((void **)&alt[0], mz * sizeof(float2)) != hipSuccess || hipMalloc((void **)&alt[1], mz * sizeof(float2)) != hipSuccess) {
        (void)hipGetLastError();
        return 0; // no room for a second set: keep what there is
    }
    hipEvent_t &e0 = sc.e0, &e1 = sc.e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    AsxPeakWs tk = W.pk;
    tk.band = nullptr; tk.tile_peak = nullptr;
    float2 *set[2][2] = { { W.zxa, W.zya }, { alt[0], alt[1] } };
    for (int k = 0; k < 2; k++) {
        float best = 0.f;
        for (int r = 0; r < 4; r++) { // the first launch warms the set up; the fastest of the other three counts
            HIP_TRY(hipEventRecord(e0, s));
            asx_launch_fwd_cols(P, d_src, d_smp, set[k][0], set[k][1], tk, (int)g, s);
            HIP_TRY(hipEventRecord(e1, s));
            HIP_TRY(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 1 && (best == 0.f || ms < best)) best = ms;
        }
        p->placement_ms[k] = best;
    }
    const int keep = p->placement_ms[1] < 0.995 * p->placement_ms[0] ? 1 : 0;
    p->placement_kept = keep;
    sc.keep_alt = true; // from here on the loop below owns both sets
    for (int i = 0; i < 2; i++) {
        float2 *drop = set[1 - keep][i];
        auto it = std::find(p->allocs.begin(), p->allocs.end(), (void *)drop);
        if (it != p->allocs.end()) p->allocs.erase(it);
        (void)hipFree(drop);
        if (keep == 1) p->allocs.push_back(set[1][i]);
    }
    W.zxa = set[keep][0];
    W.zya = set[keep][1];
    return 0;
}

extern "C" int asx_plan_placement(asx_plan *p, double ms[2], int *kept)
{
    if (!p || !ms || !kept) return fail("asx_plan_placement: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    ms[0] = p->placement_ms[0]; ms[1] = p->placement_ms[1];
    *kept = p->placement_kept;
    return 0;
}

extern "C" int asx_xcorr_batch_f32_dev(asx_plan *p, const float *d_source, const float *d_sample,
                                       size_t batch, int64_t *d_lag, double *d_coef, int32_t *d_ret,
                                       void *stream)
{
    if (!p || !d_source || !d_sample || !d_coef) return fail("asx_xcorr_batch_f32_dev: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    const size_t N = p->host.N;
    if (p->tune_placement && !p->placement_done && batch >= std::min<size_t>(p->group, 8) && // (lane 0's workspaces; a second lane keeps its own)
        tune_placement(p, d_source, d_sample, std::min(batch, p->group), s))
        return -1;
    prof_begin_call(p);
    // chunking: groups of at most `group` pairs; with two lanes a batch is cut into at least two
    // chunks (when it is big enough to fill the chip twice) that alternate between the lanes
    const bool overlap = (p->nlanes == 2) && !p->profiling && batch >= 8;
    size_t chunk = p->group;
    if (overlap && batch < 2 * chunk) chunk = (batch + 1) / 2;
    // windows: the pairs between two looks at the list of overflowed pairs (asx_plan_set_exact, on by default); a
    // window is a whole number of chunks and at most `over_cap` pairs, i.e. the whole batch unless it is very long
    const size_t window = p->exact ? std::max<size_t>(chunk, p->over_cap / chunk * chunk) : batch;
    size_t gi = 0;
    for (size_t w0 = 0; w0 < batch; w0 += window) {
        const size_t wn = std::min(window, batch - w0);
        if (overlap) {
            HIP_TRY(hipEventRecord(p->fork, s));
            for (int l = 0; l < 2; l++) HIP_TRY(hipStreamWaitEvent(p->lanes[l].stream, p->fork, 0));
        }
        for (size_t done = w0; done < w0 + wn; done += chunk, gi++) {
            const size_t g = std::min(chunk, w0 + wn - done);
            const int lane = overlap ? (int)(gi & 1) : 0;
            hipStream_t ls = overlap ? p->lanes[lane].stream : s;
            if (run_group<float>(p, d_source + done * 2 * N, d_sample + done * N, d_source + done * 2 * N,
                                 d_sample + done * N, g, d_lag ? d_lag + done : nullptr, d_coef + done,
                                 d_ret ? d_ret + done : nullptr, nullptr, ls, gi, lane, (uint32_t)(done - w0), p->exact))
                return -1;
        }
        if (overlap) {
            for (int l = 0; l < 2; l++) {
                HIP_TRY(hipEventRecord(p->lanes[l].done, p->lanes[l].stream));
                HIP_TRY(hipStreamWaitEvent(s, p->lanes[l].done, 0));
            }
        }
        // the second look, behind the window's last group (one host synchronisation per window)
        if (p->exact &&
            resolve_overflows<float>(p, d_sample + w0 * N, d_source + w0 * 2 * N, d_sample + w0 * N,
                                     d_lag ? d_lag + w0 : nullptr, d_coef + w0, d_ret ? d_ret + w0 : nullptr, s) < 0)
            return -1;
    }
    prof_end_call(p, gi);
    return 0;
}

extern "C" int asx_xcorr_debug_r_dev(asx_plan *p, const float *d_source, const float *d_sample,
                                     float *d_r, int64_t *d_lag, double *d_coef, int32_t *d_ret,
                                     void *stream)
{
    if (!p || !d_source || !d_sample || !d_coef || !d_r) return fail("asx_xcorr_debug_r_dev: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    prof_begin_call(p);
    // like every other entry point: listed and looked at again in the exact mode, only marked (ret = 1) otherwise
    // (the direct Pearson form: this entry point exists to compare decompositions and to dump r)
    int rc = run_group<float>(p, d_source, d_sample, d_source, d_sample, 1, d_lag, d_coef, d_ret, d_r, s, 0, 0, 0, p->exact, false);
    prof_end_call(p, 1);
    if (rc == 0 && p->exact && resolve_overflows<float>(p, d_sample, d_source, d_sample, d_lag, d_coef, d_ret, s) < 0) rc = -1;
    return rc;
}

static int ensure_staging(asx_plan *p)
{
    if (p->st_src) return 0;
    const size_t N = p->host.N, g = p->group;
    if (dev_alloc(p, &p->st_src, g * 2 * N) || dev_alloc(p, &p->st_smp, g * N) ||
        dev_alloc(p, &p->st_lag, g) || dev_alloc(p, &p->st_coef, g) || dev_alloc(p, &p->st_ret, g))
        return -1;
    return 0;
}

extern "C" int asx_xcorr_batch_f32(asx_plan *p, const float *source, const float *sample, size_t batch,
                                   int64_t *lag, double *coef, int32_t *ret)
{
    if (!p || !source || !sample || !lag || !coef || !ret) return fail("asx_xcorr_batch_f32: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    if (ensure_staging(p)) return -1;
    const size_t N = p->host.N;
    hipStream_t s = p->stream;
    prof_begin_call(p);
    for (size_t done = 0; done < batch; done += p->group) {
        const size_t g = std::min(p->group, batch - done);
        HIP_TRY(hipMemcpyAsync(p->st_src, source + done * 2 * N, g * 2 * N * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(p->st_smp, sample + done * N, g * N * sizeof(float), hipMemcpyHostToDevice, s));
        if (run_group<float>(p, p->st_src, p->st_smp, p->st_src, p->st_smp, g, p->st_lag, p->st_coef,
                             p->st_ret, nullptr, s, 0))
            return -1;
        // the results come back with the same synchronisation that looks at the overflow list; again behind a second look
        for (int pass = 0; pass < 2; pass++) {
            HIP_TRY(hipMemcpyAsync(lag + done, p->st_lag, g * sizeof(int64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(coef + done, p->st_coef, g * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(ret + done, p->st_ret, g * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            if (pass == 1) { HIP_TRY(hipStreamSynchronize(s)); break; }
            const int looked = resolve_overflows<float>(p, p->st_smp, p->st_src, p->st_smp, p->st_lag, p->st_coef, p->st_ret, s);
            if (looked < 0) return -1;
            if (looked == 0) break;
        }
    }
    return 0;
}

// Several plans (normally one per GPU of the node) in ONE process: the batch is block-partitioned over
// them (the first batch % nplans plans get one pair more), every block runs on its plan's device from
// its own host thread, and the results land in the caller's arrays in pair order -- the "gather" of a
// single process is the D2H copy.  No data-path exchange between devices.
extern "C" int asx_xcorr_batch_multi(asx_plan *const *plans, int nplans, const float *source, const float *sample,
                                     size_t batch, int64_t *lag, double *coef, int32_t *ret)
{
    if (!plans || nplans < 1 || !source || !sample || !lag || !coef || !ret) return fail("asx_xcorr_batch_multi: bad argument");
    const size_t N = plans[0] ? plans[0]->host.N : 0;
    for (int i = 0; i < nplans; i++)
        if (!plans[i] || plans[i]->host.N != N) return fail("asx_xcorr_batch_multi: plans must share one sample_len");
    std::vector<int> rc((size_t)nplans, 0);
    std::vector<std::string> err((size_t)nplans);
    std::vector<std::thread> workers;
    const size_t base = batch / (size_t)nplans, extra = batch % (size_t)nplans;
    size_t start = 0;
    for (int i = 0; i < nplans; i++) {
        const size_t count = base + ((size_t)i < extra ? 1 : 0);
        if (count) {
            workers.emplace_back([=, &rc, &err]() {
                rc[(size_t)i] = asx_xcorr_batch_f32(plans[i], source + start * 2 * N, sample + start * N, count,
                                                    lag + start, coef + start, ret + start);
                if (rc[(size_t)i] != 0) err[(size_t)i] = asx_last_error(); // the worker thread's message
            });
        }
        start += count;
    }
    for (std::thread &t : workers) t.join();
    for (int i = 0; i < nplans; i++)
        if (rc[(size_t)i] != 0) return fail("asx_xcorr_batch_multi: plan %d: %s", i, err[(size_t)i].c_str());
    return 0;
}

// ---------------------------------------------------------------------------
// device-resident shards + RCCL result gather.  RCCL is dlopen()ed: seven entry points by name.
// ---------------------------------------------------------------------------
extern "C" int asx_shard_range(size_t total, int nshards, int shard, size_t *start, size_t *count)
{
    if (nshards < 1 || shard < 0 || shard >= nshards || !start || !count) return fail("asx_shard_range: bad argument");
    const size_t base = total / (size_t)nshards, extra = total % (size_t)nshards;
    *count = base + ((size_t)shard < extra ? 1 : 0);
    *start = (size_t)shard * base + std::min((size_t)shard, extra);
    return 0;
}

extern "C" size_t asx_result_bytes(size_t width) { return asx_shard_record_bytes(width); }

namespace {
typedef void *nccl_comm_t;
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int /* ncclDataType_t */, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};
std::mutex g_rccl_lock;
Rccl g_rccl;
const Rccl *rccl_load()
{
    std::lock_guard<std::mutex> g(g_rccl_lock);
    if (g_rccl.handle) return &g_rccl;
    void *h = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { fail("cannot load librccl.so.1: %s", dlerror()); return nullptr; }
    Rccl r;
    r.handle = h;
    r.CommInitAll = (int (*)(nccl_comm_t *, int, const int *))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (int (*)(nccl_comm_t))dlsym(h, "ncclCommDestroy");
    r.AllGather = (int (*)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
    r.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    r.GetVersion = (int (*)(int *))dlsym(h, "ncclGetVersion");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        fail("librccl.so.1 lacks an entry point this library binds");
        dlclose(h);
        return nullptr;
    }
    g_rccl = r;
    return &g_rccl;
}
} // namespace

struct asx_comm {
    std::vector<asx_plan *> plans;
    std::vector<int> devices;      // the plans' devices, kept here: destroy must not need the plans (they may be gone)
    std::vector<nccl_comm_t> comms;
    AsxShardState state;           // the shards' result records and the width they are sized for (shard_driver.h)
};

// the ops table of csrc/shard_driver.cpp bound to HIP + RCCL
namespace {
int shard_alloc(void *ctx, int i, size_t bytes, void **out, std::string *err)
{
    asx_comm *c = (asx_comm *)ctx;
    DevGuard dg(c->devices[(size_t)i]);
    if (!dg.ok) { *err = "cannot select the device"; return -1; }
    const hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) { *out = nullptr; *err = hipGetErrorString(e); return -1; }
    return 0;
}
void shard_free(void *ctx, int i, void *rec)
{
    asx_comm *c = (asx_comm *)ctx;
    DevGuard dg(c->devices[(size_t)i]);
    (void)hipFree(rec);
}
int shard_run(void *ctx, int i, void *record, size_t width, size_t count, const float *d_src, const float *d_smp, std::string *err)
{
    asx_comm *c = (asx_comm *)ctx;
    asx_plan *p = c->plans[(size_t)i];
    if (hipSetDevice(p->device) != hipSuccess) { *err = "hipSetDevice failed"; return -1; }
    char *base = (char *)record;
    int64_t *lag = (int64_t *)base;
    double *coef = (double *)(base + width * sizeof(int64_t));
    int32_t *ret = (int32_t *)(base + width * (sizeof(int64_t) + sizeof(double)));
    if (hipMemsetAsync(base, 0, asx_shard_record_bytes(width), p->stream) != hipSuccess) { *err = "hipMemsetAsync failed"; return -1; }
    if (count && asx_xcorr_batch_f32_dev(p, d_src, d_smp, count, lag, coef, ret, p->stream) != 0) { *err = asx_last_error(); return -1; }
    return 0;
}
int shard_gather(void *ctx, int n, void *const *records, void *const *gathered, size_t rec, std::string *err)
{
    asx_comm *c = (asx_comm *)ctx;
    const Rccl *R = rccl_load();
    if (!R) { *err = asx_last_error(); return -1; }
    int grc = R->GroupStart();
    for (int i = 0; i < n && grc == 0; i++) {
        asx_plan *p = c->plans[(size_t)i];
        DevGuard dg(p->device);
        grc = R->AllGather(records[i], gathered[i], rec, 0 /* ncclInt8 / ncclChar */, c->comms[(size_t)i], p->stream);
    }
    const int erc = R->GroupEnd();
    if (grc == 0) grc = erc;
    if (grc != 0) { *err = R->GetErrorString(grc); return -1; }
    return 0;
}
int shard_sync(void *ctx, int i, std::string *err)
{
    asx_comm *c = (asx_comm *)ctx;
    asx_plan *p = c->plans[(size_t)i];
    DevGuard dg(p->device);
    const hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) { *err = hipGetErrorString(e); return -1; }
    return 0;
}
AsxShardOps shard_ops(asx_comm *c) { return AsxShardOps{ c, shard_alloc, shard_free, shard_run, shard_gather, shard_sync }; }
} // namespace

extern "C" void asx_comm_destroy(asx_comm *c)
{
    if (!c) return;
    const Rccl *R = g_rccl.handle ? &g_rccl : nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < c->devices.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->comms.size() && c->comms[i] && R) (void)R->CommDestroy(c->comms[i]);
    }
    (void)hipSetDevice(prev);
    asx_shard_release(shard_ops(c), c->state); // frees by device id: the plans are not touched
    delete c;
}

extern "C" asx_comm *asx_comm_create(asx_plan *const *plans, int nplans)
{
    if (!plans || nplans < 1) { fail("asx_comm_create: bad argument"); return nullptr; }
    for (int i = 0; i < nplans; i++) {
        if (!plans[i] || plans[i]->host.N != plans[0]->host.N) { fail("asx_comm_create: plans must share one sample_len"); return nullptr; }
        for (int j = 0; j < i; j++)
            if (plans[j]->device == plans[i]->device) { fail("asx_comm_create: one plan per device (device %d twice)", plans[i]->device); return nullptr; }
    }
    const Rccl *R = rccl_load();
    if (!R) return nullptr;
    asx_comm *c = new asx_comm();
    c->plans.assign(plans, plans + nplans);
    c->comms.assign((size_t)nplans, nullptr);
    c->devices.resize((size_t)nplans);
    for (int i = 0; i < nplans; i++) c->devices[(size_t)i] = plans[i]->device;
    const int rc = R->CommInitAll(c->comms.data(), nplans, c->devices.data());
    if (rc != 0) {
        fail("ncclCommInitAll over %d device(s) failed: %s", nplans, R->GetErrorString(rc));
        c->comms.assign((size_t)nplans, nullptr);
        asx_comm_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" int asx_xcorr_batch_multi_dev(asx_comm *c, const float *const *d_source, const float *const *d_sample,
                                         const size_t *counts, size_t width, void *const *d_gathered)
{
    if (!c) return fail("asx_xcorr_batch_multi_dev: bad argument");
    if (!rccl_load()) return -1;
    std::string err;
    if (asx_shard_drive(shard_ops(c), c->state, (int)c->plans.size(), d_source, d_sample, counts, width, d_gathered, &err) != 0)
        return fail("asx_xcorr_batch_multi_dev: %s", err.c_str());
    return 0;
}

// Frames that are exactly float32 cross PCIe as float32 (csrc/host_narrow.h): the conversion runs on the host pool into
// page-locked staging, every finished run of chunks is uploaded while later ones are still being converted.
namespace {
struct NarrowUp {
    hipStream_t s;
    const float *pin;
    float *dev;
    hipError_t err;
};
void narrow_upload(size_t first, size_t count, void *user)
{
    NarrowUp *u = (NarrowUp *)user;
    if (u->err != hipSuccess) return;
    u->err = hipMemcpyAsync(u->dev + first, u->pin + first, count * sizeof(float), hipMemcpyHostToDevice, u->s);
}
bool narrow_enabled()
{
    static const bool on = !(getenv("ASX_NARROW") && atoi(getenv("ASX_NARROW")) == 0); // ASX_NARROW=0: always 8 bytes per frame (A/B)
    return on;
}
// 1: dev[0..n) holds the frames as float32 (enqueued on s); 0: not every frame is a float32 (nothing usable was left); -1: error
int upload_narrowed(hipStream_t s, const double *host, size_t n, float *pin, float *dev)
{
    NarrowUp u{ s, pin, dev, hipSuccess };
    const int exact = asx_narrow_exact(host, pin, n, narrow_upload, &u);
    if (u.err != hipSuccess) return fail("hipMemcpyAsync of narrowed frames failed: %s", hipGetErrorString(u.err));
    return exact;
}
} // namespace

extern "C" int asx_xcorr_f64(asx_plan *p, const double *source, const double *sample, long *lag,
                             double *coefficient)
{
    if (!p || !source || !sample || !lag || !coefficient) return fail("asx_xcorr_f64: null argument");
    std::lock_guard<std::mutex> guard(p->lock);
    DevGuard dg(p->device);
    if (!dg.ok) return fail("cannot select device %d", p->device);
    if (ensure_staging(p)) return -1;
    const size_t N = p->host.N;
    if (!p->st_src64) {
        if (dev_alloc(p, &p->st_src64, 2 * N) || dev_alloc(p, &p->st_smp64, N)) return -1;
    }
    hipStream_t s = p->stream;
    prof_begin_call(p);
    // What ffmpeg decodes from 16-bit or float audio is exactly representable in float32 (src/capture/linux_capture.c:370
    // asks for f64le all the same): then 4 bytes per frame cross PCIe, and the float64 passes (exact re-evaluation, Pearson)
    // read the float32 copy widened in registers -- the same values, the same operations, the same bits.
    int narrow = 0;
    if (narrow_enabled()) {
        if (!p->pin32) {
            void *pin = nullptr;
            if (hipHostMalloc(&pin, 3 * N * sizeof(float), hipHostMallocDefault) == hipSuccess) p->pin32 = (float *)pin;
            else (void)hipGetLastError(); // no staging: the 8-byte path below
        }
        if (p->pin32) {
            narrow = upload_narrowed(s, source, 2 * N, p->pin32, p->st_src);
            if (narrow == 1) narrow = upload_narrowed(s, sample, N, p->pin32 + 2 * N, p->st_smp);
            if (narrow < 0) return -1;
        }
    }
    if (narrow == 1) {
        p->narrowed++;
        if (run_group<float>(p, p->st_src, p->st_smp, p->st_src, p->st_smp, 1, p->st_lag, p->st_coef, p->st_ret, nullptr, s, 0, 0, 0,
                             true, false))
            return -1;
    } else {
    HIP_TRY(hipMemcpyAsync(p->st_src64, source, 2 * N * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(p->st_smp64, sample, N * sizeof(double), hipMemcpyHostToDevice, s));
    asx_launch_cvt_f64_f32(p->st_src64, p->st_src, 2 * N, s);
    asx_launch_cvt_f64_f32(p->st_smp64, p->st_smp, N, s);
    if (run_group<double>(p, p->st_src, p->st_smp, p->st_src64, p->st_smp64, 1, p->st_lag, p->st_coef,
                          p->st_ret, nullptr, s, 0))
        return -1;
    }
    int64_t h_lag = 0;
    double h_coef = 0;
    int32_t h_ret = -1;
    // the result comes back with the same synchronisation that looks at the overflow list; again behind a second look
    for (int pass = 0; pass < 2; pass++) {
        HIP_TRY(hipMemcpyAsync(&h_lag, p->st_lag, sizeof(h_lag), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&h_coef, p->st_coef, sizeof(h_coef), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&h_ret, p->st_ret, sizeof(h_ret), hipMemcpyDeviceToHost, s));
        if (pass == 1) { HIP_TRY(hipStreamSynchronize(s)); break; }
        const int looked = narrow == 1 ? resolve_overflows<float>(p, p->st_smp, p->st_src, p->st_smp, p->st_lag, p->st_coef, p->st_ret, s)
                                       : resolve_overflows<double>(p, p->st_smp, p->st_src64, p->st_smp64, p->st_lag, p->st_coef, p->st_ret, s);
        if (looked < 0) return -1;
        if (looked == 0) break;
    }
    *lag = (long)h_lag;
    *coefficient = h_coef;
    return h_ret;
}

// pearson_coefficient() is called on its own by the reference's tests and may be called often: its
// device buffers are kept per device and only ever grow (the reference allocates nothing here).
struct PearsonScratch {
    std::mutex lock;
    size_t cap = 0;              // doubles per array
    double *a = nullptr, *b = nullptr, *ps = nullptr, *c = nullptr;
    AsxSeg *seg = nullptr;
    hipStream_t stream = nullptr;
};
static std::mutex g_pearson_lock;
static std::vector<PearsonScratch *> g_pearson; // by device id, created on first use, kept for the life of the process

extern "C" int asx_pearson_f64(const double *a, const double *b, size_t n, int device, double *out)
{
    if (!a || !b || !out) return fail("asx_pearson_f64: null argument");
    if (n > 0xFFFFFFFFull) return fail("asx_pearson_f64: range too long");
    const int ndev = asx_device_count();
    if (ndev == 0) return fail("no usable HIP device; this library has no CPU fallback");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail("asx_pearson_f64: device %d out of range (%d devices)", device, ndev);
    DevGuard dg(device);
    if (!dg.ok) return fail("cannot select device %d", device);
    PearsonScratch *Sp = nullptr;
    {
        std::lock_guard<std::mutex> g(g_pearson_lock);
        if (g_pearson.size() <= (size_t)device) g_pearson.resize((size_t)device + 1, nullptr);
        if (!g_pearson[(size_t)device]) g_pearson[(size_t)device] = new PearsonScratch();
        Sp = g_pearson[(size_t)device];
    }
    PearsonScratch &S = *Sp;
    std::lock_guard<std::mutex> guard(S.lock);
    if (!S.stream) {
        // committed only when every piece exists: a failed allocation leaves nothing half-initialised behind
        hipStream_t st = nullptr;
        double *ps = nullptr, *c = nullptr;
        AsxSeg *seg = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&ps, ASX_PEARSON_BLOCKS_MAX * 6 * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&c, sizeof(double)) != hipSuccess || hipMalloc((void **)&seg, sizeof(AsxSeg)) != hipSuccess) {
            const hipError_t e = hipGetLastError();
            if (st) (void)hipStreamDestroy(st);
            (void)hipFree(ps); (void)hipFree(c); (void)hipFree(seg);
            return fail("asx_pearson_f64: scratch allocation failed: %s", hipGetErrorString(e));
        }
        S.ps = ps; S.c = c; S.seg = seg; S.stream = st;
    }
    if (n > S.cap) {
        (void)hipFree(S.a); (void)hipFree(S.b);
        S.a = S.b = nullptr; S.cap = 0;
        const size_t want = std::max<size_t>(n, 4096);
        double *na = nullptr, *nb = nullptr;
        if (hipMalloc((void **)&na, want * sizeof(double)) != hipSuccess || hipMalloc((void **)&nb, want * sizeof(double)) != hipSuccess) {
            const hipError_t e = hipGetLastError();
            (void)hipFree(na); (void)hipFree(nb);
            return fail("asx_pearson_f64: hipMalloc failed: %s", hipGetErrorString(e));
        }
        S.a = na; S.b = nb; S.cap = want;
    }
    AsxSeg seg{};
    seg.lag = 0; seg.src_off = 0; seg.smp_off = 0; seg.len = (uint32_t)n; seg.peak = 0;
    HIP_TRY(hipMemcpyAsync(S.a, a, n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.b, b, n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.seg, &seg, sizeof(seg), hipMemcpyHostToDevice, S.stream));
    asx_launch_pearson_f64(S.a, S.b, 0, 0, (uint32_t)n, S.seg, S.ps, nullptr, S.c, nullptr, 1, S.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, S.c, sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return 0;
}

// ---------------------------------------------------------------------------
// result consumers
// ---------------------------------------------------------------------------
extern "C" int asx_results_to_ms_dev(const int64_t *d_lag, const double *d_coef, const int32_t *d_ret, size_t batch,
                                     double min_confidence, double sample_rate, int64_t *d_lag_ms,
                                     int32_t *d_accept, void *stream)
{
    if (!d_lag || !d_coef || !d_ret || !d_lag_ms) return fail("asx_results_to_ms_dev: null argument");
    if (!(sample_rate > 0.0)) return fail("asx_results_to_ms_dev: bad sample rate");
    asx_launch_results_to_ms(d_lag, d_coef, d_ret, batch, min_confidence, sample_rate, d_lag_ms, d_accept,
                             (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// growing-window streams (BASELINE config 5)
// ---------------------------------------------------------------------------
struct asx_stream {
    int device = 0;
    size_t cap = 0;                 // max sample frames (source capacity = 2*cap)
    size_t n_src = 0, n_smp = 0;    // frames resident
    double *src64 = nullptr, *smp64 = nullptr;
    float *src32 = nullptr, *smp32 = nullptr;
    int64_t *d_lag = nullptr;
    double *d_coef = nullptr;
    int32_t *d_ret = nullptr;
    std::vector<asx_plan *> plans;  // one per prefix length seen
    hipStream_t s = nullptr;        // uploads and conversions of new frames
    std::mutex lock;
};

extern "C" void asx_stream_destroy(asx_stream *st)
{
    if (!st) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(st->device);
    for (asx_plan *p : st->plans) asx_plan_destroy(p);
    if (st->s) { (void)hipStreamSynchronize(st->s); (void)hipStreamDestroy(st->s); }
    (void)hipFree(st->src64); (void)hipFree(st->smp64); (void)hipFree(st->src32); (void)hipFree(st->smp32);
    (void)hipFree(st->d_lag); (void)hipFree(st->d_coef); (void)hipFree(st->d_ret);
    (void)hipSetDevice(prev);
    delete st;
}

extern "C" asx_stream *asx_stream_create(size_t max_sample_len, int device)
{
    if (max_sample_len == 0) { fail("asx_stream_create: max_sample_len must be > 0"); return nullptr; }
    if (asx_device_count() == 0) { fail("no usable HIP device; this library has no CPU fallback"); return nullptr; }
    if (device < 0) HIP_TRY_NULL(hipGetDevice(&device));
    DevGuard dg(device);
    if (!dg.ok) { fail("cannot select device %d", device); return nullptr; }
    asx_stream *st = new asx_stream();
    st->device = device;
    st->cap = max_sample_len;
    const size_t n = max_sample_len;
    if (hipMalloc((void **)&st->src64, 2 * n * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&st->smp64, n * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&st->src32, 2 * n * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&st->smp32, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&st->d_lag, sizeof(int64_t)) != hipSuccess ||
        hipMalloc((void **)&st->d_coef, sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&st->d_ret, sizeof(int32_t)) != hipSuccess ||
        hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking) != hipSuccess) {
        fail("asx_stream_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        asx_stream_destroy(st);
        return nullptr;
    }
    return st;
}

extern "C" int asx_stream_lengths(const asx_stream *st, size_t *n_source, size_t *n_sample)
{
    if (!st) return -1;
    if (n_source) *n_source = st->n_src;
    if (n_sample) *n_sample = st->n_smp;
    return 0;
}

extern "C" int asx_stream_reset(asx_stream *st)
{
    if (!st) return -1;
    std::lock_guard<std::mutex> guard(st->lock);
    st->n_src = st->n_smp = 0;
    return 0;
}

extern "C" int asx_stream_append_f64(asx_stream *st, const double *source_frames, size_t n_source,
                                     const double *sample_frames, size_t n_sample)
{
    if (!st || (n_source && !source_frames) || (n_sample && !sample_frames))
        return fail("asx_stream_append_f64: null argument");
    std::lock_guard<std::mutex> guard(st->lock);
    if (st->n_src + n_source > 2 * st->cap || st->n_smp + n_sample > st->cap)
        return fail("asx_stream_append_f64: capacity exceeded");
    DevGuard dg(st->device);
    if (!dg.ok) return fail("cannot select device %d", st->device);
    // Both tracks' new frames go up on the stream's own HIP stream -- from page-locked memory (asx_host_malloc) by DMA,
    // both copies and both conversions in flight together -- and one synchronisation orders them before the plans'
    // streams, which only start after this function has returned.
    // (No float32 narrowing here, unlike asx_xcorr_f64: the frames come from page-locked memory and go by DMA as they are;
    //  a CPU pass over them to save PCIe bytes measured SLOWER, 1.98 against 1.76 ms for the six intervals.)
    if (n_source) {
        HIP_TRY(hipMemcpyAsync(st->src64 + st->n_src, source_frames, n_source * sizeof(double), hipMemcpyHostToDevice, st->s));
        asx_launch_cvt_f64_f32(st->src64 + st->n_src, st->src32 + st->n_src, n_source, st->s);
    }
    if (n_sample) {
        HIP_TRY(hipMemcpyAsync(st->smp64 + st->n_smp, sample_frames, n_sample * sizeof(double), hipMemcpyHostToDevice, st->s));
        asx_launch_cvt_f64_f32(st->smp64 + st->n_smp, st->smp32 + st->n_smp, n_sample, st->s);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st->s));
    st->n_src += n_source;
    st->n_smp += n_sample;
    return 0;
}

extern "C" int asx_stream_xcorr(asx_stream *st, size_t sample_len, long *lag, double *coefficient)
{
    if (!st || !lag || !coefficient || sample_len == 0) return fail("asx_stream_xcorr: bad argument");
    std::lock_guard<std::mutex> guard(st->lock);
    if (st->n_smp < sample_len || st->n_src < 2 * sample_len)
        return fail("asx_stream_xcorr: only %zu/%zu frames resident, %zu/%zu needed", st->n_src, st->n_smp,
                    2 * sample_len, sample_len);
    asx_plan *p = nullptr;
    for (asx_plan *q : st->plans)
        if (q->host.N == sample_len) p = q;
    if (!p) {
        p = asx_plan_create(sample_len, 1, st->device);
        if (!p) return -1;
        st->plans.push_back(p);
    }
    std::lock_guard<std::mutex> pguard(p->lock);
    DevGuard dg(st->device);
    if (!dg.ok) return fail("cannot select device %d", st->device);
    hipStream_t s = p->stream;
    prof_begin_call(p);
    if (run_group<double>(p, st->src32, st->smp32, st->src64, st->smp64, 1, st->d_lag, st->d_coef, st->d_ret,
                          nullptr, s, 0))
        return -1;
    int64_t h_lag = 0;
    double h_coef = 0;
    int32_t h_ret = -1;
    for (int pass = 0; pass < 2; pass++) { // as in asx_xcorr_f64
        HIP_TRY(hipMemcpyAsync(&h_lag, st->d_lag, sizeof(h_lag), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&h_coef, st->d_coef, sizeof(h_coef), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&h_ret, st->d_ret, sizeof(h_ret), hipMemcpyDeviceToHost, s));
        if (pass == 1) { HIP_TRY(hipStreamSynchronize(s)); break; }
        const int looked = resolve_overflows<double>(p, st->smp32, st->src64, st->smp64, st->d_lag, st->d_coef, st->d_ret, s);
        if (looked < 0) return -1;
        if (looked == 0) break;
    }
    *lag = (long)h_lag;
    *coefficient = h_coef;
    return h_ret;
}

// ---------------------------------------------------------------------------
// synthetic inputs, timing, raw memory helpers
// ---------------------------------------------------------------------------
extern "C" int asx_synth_pairs_dev(uint64_t seed, uint64_t first_pair, size_t count, size_t sample_len,
                                   int noise_shift, float *d_source, float *d_sample,
                                   int64_t *d_true_lag, void *stream)
{
    if (!d_source || !d_sample) return fail("asx_synth_pairs_dev: null argument");
    if (sample_len == 0 || sample_len > (1u << 30)) return fail("asx_synth_pairs_dev: bad sample_len");
    for (size_t done = 0; done < count; done += 32768) {
        const size_t c = std::min((size_t)32768, count - done);
        asx_launch_synth(seed, first_pair + done, c, (uint32_t)sample_len, noise_shift,
                         d_source + done * 2 * sample_len, d_sample + done * sample_len,
                         d_true_lag ? d_true_lag + done : nullptr, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int asx_plan_set_profiling(asx_plan *p, int depth)
{
    if (!p) return -1;
    std::lock_guard<std::mutex> guard(p->lock);
    p->profiling = depth > 0;
    p->prof_depth = depth > 0 ? (size_t)depth : 1;
    p->prof_calls = 0;
    p->prof_groups.assign(p->prof_depth, 0);
    if (p->evr.size() < p->prof_depth) p->evr.resize(p->prof_depth);
    p->ev_groups = 0;
    return 0;
}

// timings of the batch call `calls_back` calls ago (0 = the latest) of the profiling ring
extern "C" int asx_plan_timings_ms(asx_plan *p, int calls_back, float out[6])
{
    if (!p || !out) return -1;
    std::lock_guard<std::mutex> guard(p->lock);
    for (int i = 0; i < 6; i++) out[i] = 0.f;
    if (!p->profiling || calls_back < 0 || (size_t)calls_back >= p->prof_depth || (size_t)calls_back >= p->prof_calls)
        return fail("no profiled call recorded %d calls back", calls_back);
    const size_t ring = (p->prof_calls - 1 - (size_t)calls_back) % p->prof_depth;
    const size_t groups = p->prof_groups[ring];
    if (groups == 0) return fail("no profiled call recorded");
    if (p->evr.size() <= ring || p->evr[ring].size() < groups * 6) return fail("no profiled call recorded");
    const hipEvent_t *ev = p->evr[ring].data();
    DevGuard dg(p->device);
    for (size_t g = 0; g < groups; g++) {
        HIP_TRY(hipEventSynchronize(ev[g * 6 + 5]));
        for (int k = 0; k < 5; k++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev[g * 6 + k], ev[g * 6 + k + 1]));
            out[k] += ms;
        }
    }
    float total = 0.f;
    HIP_TRY(hipEventElapsedTime(&total, ev[0], ev[(groups - 1) * 6 + 5]));
    out[5] = total;
    return 0;
}

extern "C" int asx_plan_last_timings_ms(asx_plan *p, float out[6]) { return asx_plan_timings_ms(p, 0, out); }

extern "C" void *asx_device_malloc(size_t bytes, int device)
{
    int prev = 0;
    if (device >= 0) {
        HIP_TRY_NULL(hipGetDevice(&prev));
        HIP_TRY_NULL(hipSetDevice(device));
    }
    void *ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, bytes ? bytes : 1);
    if (device >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        fail("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return ptr;
}

extern "C" int asx_device_free(void *ptr)
{
    HIP_TRY(hipFree(ptr));
    return 0;
}

extern "C" void *asx_host_malloc(size_t bytes)
{
    void *ptr = nullptr;
    hipError_t e = hipHostMalloc(&ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        fail("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return ptr;
}

extern "C" int asx_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

extern "C" int asx_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int asx_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int asx_stream_sync(asx_plan *p, void *stream)
{
    hipStream_t s = stream ? (hipStream_t)stream : (p ? p->stream : nullptr);
    if (p) {
        DevGuard dg(p->device);
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------
// planning arithmetic exposed for the CPU test-suite (no HIP calls)
// ---------------------------------------------------------------------------
extern "C" int asx_planmath_describe(size_t sample_len, const char *split, uint32_t *F, uint32_t *src_valid,
                                     int *M1, int *M2, int *T, int *nst1, int *radix1, int *nst2,
                                     int *radix2)
{
    AsxHostPlan h;
    std::string err = asx_host_plan_build(sample_len, split, &h);
    if (!err.empty()) return fail("%s", err.c_str());
    *F = h.F; *src_valid = h.src_valid; *M1 = h.M1; *M2 = h.M2; *T = h.T;
    *nst1 = h.st1.nstages; *nst2 = h.st2.nstages;
    for (int i = 0; i < h.st1.nstages; i++) radix1[i] = h.st1.radix[i];
    for (int i = 0; i < h.st2.nstages; i++) radix2[i] = h.st2.radix[i];
    return 0;
}

// candidates of the measured mode, newline-separated "M1xM2xT" strings; returns their number
extern "C" int asx_planmath_candidates(size_t sample_len, size_t max_count, char *out, size_t cap)
{
    const std::vector<std::string> c = asx_host_plan_candidates(sample_len, max_count);
    std::string joined;
    for (const std::string &x : c) { joined += x; joined += '\n'; }
    if (joined.size() + 1 > cap) return fail("candidate list larger than buffer");
    memcpy(out, joined.c_str(), joined.size() + 1);
    return (int)c.size();
}

// tables: which = 0 pos1_of_k1 (M1 ints), 1 k1_of_pos1 (M1), 2 pos2_of_k2 (M2)
extern "C" int asx_planmath_table(size_t sample_len, const char *split, int which, int *out, size_t cap)
{
    AsxHostPlan h;
    std::string err = asx_host_plan_build(sample_len, split, &h);
    if (!err.empty()) return fail("%s", err.c_str());
    const std::vector<int> &t = which == 0 ? h.pos1_of_k1 : which == 1 ? h.k1_of_pos1 : h.pos2_of_k2;
    if (t.size() > cap) return fail("table larger than buffer");
    memcpy(out, t.data(), t.size() * sizeof(int));
    return (int)t.size();
}

// twiddle tables: which = 0 tw1, 1 tw2, 2 tw_lo, 3 tw_hi ; out = interleaved re,im floats
extern "C" int asx_planmath_twiddles(size_t sample_len, const char *split, int which, float *out, size_t cap)
{
    AsxHostPlan h;
    std::string err = asx_host_plan_build(sample_len, split, &h);
    if (!err.empty()) return fail("%s", err.c_str());
    const std::vector<float2> &t =
        which == 0 ? h.tw1 : which == 1 ? h.tw2 : which == 2 ? h.tw_lo : h.tw_hi;
    if (t.size() > cap) return fail("table larger than buffer");
    memcpy(out, t.data(), t.size() * sizeof(float2));
    return (int)t.size();
}
