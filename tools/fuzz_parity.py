#!/usr/bin/env python3
"""tools/fuzz_parity.py [seconds] [seed] -- randomized differential run of the HIP path against the oracle (GPU box).
Longer and wider than tests/test_gpu_parity.py::test_randomized_differential: lengths up to 300 000 including the
production lengths 144 000 and 288 000 (compile-time-schedule kernels: wave pairs, fed stages) and lengths that run the
run-time-schedule kernels, tonal / low-pass / sparse / offset / noise signals, both entry points (batched float32 and
cross_correlation(double*)).  Lag must equal the oracle's whenever its margin is defined, coefficient within 1e-5."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
import __graft_entry__ as g
import oracle
asx = g.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = ctypes.CDLL(os.path.join(g.PKG_DIR, "libaudiosync.so"))
dp = ctypes.POINTER(ctypes.c_double)
L.cross_correlation.restype = ctypes.c_int
L.cross_correlation.argtypes = [dp, dp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_long), dp]
PROD = [144000, 288000, 480000] if os.environ.get("FUZZ_BIG") == "1" else [144000, 288000]
if os.environ.get("FUZZ_HUGE") == "1": PROD = [720000, 960000, 1440000]
t0 = time.time(); trials = checked = 0; layouts = {}; modes = [0, 0, 0]; worst = 0.0; said = t0
while time.time() - t0 < budget:
    if time.time() - said > 60.0:   # gpurun takes a command that writes nothing for seven minutes to be hung
        said = time.time(); print("... %d problems, %d pairs checked after %.0f s" % (trials, checked, said - t0), flush=True)
    r = rng.uniform()
    if os.environ.get("FUZZ_BIG") == "1": r *= 0.5   # FUZZ_BIG=1: only the production and the large lengths
    if os.environ.get("FUZZ_HUGE") == "1": r = 0.0   # FUZZ_HUGE=1: only 15 / 20 / 30 s (720 000, 960 000, 1 440 000)
    if r < 0.15: n = int(rng.choice(PROD))
    elif r < 0.35: n = int(rng.integers(20000, 300001))
    elif r < 0.5: n = int(rng.choice([48000, 44100, 96000, 65536, 100000, 131072, 250000]))
    else: n = int(rng.integers(1, 20001))
    batch = int(rng.integers(1, 4)) if n < 100000 else 1
    kind = int(rng.integers(0, 8))
    i = np.arange(2 * n, dtype=np.float64)
    src = rng.uniform(-1, 1, (batch, 2 * n))
    if kind == 1:
        k = int(rng.integers(2, 40)); src = np.cumsum(src, axis=1); src[:, k:] = src[:, k:] - src[:, :-k]
    elif kind == 2: src = np.where(rng.uniform(size=src.shape) < 0.01, src, 0.0)
    elif kind == 3: src = src + rng.uniform(-100, 100)
    elif kind == 4: src = np.sin(rng.uniform(0.001, 3.0) * i)[None, :] + 0.01 * src
    elif kind == 5: src = np.sign(np.sin(rng.uniform(0.01, 1.0) * i))[None, :] * np.ones((batch, 1)) + 1e-3 * src
    elif kind == 6: src = src + 10.0 ** rng.uniform(2, 4) * rng.choice([-1, 1])     # a large offset; the sample gets one below
    both_offset = 10.0 ** rng.uniform(2, 4) * rng.choice([-1, 1]) if kind == 6 else 0.0
    silent = kind == 7 and rng.uniform() < 0.5
    smp = np.empty((batch, n))
    for b in range(batch):
        d = int(rng.integers(-n + 1, n)) if n > 1 else 0
        idx = np.arange(n) + d
        ok = (idx >= 0) & (idx < 2 * n)
        smp[b] = np.where(ok, rng.choice([0.7, -0.4]) * src[b, np.clip(idx, 0, 2 * n - 1)], 0.0) + rng.choice([0.0, 0.01, 0.3]) * rng.uniform(-1, 1, n)
        if kind == 6: smp[b] = smp[b] - smp[b].mean() + both_offset                   # offsets of 1e2 .. 1e4 sigma in BOTH tracks
        if silent: smp[b] = 0.0                                                       # a digitally silent capture
    s32, t32 = src.astype(np.float32), smp.astype(np.float32)
    with asx.Plan(n, batch, 0) as plan:
        lag, coef, ret = plan.xcorr_batch_f32(s32, t32)
        # the device-resident entry point (what bench.py times): the same answers, bit for bit -- it takes the second look
        # at overflowed pairs itself since round 4
        d_s = torch.from_numpy(s32).cuda(); d_t = torch.from_numpy(t32).cuda()
        d_lag = torch.full((batch,), -99, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda")
        d_ret = torch.full((batch,), 7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        plan.xcorr_batch_dev(d_s.data_ptr(), d_t.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr())
        plan.sync()
        assert np.array_equal(d_lag.cpu().numpy(), np.asarray(lag)) and np.array_equal(d_ret.cpu().numpy(), np.asarray(ret)), ("dev", trials, n, kind)
        dc = d_coef.cpu().numpy()
        assert all((a == b) or (a != a and b != b) for a, b in zip(dc, np.asarray(coef))), ("dev coef", trials, n, kind)
        layouts[plan.layout] = layouts.get(plan.layout, 0) + 1
        modes = [a + b for a, b in zip(modes, plan.pearson_modes())]   # both entry points count (round 5: the spectral Pearson form)
    for b in range(batch):
        o_ret, o_lag, o_coef, o_r, margin = oracle.cross_correlation(s32[b], t32[b], want_results=True)
        if margin < 1.0 + (1e-11 if kind == 6 else 1e-9) and not silent:
            continue
        assert int(ret[b]) == o_ret, ("ret", trials, n, kind, b, int(ret[b]), o_ret)
        assert int(lag[b]) == o_lag, ("lag", trials, n, kind, b, int(lag[b]), o_lag, margin)
        if o_ret == 0:
            assert abs(float(coef[b]) - o_coef) < 1e-5, ("coef", trials, n, kind, b, float(coef[b]), o_coef)
            worst = max(worst, abs(float(coef[b]) - o_coef))
        checked += 1
    if trials % 7 == 0:   # the reference API on the first pair, float64 inputs with low bits float32 cannot hold
        s64 = s32[0].astype(np.float64) * (1.0 + 1e-9); t64 = t32[0].astype(np.float64) * (1.0 - 3e-10)
        lg = ctypes.c_long(0); cf = ctypes.c_double(0)
        rc = L.cross_correlation(s64.ctypes.data_as(dp), t64.ctypes.data_as(dp), n, ctypes.byref(lg), ctypes.byref(cf))
        o_ret, o_lag, o_coef, o_r, margin = oracle.cross_correlation(s64, t64, want_results=True)
        if margin >= 1.0 + (1e-11 if kind == 6 else 1e-9) or silent:
            assert rc == o_ret and lg.value == o_lag, ("f64", trials, n, kind, rc, o_ret, lg.value, o_lag, margin)
            if o_ret == 0:
                assert abs(cf.value - o_coef) < 1e-5, ("f64 coef", trials, n, cf.value, o_coef)
            checked += 1
    trials += 1
print("fuzz ok: %d problems, %d pairs checked against the oracle in %.0f s; plans by decomposition: %s; pairs by Pearson form "
      "(spectral, + wrap-around correction, direct under the spectral setting) %s; worst |coefficient - oracle| %.3g"
      % (trials, checked, time.time() - t0, layouts, modes, worst))
