import sys, time, ctypes, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, oracle
import __graft_entry__ as graft
mod = graft.load()
n = 144000
rng = np.random.default_rng(41)
for ratio in [1e2, 1e3, 1e4, -1e3]:
    d = int(rng.integers(-n // 2, n // 2)); x = rng.normal(size=3 * n)
    src = (ratio + x[n:3*n]).astype(np.float32); smp = (abs(ratio) * 0.7 + 0.5 * x[n+d:2*n+d] + 0.3 * rng.normal(size=n)).astype(np.float32)
    t0 = time.perf_counter(); o = oracle.cross_correlation(src, smp); t1 = time.perf_counter()
    with mod.Plan(n, 1, 0) as plan:
        plan.xcorr_batch_f32(src[None], smp[None])
        t2 = time.perf_counter(); r = plan.xcorr_batch_f32(src[None], smp[None]); t3 = time.perf_counter()
        rep = plan.peak_repairs()
        t4 = time.perf_counter(); r64 = plan.xcorr_f64(src.astype(np.float64), smp.astype(np.float64)); t5 = time.perf_counter()
    print("ratio %g oracle %.3fs lag %d | batch_f32 %.4fs lag %d repairs %d | f64 %.4fs lag %d" % (ratio, t1-t0, o[1], t3-t2, int(r[0][0]), rep, t5-t4, r64[1]), flush=True)
