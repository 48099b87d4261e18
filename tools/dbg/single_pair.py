"""tools/dbg/single_pair.py [N] -- one resident pair: wall time per call against the kernels' own time (HIP events)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1440000
d_src = torch.empty(2 * n, dtype=torch.float32, device="cuda"); d_smp = torch.empty(n, dtype=torch.float32, device="cuda")
d_true = torch.empty(1, dtype=torch.int64, device="cuda"); d_lag = torch.zeros(1, dtype=torch.int64, device="cuda")
d_coef = torch.zeros(1, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(1, dtype=torch.int32, device="cuda")
asx.synth_pairs_dev(7, 0, 1, n, 1, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), 0)
torch.cuda.synchronize()
plan = asx.Plan(n, 1, 0)
def call():
    plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), 1, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    plan.sync()
for _ in range(50): call()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
print("N %d wall per call: median %.1f us, min %.1f us; lag ok %s" % (n, statistics.median(ts) * 1e6, min(ts) * 1e6, bool(torch.equal(d_lag, d_true))))
plan.set_profiling(8)
for _ in range(8): call()
rows = [plan.last_timings_ms(b) for b in range(8)]
print("kernel us (median of 8):", {k: round(statistics.median(r[k] for r in rows) * 1e3, 1) for k in rows[0]})
# back-to-back calls without a sync in between: the launch-bound rate
t0 = time.perf_counter()
for _ in range(200):
    plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), 1, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
plan.sync()
print("200 calls back to back: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
