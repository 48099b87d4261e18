"""tools/dbg/fwd_variance_ws.py -- the counterpart of fwd_variance.py: the INPUTS stay where they are, the plan (its
workspaces) is destroyed and created again, with spacers that stay allocated in between."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n, pairs = 1440000, 124
d_lag = torch.zeros(pairs, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(pairs, dtype=torch.float64, device="cuda")
d_ret = torch.zeros(pairs, dtype=torch.int32, device="cuda"); d_true = torch.empty(pairs, dtype=torch.int64, device="cuda")
d_src = torch.empty(pairs * 2 * n, dtype=torch.float32, device="cuda"); d_smp = torch.empty(pairs * n, dtype=torch.float32, device="cuda")
asx.synth_pairs_dev(7, 0, pairs, n, 1, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), 0)
torch.cuda.synchronize()
def med(plan, steps=40):
    for _ in range(30):
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    plan.set_profiling(steps)
    for _ in range(steps):
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    torch.cuda.synchronize()
    rows = [plan.last_timings_ms(b) for b in range(steps)]
    plan.set_profiling(0)
    return {k: round(statistics.median(r[k] for r in rows), 4) for k in ("fwd_cols", "rows", "inv_cols", "pearson", "total")}
keep = []
for i in range(8):
    plan = asx.Plan(n, pairs, 0)
    print("plan %d: %s" % (i, med(plan)), flush=True)
    plan.close()
    if i % 2 == 1:
        keep.append(torch.empty((3 + i) << 28, dtype=torch.uint8, device="cuda"))   # holds on to some memory: the next plan lands elsewhere
