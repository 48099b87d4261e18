"""tools/dbg/fwd_variance.py -- does k_fwd_cols' run-to-run spread (0.95-1.07 ms) follow the placement of its buffers?
One process, several allocations of the inputs (a spacer of a different size in front of each), kernel medians of each."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n, pairs = 1440000, 124
d_lag = torch.zeros(pairs, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(pairs, dtype=torch.float64, device="cuda")
d_ret = torch.zeros(pairs, dtype=torch.int32, device="cuda"); d_true = torch.empty(pairs, dtype=torch.int64, device="cuda")
plan = asx.Plan(n, pairs, 0)
def med(d_src, d_smp, steps=40):
    for _ in range(30):
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    plan.set_profiling(steps)
    for _ in range(steps):
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    torch.cuda.synchronize()
    rows = [plan.last_timings_ms(b) for b in range(steps)]
    plan.set_profiling(0)
    return {k: round(statistics.median(r[k] for r in rows), 4) for k in ("fwd_cols", "rows", "inv_cols", "pearson", "total")}
spacers = [0, 1 << 20, 3 << 20, 17 << 20, (64 << 20) + 4096 * 5, 999 << 20, 0, 0]
for i, sp in enumerate(spacers):
    pad = torch.empty(sp, dtype=torch.uint8, device="cuda") if sp else None
    d_src = torch.empty(pairs * 2 * n, dtype=torch.float32, device="cuda"); d_smp = torch.empty(pairs * n, dtype=torch.float32, device="cuda")
    asx.synth_pairs_dev(7, 0, pairs, n, 1, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), 0)
    torch.cuda.synchronize()
    m = med(d_src, d_smp)
    print("alloc %d spacer %d MB src %#x (mod 2MB %#x) smp %#x (mod 2MB %#x): %s" % (i, sp >> 20, d_src.data_ptr(), d_src.data_ptr() % (2 << 20), d_smp.data_ptr(), d_smp.data_ptr() % (2 << 20), m), flush=True)
    del d_src, d_smp, pad
    torch.cuda.empty_cache()
