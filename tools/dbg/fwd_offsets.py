"""tools/dbg/fwd_offsets.py -- k_fwd_cols against the relative placement of its two inputs inside ONE slab (fixed physical pages):
the sample array is moved by `off` bytes; kernel medians for every offset."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n, pairs = 1440000, 124
d_lag = torch.zeros(pairs, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(pairs, dtype=torch.float64, device="cuda")
d_ret = torch.zeros(pairs, dtype=torch.int32, device="cuda"); d_true = torch.empty(pairs, dtype=torch.int64, device="cuda")
plan = asx.Plan(n, pairs, 0)
slab = torch.empty(pairs * 3 * n + (64 << 20), dtype=torch.float32, device="cuda")
def med(ps, pm, steps=30):
    for _ in range(25):
        plan.xcorr_batch_dev(ps, pm, pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    plan.set_profiling(steps)
    for _ in range(steps):
        plan.xcorr_batch_dev(ps, pm, pairs, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
    torch.cuda.synchronize()
    rows = [plan.last_timings_ms(b) for b in range(steps)]
    plan.set_profiling(0)
    return {k: round(statistics.median(r[k] for r in rows), 4) for k in ("fwd_cols", "rows", "inv_cols", "pearson", "total")}
base = slab.data_ptr()
offs = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 0]
both = len(sys.argv) > 1 and sys.argv[1] == "both"
for off in offs:
    ps = base + (off if both else 0); pm = base + pairs * 2 * n * 4 + (32 << 20) + (off if both else 0)
    asx.synth_pairs_dev(7, 0, pairs, n, 1, ps, pm, d_true.data_ptr(), 0)
    torch.cuda.synchronize()
    m = med(ps, pm)
    ok = bool(torch.equal(d_lag, d_true))
    print("off %9d: %s ok %s" % (off, m, ok), flush=True)
