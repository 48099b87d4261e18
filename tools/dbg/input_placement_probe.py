"""tools/dbg/input_placement_probe.py [sets] -- does the physical placement of the CALLER's input buffers move the forward column kernel?
Allocates `sets` input sets one after the other (all alive), fills each with the same synthetic pairs, and times the same plan on each
(median ms per step over 30 steps, twice, interleaved)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n, batch = 1440000, 124
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
st = torch.cuda.current_stream().cuda_stream
bufs = []
for k in range(sets):
    d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda"); d_smp = torch.empty(batch * n, dtype=torch.float32, device="cuda")
    asx.synth_pairs_dev(1, 0, batch, n, 1, d_src.data_ptr(), d_smp.data_ptr(), 0, st)
    bufs.append((d_src, d_smp))
d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
plan = asx.Plan(n, batch, 0)
def run(k, steps):
    s_, t_ = bufs[k]
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.xcorr_batch_dev(s_.data_ptr(), t_.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
        plan.sync(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for k in range(sets): run(k, 30)   # clocks up
for rep in range(2):
    print("round %d:" % rep, "  ".join("set %d (%x): %.3f ms" % (k, bufs[k][0].data_ptr() >> 20, run(k, 30)) for k in range(sets)))
