#!/usr/bin/env python3
"""tools/dbg/single_pair_loop.py [N] [reps] -- one resident float32 pair, `reps` calls of the batched entry point with batch 1
(BASELINE configs[1]); prints the mean wall time per call.  Run under rocprofv3 --kernel-trace by tools/dbg/single_trace.sh."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
asx = g.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1440000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(1)
big = rng.uniform(-1, 1, 3 * n).astype(np.float32)
src = big[: 2 * n].copy(); smp = (0.5 * big[1000: 1000 + n] + 0.25 * rng.uniform(-1, 1, n)).astype(np.float32)
d_src = torch.from_numpy(src).cuda(); d_smp = torch.from_numpy(smp).cuda()
d_lag = torch.zeros(1, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(1, dtype=torch.float64, device="cuda")
d_ret = torch.zeros(1, dtype=torch.int32, device="cuda")
with asx.Plan(n, 1, 0) as plan:
    def call():
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), 1, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
        plan.sync()
    for _ in range(5): call()
    t0 = time.perf_counter()
    for _ in range(reps): call()
    dt = (time.perf_counter() - t0) / reps
    print("N=%d: %.1f us per call, lag %d ret %d coef %.6f" % (n, dt * 1e6, int(d_lag.item()), int(d_ret.item()), float(d_coef.item())))
