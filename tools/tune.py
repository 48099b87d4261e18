#!/usr/bin/env python3
"""tuning aid (run on the GPU box): time every reasonable split of each production length."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
asx = g.load()
lengths = [int(a) for a in sys.argv[1:]] or [144000, 288000, 480000, 720000, 960000, 1440000]
for n in lengths:
    M = n  # F/2 for smooth lengths
    batch = max(8, min(512, (3 << 28) // (12 * n)))
    d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda")
    d_smp = torch.empty(batch * n, dtype=torch.float32, device="cuda")
    d_true = torch.empty(batch, dtype=torch.int64, device="cuda")
    d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda")
    d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda")
    d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    asx.synth_pairs_dev(7, 0, batch, n, 1, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), st)
    cands = [None]
    for m1 in range(8, 4097):
        if M % m1: continue
        m2 = M // m1
        if m2 > 2048 or m2 < 64: continue
        for t in (8, 16, 32):
            if m1 * t * 8 > 80 * 1024: continue
            if t == 8 or m1 * t * 8 >= 24 * 1024:
                cands.append("%dx%dx%d" % (m1, m2, t))
    res = []
    for sp in cands:
        try:
            d = asx.planmath_describe(n, sp)
        except Exception:
            continue
        if max(d["radix1"] + d["radix2"]) > int(os.environ.get("TUNE_MAXR", "12")) or len(d["radix1"]) > 3 or len(d["radix2"]) > 3:
            continue
        try:
            plan = asx.Plan(n, batch, 0, split=sp)
        except Exception as e:
            continue
        for _ in range(2):
            plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        ok = bool(torch.equal(d_lag, d_true))
        res.append((batch / dt, "%dx%dx%d" % plan.split, plan.threads, ok, sp is None))
        plan.close()
    res.sort(reverse=True)
    if os.environ.get("TUNE_DUMP"):   # every candidate, machine-readable (input of the planner's cost model fit)
        for r in res:
            d = asx.planmath_describe(n, None if r[4] else r[1])
            print("DUMP %d %s %.1f %s %s %d" % (n, r[1], r[0], ",".join(map(str, d["radix1"])), ",".join(map(str, d["radix2"])), int(r[4])))
    print("N=%d batch=%d" % (n, batch))
    for r in res[:6]:
        print("   %9.0f /s  %-14s threads=%s ok=%s%s" % (r[0], r[1], r[2], r[3], "  <- planner default" if r[4] else ""))
    for r in res:
        if r[4]:
            print("   default: %9.0f /s  %s" % (r[0], r[1]))
