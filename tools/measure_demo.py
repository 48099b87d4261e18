import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
for n in [int(a) for a in sys.argv[1:]]:
    batch = max(8, min(1024, (1 << 29) // (12 * n)))
    d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda")
    d_smp = torch.empty(batch * n, dtype=torch.float32, device="cuda")
    d_true = torch.empty(batch, dtype=torch.int64, device="cuda")
    d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    asx.synth_pairs_dev(7, 0, batch, n, 1, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), st)
    torch.cuda.synchronize()
    out = []
    for mode in (None, "measure"):
        t0 = time.perf_counter()
        plan = asx.Plan(n, batch, 0, split=mode)
        tplan = time.perf_counter() - t0
        for _ in range(2):
            plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out.append("%s: %dx%dx%d %.0f/s (plan %.2f s, lags %s)" % (mode or "model", *plan.split, batch / dt, tplan, "ok" if torch.equal(d_lag, d_true) else "differ from planted"))
        plan.close()
    print("N=%d batch=%d  " % (n, batch) + "   ".join(out))
