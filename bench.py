#!/usr/bin/env python3
"""bench.py — batched FFT cross-correlation throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (asx_xcorr_batch_f32_dev: rfft, rfft,
conj-multiply, irfft, |.|-argmax, Pearson) over one batch of synthetic 48 kHz
mono float32 pairs already resident in HBM.  N=1 workload: BASELINE.json's
headline, N = 1 440 000 frames per sample.  With N>1 ranks (torchrun, one
process per GPU) the pairs are sharded over ranks, no data-path collective;
RCCL only gathers the 20-byte results (weak scaling: per-GPU batch fixed).

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
`roofline` (algorithmic bytes 52*N per pair, SURVEY.md 8d, over the measured
time of the dominant kernel) and `cpu_baseline` (the oracle port timed on the
host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_FRAME = 52             # A(N) = 52*N bytes per cross-correlation


def cpu_baseline(sample_len, seconds_budget=20.0):
    """the oracle (C restatement of the reference, float64) on this host's cores,
    one worker per core over distinct pairs; bounded sample."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    cores = os.cpu_count() or 1
    workers = min(cores, 16)
    pairs = [oracle.synth_pair(1, p, sample_len, 1) for p in range(workers)]
    oracle.cross_correlation(pairs[0][0][: 2 * 4800], pairs[0][1][:4800])  # load the library
    t0 = time.perf_counter()
    oracle.cross_correlation(pairs[0][0], pairs[0][1])
    one = time.perf_counter() - t0
    rounds = max(1, int(seconds_budget / max(one, 1e-3) / 1.5))
    rounds = min(rounds, 4)

    def work(i):
        for _ in range(rounds):
            oracle.cross_correlation(pairs[i][0], pairs[i][1])
        return rounds

    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:
        done = sum(ex.map(work, range(workers)))
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "cross-correlations/s", "cores": workers, "kind": "port",
            "sample": "%d pairs of N=%d float32 (widened to float64), oracle/xcorr_oracle.c, "
                      "%d threads, single-pair latency %.3f s" % (done, sample_len, workers, one)}


def side_mode(args):
    """BASELINE configs 2 and 5 on one GPU; not the driver's contract line, same JSON style."""
    import numpy as np
    import torch
    import __graft_entry__ as graft
    import oracle
    asx = graft.load()
    torch.cuda.set_device(0)
    sr = 48000
    n_max = 30 * sr
    src32, smp32, true_lag = oracle.synth_pair(20260101, 0, n_max, 1)
    if args.mode == "streaming":
        # a delay every prefix can see (the generator's is up to +-0.75*N of the 30 s window)
        true_lag = 12345
        rng = np.random.default_rng(1)
        smp32 = (0.5 * src32[true_lag: true_lag + n_max] + 0.25 * rng.uniform(-1, 1, n_max)).astype(np.float32)
    src, smp = src32.astype(np.float64), smp32.astype(np.float64)
    out = {"n_gpus": 1, "data": "synthetic", "dtype": "f32", "true_lag": int(true_lag)}
    if args.mode == "streaming":
        seconds = (3, 6, 10, 15, 20, 30)                       # src/audiosync.c:50-57
        st = asx.Stream(n_max, 0)
        for s_ in seconds:                                      # plans built once, outside the timing
            st.append(src[st.lengths()[0]: 2 * s_ * sr], smp[st.lengths()[1]: s_ * sr]); st.xcorr(s_ * sr)
        per = {}
        reps = max(1, args.steps)
        t_all = 0.0
        for _ in range(reps):
            st.reset()
            for s_ in seconds:
                n = s_ * sr
                t0 = time.perf_counter()
                a, b = st.lengths()
                st.append(src[a: 2 * n], smp[b: n])
                ret, lag, coef = st.xcorr(n)
                dt = time.perf_counter() - t0
                per[s_] = per.get(s_, 0.0) + dt / reps
                t_all += dt / reps
                assert ret == 0 and lag == true_lag
        out.update({"metric": "growing-window run, 6 intervals 144000..1440000 frames, incremental upload + plan reuse",
                    "value": t_all * 1e3, "unit": "ms", "higher_is_better": False,
                    "ms_per_interval": {str(k): v * 1e3 for k, v in per.items()},
                    "config": {"workload": "streaming 3/6/10/15/20/30 s prefixes of one 30 s pair, f64 host buffers"}})
    else:
        n = args.sample_len
        plan = asx.Plan(n, 1, 0, split=args.split)
        d_src = torch.from_numpy(src32[: 2 * n]).cuda(); d_smp = torch.from_numpy(smp32[:n]).cuda()
        d_lag = torch.zeros(1, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(1, dtype=torch.float64, device="cuda")
        d_ret = torch.zeros(1, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        def resident():
            plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), 1, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), stream)
            torch.cuda.synchronize()
        def dropin():
            return plan.xcorr_f64(src[: 2 * n], smp[:n])
        for f in (resident, dropin):
            for _ in range(3): f()
        reps = max(5, args.steps)
        t0 = time.perf_counter()
        for _ in range(reps): resident()
        t_res = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps): dropin()
        t_abi = (time.perf_counter() - t0) / reps
        out.update({"metric": "single pair latency N=%d" % n, "value": t_res * 1e3, "unit": "ms", "higher_is_better": False,
                    "resident_float32_ms": t_res * 1e3, "double_abi_incl_h2d_ms": t_abi * 1e3,
                    "config": {"workload": "one pair, N=%d; resident float32 vs cross_correlation(double*) incl. PCIe" % n}})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sample-len", type=int, default=1440000)
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = auto)")
    ap.add_argument("--noise-shift", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--split", default=None)
    ap.add_argument("--mode", default="batched", choices=["batched", "streaming", "single"],
                    help="batched = the headline workload (default); streaming = BASELINE config 5 "
                         "(growing window 3..30 s, plan reuse); single = config 2 (one pair, latency)")
    args = ap.parse_args()
    if args.mode != "batched":
        return side_mode(args)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    asx = graft.load()
    from audiosync_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ASX_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, result gather) with a single
    # rank too -- the only way to exercise it on a one-GPU box (launch under torch.distributed.run)
    multi = world > 1 or os.environ.get("ASX_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    n = args.sample_len
    batch = args.batch or max(8, min(4096, (2 << 30) // (12 * n)))   # ~2 GiB of inputs per GPU
    d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device=dev)
    d_smp = torch.empty(batch * n, dtype=torch.float32, device=dev)
    d_true = torch.empty(batch, dtype=torch.int64, device=dev)
    # a shard's results live back to back in one byte buffer so that the gather is ONE collective with
    # no packing kernels (sharding.result_buffer)
    res_buf, (d_lag, d_coef, d_ret) = sharding.result_buffer(batch, dev)
    gather_out = torch.empty((world, sharding.result_bytes(batch)), dtype=torch.uint8, device=dev) if multi else None
    gathered = [None]
    stream = torch.cuda.current_stream().cuda_stream
    # distinct pairs on every rank: pair ids [rank*batch, (rank+1)*batch)
    asx.synth_pairs_dev(20260101, rank * batch, batch, n, args.noise_shift, d_src.data_ptr(),
                        d_smp.data_ptr(), d_true.data_ptr(), stream)
    plan = asx.Plan(n, batch, torch.cuda.current_device(), split=args.split)

    def step():
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(),
                             d_coef.data_ptr(), d_ret.data_ptr(), stream)
        if multi:
            # the only exchange on this path: RCCL all-gather of the 20 result bytes per pair of every
            # shard, in stream order behind the kernels that produce them
            gathered[0] = sharding.gather_result_buffers(res_buf, batch, out=gather_out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was timed: the planted delays
    ok = bool(torch.equal(d_lag, d_true)) and int(d_ret.abs().sum()) == 0
    if multi and gathered[0] is not None:   # ... and this rank's slice of what the gather delivered
        views = gathered[0][1]
        ok = ok and len(views) == world and bool(torch.equal(views[rank][0], d_true))

    # per-kernel durations, HIP events on the stream the kernels ran on (one extra profiled step)
    plan.set_profiling(True)
    step()
    torch.cuda.synchronize()
    timings = plan.last_timings_ms()
    plan.set_profiling(False)

    if rank == 0:
        pairs_total = batch * world * args.steps
        value = pairs_total / dt
        groups = (batch + plan.group - 1) // plan.group
        per_launch_pairs = min(batch, plan.group)
        fam = ("fwd_cols", "rows", "inv_cols", "pearson")
        dom = max(fam, key=lambda k: timings[k])
        # DESIGN.md "Algorithmic bytes": A(N) = 52*N per cross-correlation, attributed
        # to the kernel that moves them: fwd_cols 28N (inputs in, both spectra out), rows 16N (both
        # spectra in), inv_cols 0 (its 8N read is an intermediate, not algorithmic), pearson 8N.
        algo_share = {"fwd_cols": 28, "rows": 16, "inv_cols": 0, "pearson": 8}
        dom_launch_ms = timings[dom] / groups
        dom_bytes = algo_share[dom] * n * per_launch_pairs
        achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
        path_gbs = BYTES_PER_FRAME * n * value / world / 1e9   # per GPU, from the timed steps
        m1, m2, tcols = plan.split
        split = "%dx%dx%d" % (m1, m2, tcols)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tpath):
            # HBM bytes per launch from the PMC counters (tools/traffic.sh: separate --pmc passes,
            # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); only if it is this workload
            tj = json.load(open(tpath))
            c = tj.get("config", {})
            if c.get("sample_len") == n and c.get("split") == split and c.get("group") == plan.group:
                for kname, kv in tj["kernels"].items():
                    if kname.startswith("k_" + dom):
                        traffic = kv["hbm_bytes_per_launch"]
        roofline = {
            "bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_launch_ms,
            "launches_per_step": groups, "pairs_per_launch": per_launch_pairs,
            "path": {"algorithmic_bytes_per_pair": BYTES_PER_FRAME * n, "achieved": path_gbs,
                     "frac": path_gbs / HBM_PEAK_GBS,
                     "basis": "52*N bytes per pair x pairs/s per GPU over the timed steps; kernel_ms_per_step is one extra event-timed step"},
            "kernel_ms_per_step": {k: timings[k] for k in ("fwd_cols", "rows", "inv_cols", "finalize", "pearson", "total")},
        }
        line = {
            "metric": "cross-correlations/sec (N=%d float32 pairs)" % n,
            "value": value, "unit": "cross-correlations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batched xcorr, N=%d frames/sample (source 2N), %d pairs per GPU per step, "
                                   "48 kHz mono float32, planted delays, SNR 0 dB" % (n, batch),
                       "sample_len": n, "pairs_per_gpu": batch, "group": plan.group,
                       "split": split, "threads_cols_rows": list(plan.threads),
                       "parallelism": "pairs sharded over %d GPU(s), RCCL all_gather of results" % world},
            "results_ok": ok,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(line), flush=True)
    plan.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
