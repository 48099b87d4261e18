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
# share of A(N) attributed to each kernel family (DESIGN.md "Algorithmic bytes")
KERNEL_ALGO_BYTES = {"fwd_cols": 28, "rows": 16 + 0, "inv_cols": 0, "pearson": 8}


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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    asx = graft.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    n = args.sample_len
    batch = args.batch or max(8, min(4096, (1 << 30) // (12 * n)))   # ~1 GiB of inputs per GPU
    d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device=dev)
    d_smp = torch.empty(batch * n, dtype=torch.float32, device=dev)
    d_true = torch.empty(batch, dtype=torch.int64, device=dev)
    d_lag = torch.zeros(batch, dtype=torch.int64, device=dev)
    d_coef = torch.zeros(batch, dtype=torch.float64, device=dev)
    d_ret = torch.zeros(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # distinct pairs on every rank: pair ids [rank*batch, (rank+1)*batch)
    asx.synth_pairs_dev(20260101, rank * batch, batch, n, args.noise_shift, d_src.data_ptr(),
                        d_smp.data_ptr(), d_true.data_ptr(), stream)
    plan = asx.Plan(n, batch, torch.cuda.current_device(), split=args.split)

    def step():
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(),
                             d_coef.data_ptr(), d_ret.data_ptr(), stream)
        if world > 1:
            # the only exchange on this path: gather (lag, coef, ret) of every shard
            out_lag = [torch.empty_like(d_lag) for _ in range(world)]
            out_coef = [torch.empty_like(d_coef) for _ in range(world)]
            dist.all_gather(out_lag, d_lag)
            dist.all_gather(out_coef, d_coef)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness of what was timed: the planted delays
    ok = bool(torch.equal(d_lag, d_true)) and int(d_ret.abs().sum()) == 0

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
        dom = max(("fwd_cols", "rows", "inv_cols", "pearson"), key=lambda k: timings[k])
        # roofline of the whole path over the in-stream time of one step (events), and of the
        # dominant kernel on its own share of the algorithmic bytes
        path_gbs = BYTES_PER_FRAME * n * batch / (timings["total"] * 1e-3) / 1e9
        dom_launch_ms = timings[dom] / groups
        dom_bytes = {"fwd_cols": 28, "rows": 24, "inv_cols": 8, "pearson": 8}[dom] * n * min(batch, plan.group)
        roofline = {
            "bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": path_gbs / HBM_PEAK_GBS, "traffic": None,
            "basis": "52*N algorithmic bytes per pair over the event-timed stream time of one batch",
            "dominant_kernel": {"name": "k_" + dom, "avg_launch_ms": dom_launch_ms,
                                "launches_per_step": groups,
                                "achieved_GBs": dom_bytes / (dom_launch_ms * 1e-3) / 1e9,
                                "bytes_per_launch": dom_bytes},
            "kernel_ms_per_step": {k: timings[k] for k in ("fwd_cols", "rows", "inv_cols", "finalize", "pearson", "total")},
        }
        m1, m2, tcols = plan.split
        line = {
            "metric": "cross-correlations/sec (N=%d float32 pairs)" % n,
            "value": value, "unit": "cross-correlations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batched xcorr, N=%d frames/sample (source 2N), %d pairs per GPU per step, "
                                   "48 kHz mono float32, planted delays, SNR 0 dB" % (n, batch),
                       "sample_len": n, "pairs_per_gpu": batch, "group": plan.group,
                       "split": "%dx%dx%d" % (m1, m2, tcols), "threads_cols_rows": list(plan.threads), "parallelism": "pairs sharded over %d GPU(s), RCCL all_gather of results" % world},
            "results_ok": ok,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
