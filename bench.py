#!/usr/bin/env python3
"""bench.py — batched FFT cross-correlation throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (asx_xcorr_batch_f32_dev: rfft, rfft, conj-multiply, irfft,
|.|-argmax with exact re-evaluation of near-ties, Pearson) over one batch of synthetic 48 kHz mono
float32 pairs already resident in HBM.

Workloads
  headline  (the JSON line's `value`)  BASELINE.json's metric: N = 1 440 000 frames per sample, one
            launch group of pairs per GPU per step.  With N>1 ranks the per-GPU batch is fixed
            (`"scaling": "weak"`): pairs are independent, there is no data-path collective, RCCL only
            all-gathers the 20 result bytes per pair, inside the timed region.
  config4   (the JSON line's `config4` object)  BASELINE.json configs[3] as written: a FIXED batch of
            8192 pairs, N = 480 000, block-partitioned over the ranks (strong scaling), inputs
            generated on the device per shard, the same gather.
  config4_capi  the same batch through the C-ABI's own multi-device form (ONE process, asx_comm_create over
            one plan per GPU, asx_xcorr_batch_multi_dev): rank 0 runs it behind the distributed legs.

Ranks: `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one process per GPU)
as a CHILD process before anything touches the GPU, and exits with the child's code; under torchrun
(WORLD_SIZE set) it is a rank.  `--dry-run` walks the same rank / shard / gather code with the gloo
backend and fabricated results (no GPU): the CPU test of the N>1 path.

Rank 0 prints ONE JSON line carrying `roofline` (algorithmic bytes 52*N per pair, SURVEY.md 8d, over
the live HIP-event duration of the dominant kernel) and `cpu_baseline` (the reference's cost model on
the host cores: FFTW3 if the node has it, else the oracle's own DFT -- labelled).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_FRAME = 52             # A(N) = 52*N bytes per cross-correlation
COEF_TOL = 1e-5                  # BASELINE.json north_star: Pearson coefficient within 1e-5 of the reference's
ALGO_SHARE = {"fwd_cols": 28, "rows": 16, "inv_cols": 0, "pearson": 8}   # DESIGN.md "Algorithmic bytes"
FAMILIES = ("fwd_cols", "rows", "inv_cols", "finalize", "pearson", "total")
CONFIG4_BATCH, CONFIG4_N = 8192, 480000
# BASELINE.json configs[3] is a STRONG-scaling curve (fixed batch): the speed-up at N GPUs is this object's `value` in the
# N-GPU line over its `value` in the 1-GPU line of the same sweep; the top-level `value` is the weak-scaling headline.
CONFIG4_SPEEDUP_BASIS = ("config4.value(n_gpus = N) / config4.value(n_gpus = 1) of the same sweep: a fixed batch of %d pairs of "
                         "N = %d block-partitioned over the ranks, result all-gather inside the timed region" % (CONFIG4_BATCH, CONFIG4_N))
SEED = 20260101


# --------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1 only)
# --------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup quota if there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _baseline_worker(n, pairs, rounds, first, cpu, ready, go, out):
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})     # one worker per CPU, evenly spread over the usable set: no migration, no sharing
        except (AttributeError, OSError):
            pass
    import oracle
    w = oracle.Worker(n)
    w.run(*pairs[first % len(pairs)])          # touch every buffer once: page faults stay out of the timed region
    ready.wait()
    go.wait()
    done = 0
    for r in range(rounds):
        ret, _, _ = w.run(*pairs[(first + r) % len(pairs)])
        done += 1 if ret == 0 else 0
    out.put(done)


def _spread_cpus(workers):
    """`workers` CPUs of this process's affinity set, evenly spaced (SMT siblings and neighbours last)"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return [None] * workers
    if workers > len(cpus):
        return [cpus[i % len(cpus)] for i in range(workers)]
    return [cpus[(i * len(cpus)) // workers] for i in range(workers)]


def _throughput(n, pairs, workers, rounds):
    """`workers` independent single-threaded PROCESSES, each pinned to its own CPU, each making `rounds` calls on its own buffers"""
    import multiprocessing as mp
    ctx = mp.get_context("fork")               # no GPU context exists in this process yet (see main())
    ready, go, out = ctx.Barrier(workers + 1), ctx.Event(), ctx.Queue()
    cpus = _spread_cpus(workers)
    procs = [ctx.Process(target=_baseline_worker, args=(n, pairs, rounds, i, cpus[i], ready, go, out)) for i in range(workers)]
    for p in procs:
        p.start()
    ready.wait()
    t0 = time.perf_counter()
    go.set()
    done = sum(out.get() for _ in procs)
    dt = time.perf_counter() - t0
    for p in procs:
        p.join()
    return done / dt, dt, done


def cpu_baseline(sample_len, point_seconds=3.0, sweeps=3):
    """The reference's CPU path on this node's host cores (oracle/cpu_baseline.c), run BEFORE this process touches the
    GPU.  (i) faithful single-call latency -- two threads for the forward transforms, plans and allocations per call, the
    cost model of src/cross_correlation.c:33-36,159-239 -- at N and at BASELINE configs[0]'s N = 144 000; (ii) node
    throughput -- independent single-threaded worker processes, one per CPU and pinned to it, plans and buffers kept
    between calls.  Three worker counts (half of, all of, twice the usable CPUs), each measured for
    >= `point_seconds` of wall time per sweep, `sweeps` sweeps; a point's rate is the MEDIAN of its sweeps, `value` the
    best point, `spread` (max - min) / median of that point over the sweeps (round 4's points ran 0.4-1.8 s
    once, and moved 1.6x between rounds).  Backend: FFTW3 if libfftw3.so.3 can be dlopen()ed here, else the oracle's own
    DFT (labelled).  About 45 s in all."""
    import oracle
    cores, usable = os.cpu_count() or 1, cpu_quota()
    backend = oracle.baseline_backend()
    distinct = 8
    pairs = [oracle.synth_pair(1, p, sample_len, 1) for p in range(distinct)]
    pairs = [(s.astype("float64"), t.astype("float64")) for s, t, _ in pairs]
    small = oracle.synth_pair(1, 0, 144000, 1)
    small = (small[0].astype("float64"), small[1].astype("float64"))

    def latency(s, t, reps):
        oracle.cross_correlation_faithful(s, t)
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle.cross_correlation_faithful(s, t)
            lat.append(time.perf_counter() - t0)
        return statistics.median(lat)

    one = latency(*pairs[0], 3)
    one_small = latency(*small, 5)
    w1 = oracle.Worker(sample_len)
    w1.run(*pairs[0])
    t0 = time.perf_counter()
    w1.run(*pairs[1])
    t_worker = time.perf_counter() - t0
    w1.close()
    del w1
    cands = sorted({c for c in (max(1, usable // 2), usable, min(2 * usable, cores)) if c >= 1})
    rounds = max(2, int(point_seconds / max(t_worker, 1e-3) + 0.999))   # calls per worker and point: >= point_seconds of wall time
    _throughput(sample_len, pairs, usable, max(2, rounds // 3))         # one untimed point: page cache, clocks (the first sweep of
    runs = {wk: [] for wk in cands}                                     # every point ran 10-30 % low without it)
    for _ in range(sweeps):
        for wk in cands:
            # more workers than usable CPUs share them: fewer calls each, the same >= point_seconds of wall time per point
            rate, dt, done = _throughput(sample_len, pairs, wk, max(2, rounds * min(wk, usable) // wk))
            runs[wk].append({"calls": done, "seconds": round(dt, 3), "per_s": round(rate, 2)})
    points = []
    for wk in cands:
        rates = sorted(r["per_s"] for r in runs[wk])
        med = statistics.median(rates)
        points.append({"workers": wk, "per_s": med, "spread": round((rates[-1] - rates[0]) / med, 4) if med else None,
                       "sweeps": runs[wk]})
    best = max(points, key=lambda p: p["per_s"])
    return {"value": best["per_s"], "unit": "cross-correlations/s", "cores": best["workers"], "spread": best["spread"],
            "kind": "reference" if backend == "fftw3" else "port", "backend": backend,
            "nproc": cores, "usable_cpus": usable, "cpu_model": cpu_model(), "sweep": points,
            "single_call_latency_s": one, "single_call_latency_s_N144000": one_small,
            "worker_call_s": t_worker,
            "single_call_model": "2 threads for the forward transforms, plan + 4 allocations per call "
                                 "(src/cross_correlation.c:33-36,159-239)",
            "sample": "%d calls of N=%d (float32 values widened to float64) per sweep by %d single-threaded worker processes, each "
                      "pinned to its own CPU (plans and buffers kept per worker), %d distinct pairs, >= %.0f s per point, median of %d "
                      "sweeps over three worker counts, best point; taken before the first GPU call; backend %s" %
                      (best["sweeps"][0]["calls"], sample_len, best["workers"], distinct, point_seconds, sweeps,
                       "FFTW3 (dlopen libfftw3.so.3)" if backend == "fftw3" else "oracle/fft64.c (no libfftw3 on this node)")}


def oracle_pair0(sample_len, noise_shift):
    """cpu_baseline leg only: the oracle's (ret, lag, coefficient) for pair 0 of the bench workload -- oracle.synth_pair is the host
    twin of k_synth, bit for bit (tests/test_gpu_parity.py::test_device_generator_matches_the_oracle's)"""
    import oracle
    src, smp, true_lag = oracle.synth_pair(SEED, 0, sample_len, noise_shift)
    ret, lag, coef = oracle.cross_correlation(src, smp)
    return {"ret": int(ret), "lag": int(lag), "coefficient": float(coef), "planted_lag": int(true_lag)}


# --------------------------------------------------------------------------------------------------
# BASELINE configs 2 and 5 on one GPU (not the driver's contract line, same JSON style)
# --------------------------------------------------------------------------------------------------
def side_mode(args):
    import numpy as np
    import torch
    import __graft_entry__ as graft
    import oracle
    asx = graft.load()
    torch.cuda.set_device(0)
    sr = 48000
    n_max = 30 * sr
    src32, smp32, true_lag = oracle.synth_pair(SEED, 0, n_max, 1)
    if args.mode == "streaming":
        # a delay every prefix can see (the generator's is up to +-0.75*N of the 30 s window)
        true_lag = 12345
        rng = np.random.default_rng(1)
        smp32 = (0.5 * src32[true_lag: true_lag + n_max] + 0.25 * rng.uniform(-1, 1, n_max)).astype(np.float32)
    src, smp = src32.astype(np.float64), smp32.astype(np.float64)
    if args.mode == "streaming":
        # the tracks live where audiosync_run() keeps them: page-locked host memory (host/audiosync.c, asx_host_malloc)
        pin_src, pin_smp = asx.PinnedArray(src.size), asx.PinnedArray(smp.size)
        pin_src.array[:] = src; pin_smp.array[:] = smp
        src, smp = pin_src.array, pin_smp.array
    out = {"n_gpus": 1, "data": "synthetic", "dtype": "f32", "true_lag": int(true_lag)}
    if args.mode == "streaming":
        seconds = (3, 6, 10, 15, 20, 30)                       # src/audiosync.c:50-57
        st = asx.Stream(n_max, 0)
        for s_ in seconds:                                      # plans built once, outside the timing
            st.append(src[st.lengths()[0]: 2 * s_ * sr], smp[st.lengths()[1]: s_ * sr]); st.xcorr(s_ * sr)
        samples = {s_: [] for s_ in seconds}
        reps = max(3, args.steps)
        for _ in range(reps):
            st.reset()
            for s_ in seconds:
                n = s_ * sr
                t0 = time.perf_counter()
                a, b = st.lengths()
                st.append(src[a: 2 * n], smp[b: n])
                ret, lag, coef = st.xcorr(n)
                samples[s_].append(time.perf_counter() - t0)
                assert ret == 0 and lag == true_lag
        # per interval the MEDIAN over the repetitions (round 5: one host hiccup of a millisecond in five repetitions -- either
        # build, any interval -- moved the mean of that interval by 0.2 ms); the means are reported next to it
        per = {s_: statistics.median(v) for s_, v in samples.items()}
        t_all = sum(per.values())
        out.update({"metric": "growing-window run, 6 intervals 144000..1440000 frames, incremental upload + plan reuse",
                    "value": t_all * 1e3, "unit": "ms", "higher_is_better": False, "repetitions": reps,
                    "ms_per_interval": {str(k): v * 1e3 for k, v in per.items()},
                    "ms_per_interval_mean": {str(k): sum(v) / len(v) * 1e3 for k, v in samples.items()},
                    "config": {"workload": "streaming 3/6/10/15/20/30 s prefixes of one 30 s pair, f64 host buffers"}})
    else:
        out.update(single_pair_latency(asx, torch, src32, smp32, args.sample_len, max(5, args.steps), args.split))
    print(json.dumps(out), flush=True)


def single_pair_latency(asx, torch, src32, smp32, n, reps, split=None):
    """BASELINE config 2: one pair, resident float32 and through cross_correlation(double*) incl. the PCIe copy"""
    import numpy as np
    plan = asx.Plan(n, 1, torch.cuda.current_device(), split=split)
    src, smp = src32[: 2 * n].astype(np.float64), smp32[:n].astype(np.float64)
    d_src = torch.from_numpy(src32[: 2 * n]).cuda(); d_smp = torch.from_numpy(smp32[:n]).cuda()
    d_lag = torch.zeros(1, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(1, dtype=torch.float64, device="cuda")
    d_ret = torch.zeros(1, dtype=torch.int32, device="cuda")

    def resident():
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), 1, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), 0)
        plan.sync()

    def dropin():
        return plan.xcorr_f64(src, smp)

    for f in (resident, dropin):
        for _ in range(3):
            f()
    t0 = time.perf_counter()
    for _ in range(reps):
        resident()
    t_res = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        dropin()
    t_abi = (time.perf_counter() - t0) / reps
    plan.close()
    return {"metric": "single pair latency N=%d" % n, "value": t_res * 1e3, "unit": "ms", "higher_is_better": False,
            "resident_float32_ms": t_res * 1e3, "double_abi_incl_h2d_ms": t_abi * 1e3,
            "config": {"workload": "one pair, N=%d; resident float32 vs cross_correlation(double*) incl. PCIe" % n}}


# --------------------------------------------------------------------------------------------------
# BASELINE configs[3] through the C-ABI's own multi-device form: ONE process, one plan per GPU, asx_comm_create +
# asx_xcorr_batch_multi_dev (csrc/asx_api.hip: a host thread and a stream per device, one ncclAllGather of the result records
# inside the library).  The torch.distributed form above it is the headline's; this leg exists so that a multi-GPU run measures
# BOTH forms without a code change.  With one GPU the communicator has one rank.
# --------------------------------------------------------------------------------------------------
def capi_leg(ngpus, steps):
    import numpy as np
    import torch
    import __graft_entry__ as graft
    asx = graft.load()
    n, total = CONFIG4_N, CONFIG4_BATCH
    width = (total + ngpus - 1) // ngpus
    rec = asx.result_bytes(width)
    plans, srcs, smps, trues, gathered, counts = [], [], [], [], [], []
    for g in range(ngpus):
        start, count = asx.shard_range(total, ngpus, g)
        with torch.cuda.device(g):
            c = max(count, 1)
            d_src = torch.empty(c * 2 * n, dtype=torch.float32, device="cuda")
            d_smp = torch.empty(c * n, dtype=torch.float32, device="cuda")
            d_true = torch.full((width,), 0, dtype=torch.int64, device="cuda")
            if count:
                asx.synth_pairs_dev(SEED, start, count, n, 0, d_src.data_ptr(), d_smp.data_ptr(), d_true.data_ptr(), 0)
            torch.cuda.synchronize()
            plans.append(asx.Plan(n, c, g))
            gathered.append(torch.zeros(ngpus * rec, dtype=torch.uint8, device="cuda"))
        srcs.append(d_src); smps.append(d_smp); trues.append(d_true); counts.append(count)
    comm = asx.Comm(plans)
    args = ([t.data_ptr() for t in srcs], [t.data_ptr() for t in smps], counts, width, [t.data_ptr() for t in gathered])
    comm.run(*args)                                   # warm-up (communicator, clocks)
    t0 = time.perf_counter()
    for _ in range(steps):
        comm.run(*args)                               # returns after every device's stream has been synchronised
    dt = time.perf_counter() - t0
    ok = True
    for dev in range(ngpus):                          # every device holds the record of every shard
        g_all = gathered[dev].cpu().numpy()
        for sh in range(ngpus):
            r = g_all[sh * rec: (sh + 1) * rec]
            lag = r[: 8 * width].view(np.int64); ret = r[16 * width: 20 * width].view(np.int32)
            ok = ok and bool(np.array_equal(lag[: counts[sh]], trues[sh].cpu().numpy()[: counts[sh]])) and not ret[: counts[sh]].any()
    comm.close()
    for p_ in plans:
        p_.close()
    return {"metric": "cross-correlations/sec (fixed batch %d x N=%d, C-ABI multi-device form)" % (total, n),
            "value": total * steps / dt, "unit": "cross-correlations/s", "scaling": "strong", "n_gpus": ngpus, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "pairs_total": total, "results_ok": ok,
            "path_frac_of_hbm_roofline": BYTES_PER_FRAME * n * total * steps / dt / ngpus / 1e9 / HBM_PEAK_GBS,
            "workload": "BASELINE configs[3] in ONE process: asx_comm_create over %d plan(s), asx_xcorr_batch_multi_dev "
                        "(one host thread + stream per device, ncclAllGather of the result records inside the library)" % ngpus}


def capi_leg_guarded(ngpus, steps, in_process):
    """in-process with one GPU; as a child process with a time limit otherwise: the N > 1 form of this leg has never met a second
    GPU (no multi-GPU node was available to any round), and a fault in it must not take the headline line down with it"""
    try:
        if in_process:
            # RCCL prints a version banner on STDOUT when the library's communicator comes up; the contract is ONE JSON line
            # there, so file descriptor 1 points at stderr for the length of this leg (as around the process group's start-up)
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            try:
                return capi_leg(ngpus, steps)
            finally:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", "capi", "--gpus", str(ngpus), "--steps4", str(steps)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": "child exited with %d: %s" % (p.returncode, p.stderr[-500:])}
        return json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001 -- reported in the line, never fatal
        return {"error": "%s: %s" % (type(e).__name__, e)}


# --------------------------------------------------------------------------------------------------
# ranks
# --------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """N ranks as a child `python -m torch.distributed.run`; this process never touches the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """the rank / shard / gather code path on CPU (gloo), with fabricated results"""
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    graft.load()
    from audiosync_amd import sharding
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("ASX_BENCH_DRYRUN_FAIL") == "1" and rank == world - 1:
        raise RuntimeError("injected rank failure (tests/test_bench_cli.py)")
    ok = True
    report = {}
    for name, total in (("headline", 6 * world), ("config4", CONFIG4_BATCH)):
        start, count = sharding.shard_range(total, rank, world)
        width = (total + world - 1) // world
        buf, (lag, coef, ret) = sharding.result_buffer(width, "cpu")
        ids = torch.arange(start, start + count)
        lag[:count] = 3 * ids - 7
        coef[:count] = ids.double() / total
        ret[:count] = -(ids % 2).int()
        out, views = sharding.gather_result_buffers(buf, width)
        for r in range(world):
            s_r, c_r = sharding.shard_range(total, r, world)
            want = torch.arange(s_r, s_r + c_r)
            ok = ok and bool(torch.equal(views[r][0][:c_r], 3 * want - 7)) and bool(torch.equal(views[r][2][:c_r], -(want % 2).int()))
        report[name] = {"total": total, "shard": [start, count]}
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    # the per-rank record of the GPU run (run_rank: `per_rank`), with fabricated numbers: the same object gather
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "pairs": report["headline"]["shard"][1], "ms_per_step": 1.0 + rank,
                                      "kernel_ms_per_step": {k: 0.1 * (rank + 1) for k in FAMILIES}})
    if rank == 0:
        width4 = (CONFIG4_BATCH + world - 1) // world
        base = load_baseline_line(args.baseline_json)
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size_seen": dist.get_world_size(), "backend": "gloo",
                          "per_rank": per_rank, "baseline_seen": None if base is None else base.get("n_gpus"),
                          "results_ok": bool(t.item() == 1.0), "shards_rank0": report, "scaling": "weak",
                          "config4": {"scaling": "strong", "speedup_basis": CONFIG4_SPEEDUP_BASIS, "n_gpus": world,
                                      "pairs_total": CONFIG4_BATCH, "pairs_per_gpu": report["config4"]["shard"][1],
                                      "record_bytes_per_rank": sharding.result_bytes(width4)}}), flush=True)
    dist.destroy_process_group()
    return 0 if t.item() == 1.0 else 1


class Workload:
    """`count` pairs of sample_len frames on this rank (pair ids first_pair ...), results gathered over ranks"""

    def __init__(self, asx, sharding, torch, dist, dev, stream, n, total, world, rank, multi, noise_shift, split):
        self.torch, self.dist, self.sharding, self.multi, self.world, self.rank = torch, dist, sharding, multi, world, rank
        self.n, self.total = n, total
        self.start, self.count = sharding.shard_range(total, rank, world)
        self.width = (total + world - 1) // world          # every rank's result buffer holds `width` pairs
        self.stream = stream
        self.sh = stream.cuda_stream
        assert self.sh != 0, "an explicit (non-null) stream is handed to the library and to the collective"
        c = max(self.count, 1)
        self.d_src = torch.empty(c * 2 * n, dtype=torch.float32, device=dev)
        self.d_smp = torch.empty(c * n, dtype=torch.float32, device=dev)
        self.d_true = torch.full((self.width,), -1, dtype=torch.int64, device=dev)
        self.res_buf, (self.d_lag, self.d_coef, self.d_ret) = sharding.result_buffer(self.width, dev)
        self.gather_out = torch.empty((world, sharding.result_bytes(self.width)), dtype=torch.uint8, device=dev) if multi else None
        self.true_out = torch.empty((world, self.width), dtype=torch.int64, device=dev) if multi else None
        self.gathered = None
        if self.count:
            asx.synth_pairs_dev(SEED, self.start, self.count, n, noise_shift, self.d_src.data_ptr(),
                                self.d_smp.data_ptr(), self.d_true.data_ptr(), self.sh)
        self.plan = asx.Plan(n, c, torch.cuda.current_device(), split=split)

    def step(self):
        if self.count:
            self.plan.xcorr_batch_dev(self.d_src.data_ptr(), self.d_smp.data_ptr(), self.count, self.d_lag.data_ptr(),
                                      self.d_coef.data_ptr(), self.d_ret.data_ptr(), self.sh)
        if self.multi:
            # the only exchange on this path: RCCL all-gather of the 20 result bytes per pair of every shard,
            # on the same explicit stream, behind the kernels that produce them
            self.gathered = self.sharding.gather_result_buffers(self.res_buf, self.width, out=self.gather_out)

    def timed(self, steps, warmup, precondition=0):
        torch, dist = self.torch, self.dist
        # `precondition` extra untimed steps BEFORE the W warm-up steps: after any host synchronisation the chip needs
        # ~20 steps (~50 ms) to climb back to its working clock (k_rows 1.35 -> 1.10 ms, k_inv_cols 0.54 -> 0.40 ms over
        # the first steps behind a synchronisation; profiles/r3_bench.json, kernel_ms_series), far more than W = 5 steps
        for _ in range(precondition + warmup):
            self.step()
        torch.cuda.synchronize()
        if self.multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if self.multi:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.last_dt_local = dt      # this rank's own clock (the line reports the maximum over ranks; `per_rank` keeps each)
        if self.multi:
            t = torch.tensor([dt], dtype=torch.float64, device=self.d_lag.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def verify(self, oracle0=None):
        """one more step into cleared buffers; every rank's slice of what the gather delivered must equal that
        rank's planted delays (gathered separately), ret all zero.  The COEFFICIENT the timed path produced (the spectral
        Pearson form on real-column plans) is checked too: against the direct reduction of
        src/cross_correlation.c:74-116 on every pair of this rank (one more step with asx_plan_set_pearson(plan, 0), compared
        on the device, tolerance COEF_TOL = north_star's 1e-5) and, on rank 0, against the oracle's answer for pair 0 that the
        cpu_baseline leg left (`oracle0`).  -> (ok, {"coef_max_delta", "coef_oracle_delta", ...})"""
        torch, dist = self.torch, self.dist
        spectral = self.plan.layout == "real-column" and os.environ.get("ASX_PEARSON") != "direct"   # the same on every rank
        self.res_buf.zero_()
        if self.multi:
            self.gather_out.zero_()
        torch.cuda.synchronize()
        modes0 = self.plan.pearson_modes() if spectral else None
        self.step()
        torch.cuda.synchronize()
        ok = bool(torch.equal(self.d_lag[: self.count], self.d_true[: self.count])) and int(self.d_ret[: self.count].abs().sum()) == 0
        info = {"coef_tol": COEF_TOL, "coef_max_delta": None, "coef_oracle_delta": None, "coef_pairs_compared": 0}
        if spectral:   # this rank's pairs of ONE step by the form that produced their coefficient: [spectral, spectral + wrap-around part, direct]
            info["pearson_modes_per_step"] = [int(b - a) for a, b in zip(modes0, self.plan.pearson_modes())]
        coef_timed = self.d_coef[: self.count].clone()
        lag0, ret0 = (int(self.d_lag[0]), int(self.d_ret[0])) if self.count else (None, None)
        gathered_timed = self.gathered
        if spectral:
            gather_keep = self.gather_out.clone() if self.multi else None
            self.plan.set_pearson(False)
            self.step()                                   # (every rank: the step carries the gather)
            torch.cuda.synchronize()
            self.plan.set_pearson(True)
            if self.count:
                delta = float((coef_timed - self.d_coef[: self.count]).abs().max())
                finite = bool(torch.isfinite(coef_timed).all()) and bool(torch.isfinite(self.d_coef[: self.count]).all())
                info["coef_max_delta"] = delta
                info["coef_pairs_compared"] = self.count
                ok = ok and finite and delta < COEF_TOL
                ok = ok and bool(torch.equal(self.d_lag[: self.count], self.d_true[: self.count]))
                self.d_coef[: self.count].copy_(coef_timed)
            if self.multi:                                # what is checked below is what the TIMED form's step gathered
                self.gather_out.copy_(gather_keep)
                self.gathered = gathered_timed
        if oracle0 is not None and self.start == 0 and self.count:
            d = abs(float(coef_timed[0]) - oracle0["coefficient"])
            info["coef_oracle_delta"] = d
            ok = ok and oracle0["ret"] == ret0 and oracle0["lag"] == lag0 and d < COEF_TOL
        if self.multi:
            dist.all_gather_into_tensor(self.true_out.view(-1), self.d_true)
            torch.cuda.synchronize()
            views = self.gathered[1]
            ok = ok and len(views) == self.world
            for r in range(self.world):
                c_r = self.sharding.shard_range(self.total, r, self.world)[1]
                ok = ok and bool(torch.equal(views[r][0][:c_r], self.true_out[r][:c_r])) and int(views[r][2][:c_r].abs().sum()) == 0
            t = torch.tensor([1.0 if ok else 0.0], device=self.d_lag.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item() == 1.0)
            t = torch.tensor([-1.0 if info["coef_max_delta"] is None else info["coef_max_delta"], float(info["coef_pairs_compared"])],
                             dtype=torch.float64, device=self.d_lag.device)
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
            info["coef_max_delta"] = None if float(t[0]) < 0 else float(t[0])
            info["coef_pairs_compared"] = int(t[1])
        return ok, info

    def kernel_medians(self, steps, lead=0):
        """per-kernel HIP-event durations of `steps` CONSECUTIVE steps behind `lead` untimed ones (the profiling ring keeps
        the last `steps` calls): medians.  Called right behind the timed region, so the window is as pre-conditioned as
        the timed steps were."""
        torch = self.torch
        self.plan.set_profiling(steps)
        for _ in range(lead + steps):
            self.step()
        torch.cuda.synchronize()
        rows = [self.plan.last_timings_ms(b) for b in range(steps)]
        self.plan.set_profiling(0)
        return ({k: statistics.median(r[k] for r in rows) for k in FAMILIES},
                {k: [min(r[k] for r in rows), max(r[k] for r in rows)] for k in FAMILIES},
                {k: [round(r[k], 4) for r in reversed(rows)] for k in ("fwd_cols", "rows", "inv_cols")})

    def close(self):
        self.plan.close()
        del self.d_src, self.d_smp


def traffic_from_profiles(n, split, group, dom):
    """HBM bytes per launch of the dominant kernel from the committed PMC run (tools/traffic.sh: separate --pmc
    passes, FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes) -- NOT measured in this run;
    only returned when that run was this workload (sample_len, split, group)."""
    import glob
    import re
    def round_of(path):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        return int(m.group(1)) if m else 0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), key=round_of, reverse=True):
        if "experiments" in path:
            continue
        name = os.path.basename(path)
        try:
            tj = json.load(open(path))
        except (OSError, ValueError):
            continue
        c = tj.get("config", {})
        if c.get("sample_len") == n and c.get("split") == split and c.get("group") == group:
            for kname, kv in tj["kernels"].items():
                if kname.startswith("k_" + dom):
                    return kv["hbm_bytes_per_launch"], "profiles/%s (PMC run of %s, commit %s)" % (name, tj.get("date", "?"), tj.get("commit", "?"))
    return None, ("no PMC run on file for this configuration (sample_len %d, split %s, %d pairs per launch, kernel k_%s): "
                  "tools/traffic.sh collects one" % (n, split, group, dom))


def load_baseline_line(path):
    """--baseline-json: a file holding the JSON line of an earlier run of this bench (normally the 1-GPU one); None if absent"""
    if not path:
        return None
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1])
    except (OSError, ValueError, IndexError):
        return None


def run_rank(args, cpu=None):
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
        # RCCL prints a version banner on STDOUT when its communicator comes up; the contract is ONE JSON
        # line there, so file descriptor 1 points at stderr until the first collective has run.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            # a CPU-side group for waits that must not occupy the GPUs (the ranks idle while rank 0 runs the config4_capi leg:
            # inside an RCCL barrier they would spin in a kernel on the very devices that leg uses)
            cpu_group = dist.new_group(backend="gloo")
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    else:
        torch.cuda.set_device(0)
        cpu_group = None
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.Stream(dev)          # explicit: the library and the collective share THIS stream
    line = None
    with torch.cuda.stream(stream):
        n = args.sample_len
        batch = args.batch or max(8, min(4096, (2 << 30) // (12 * n)))   # ~2 GiB of inputs per GPU
        w = Workload(asx, sharding, torch, dist, dev, stream, n, batch * world, world, rank, multi, args.noise_shift, args.split)
        pre = max(0, args.precondition - args.warmup)
        # two clocks: the protocol exactly as the driver states it -- W warm-up steps, K timed steps, which
        # start inside the chip's clock ramp after idle -- and then the same K steps behind `pre` more untimed ones.
        # Rounds 1 and 2 reported the first form as `value`, round 3 and later the second.
        dt_cold = w.timed(args.steps, args.warmup, 0)
        dt = w.timed(args.steps, args.warmup, pre)
        prof_steps = args.profile_steps or max(20, args.steps)
        med, spread, series = w.kernel_medians(prof_steps, lead=10 if args.profile_steps == 0 else 0)
        dt_local = w.last_dt_local
        ok, coef_check = w.verify(cpu.get("oracle_pair0") if cpu else None)
        plan_group, plan_split, plan_threads, plan_layout = w.plan.group, w.plan.split, w.plan.threads, w.plan.layout
        overflows, w_count = w.plan.peak_overflows(), w.count
        spectral_modes = list(w.plan.pearson_modes()) if plan_layout == "real-column" and os.environ.get("ASX_PEARSON") != "direct" else None
        w.close()
        torch.cuda.empty_cache()

        cfg4 = None
        if not args.no_config4:
            w4 = Workload(asx, sharding, torch, dist, dev, stream, CONFIG4_N, CONFIG4_BATCH, world, rank, multi, 0, None)
            dt4 = w4.timed(args.steps4, 1)
            dt4_local = w4.last_dt_local
            ok4, coef_check4 = w4.verify()
            cfg4 = {"metric": "cross-correlations/sec (fixed batch %d x N=%d, strong scaling)" % (CONFIG4_BATCH, CONFIG4_N),
                    "value": CONFIG4_BATCH * args.steps4 / dt4, "unit": "cross-correlations/s", "scaling": "strong",
                    "speedup_basis": CONFIG4_SPEEDUP_BASIS,
                    "n_gpus": world, "steps": args.steps4, "ms_per_step": dt4 / args.steps4 * 1e3,
                    "pairs_total": CONFIG4_BATCH, "pairs_per_gpu": w4.count, "results_ok": ok4, "coef_check": coef_check4,
                    "path_frac_of_hbm_roofline": BYTES_PER_FRAME * CONFIG4_N * CONFIG4_BATCH * args.steps4 / dt4 / world / 1e9 / HBM_PEAK_GBS,
                    "workload": "BASELINE configs[3]: %d pairs, N=%d, SNR -6 dB, block-partitioned over %d rank(s), inputs "
                                "generated on the device per shard (%.1f GB per rank), result all-gather timed" %
                                (CONFIG4_BATCH, CONFIG4_N, world, 12.0 * CONFIG4_N * w4.count / 1e9)}
            w4.close()
            torch.cuda.empty_cache()

        # Every rank's own numbers, gathered on the CPU (gloo) behind the timed regions: should the first run on
        # more than one GPU fall short of >= 3.5x / >= 7x, the one JSON line says where -- a slow rank (its kernels), the result
        # gather (ms_per_step far above kernel_ms_sum on every rank) or the host thread (one rank's ms_per_step alone).
        per_rank = None
        if multi:
            mine = {"rank": rank, "device": torch.cuda.current_device(), "pairs": w_count,
                    "ms_per_step": dt_local / args.steps * 1e3,
                    "kernel_ms_per_step": {k: round(v, 4) for k, v in med.items()},
                    "kernel_ms_sum": round(sum(med[k] for k in ("fwd_cols", "rows", "inv_cols", "finalize", "pearson")), 4),
                    "config4_ms_per_step": (dt4_local / args.steps4 * 1e3) if not args.no_config4 else None}
            per_rank = [None] * dist.get_world_size()
            dist.all_gather_object(per_rank, mine, group=cpu_group)

        cfg4c = None
        if not args.no_config4 and not args.no_capi:
            if multi:
                torch.cuda.synchronize()
                dist.barrier(group=cpu_group)        # every rank has released its config4 buffers and its GPU is idle
            if rank == 0:
                cfg4c = capi_leg_guarded(world, args.steps4, in_process=(world == 1 and os.environ.get("ASX_BENCH_CAPI_CHILD") != "1"))
            if multi:
                dist.barrier(group=cpu_group)        # the others wait on the CPU, not in a kernel on the devices the leg uses

        single = None
        if world == 1 and not args.no_single:
            import numpy as np
            s32 = torch.empty(2 * n, dtype=torch.float32, device=dev); t32 = torch.empty(n, dtype=torch.float32, device=dev)
            asx.synth_pairs_dev(SEED, 0, 1, n, 1, s32.data_ptr(), t32.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            single = single_pair_latency(asx, torch, s32.cpu().numpy(), t32.cpu().numpy(), n, 10)

        if rank == 0:
            value = batch * world * args.steps / dt
            groups = (batch + plan_group - 1) // plan_group
            per_launch_pairs = min(batch, plan_group)
            # the dominant kernel = the longest one; the forward column kernel and the row kernel are within a few percent of
            # each other (which is ahead changes with the physical placement of a process's buffers), so among kernels within
            # 10 % of the longest the one with the LOWER roofline fraction is reported: the conservative reading
            longest = max(med[k] for k in ("fwd_cols", "rows", "inv_cols", "pearson"))
            shared = [k for k in ("fwd_cols", "rows", "inv_cols", "pearson") if ALGO_SHARE[k] > 0]
            near = [k for k in shared if med[k] >= 0.9 * longest]
            # (k_inv_cols has no algorithmic share -- Q is the intermediate; should it ever be the longest kernel with nothing
            # near it, the longest kernel that has a share is reported)
            dom = min(near, key=lambda k: ALGO_SHARE[k] / med[k]) if near else max(shared, key=lambda k: med[k])
            dom_launch_ms = med[dom] / groups
            dom_bytes = ALGO_SHARE[dom] * n * per_launch_pairs
            achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
            path_gbs = BYTES_PER_FRAME * n * value / world / 1e9   # per GPU, from the timed steps
            split = "%dx%dx%d" % plan_split
            traffic, traffic_src = traffic_from_profiles(n, split, plan_group, dom)
            ksum = sum(med[k] for k in ("fwd_cols", "rows", "inv_cols", "finalize", "pearson"))
            suffix = "_r" if plan_layout == "real-column" else ""
            per_kernel = {}
            for k in ("fwd_cols", "rows", "inv_cols", "pearson"):
                kb = ALGO_SHARE[k] * n * per_launch_pairs
                ka = kb / (med[k] / groups * 1e-3) / 1e9
                e = {"algorithmic_bytes_per_launch": kb, "avg_launch_ms": med[k] / groups, "achieved": ka, "frac": ka / HBM_PEAK_GBS}
                if k == "pearson" and spectral_modes is not None:
                    # the spectral form does not MOVE the 8 N bytes of the reference's reduction (it takes r[peak] and band sums the
                    # forward pass kept): their quotient by the family's time is work per second, not a bandwidth
                    e["frac"] = None
                    e["note"] = ("spectral Pearson form: pairs by form (spectral, + wrap-around correction, direct) = %s; "
                                 "`achieved` is algorithmic bytes per second, not bytes moved" % (spectral_modes,))
                per_kernel["k_" + k + (suffix if k != "pearson" else "_partial")] = e
            roofline = {
                "bound": "hbm", "kernel": "k_" + dom + (suffix if dom != "pearson" else "_partial"), "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_launch_ms,
                "launches_per_step": groups, "pairs_per_launch": per_launch_pairs,
                "path": {"algorithmic_bytes_per_pair": BYTES_PER_FRAME * n, "achieved": path_gbs,
                         "frac": path_gbs / HBM_PEAK_GBS,
                         "basis": "52*N bytes per pair x pairs/s per GPU over the timed steps"},
                "kernels": per_kernel,
                "kernel_ms_per_step": med, "kernel_ms_min_max": spread, "kernel_ms_sum": ksum,
                "kernel_ms_series": {k: v[:40] for k, v in series.items()},
                "kernel_ms_basis": "HIP events on the launch stream, median over %d consecutive steps right behind the timed "
                                   "region and 10 more untimed steps (the same pre-conditioned window; the only host "
                                   "synchronisation inside it is the library's own look at its overflow list, once per step)"
                                   % prof_steps,
            }
            line = {
                "metric": "cross-correlations/sec (N=%d float32 pairs)" % n,
                "value": value, "unit": "cross-correlations/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "preconditioning_steps": pre,
                # every untimed step that ran in front of the region `value` times: the W + K steps of the un-preconditioned leg
                # (`value_no_precondition`, taken first), then `pre` + W more
                "untimed_steps_before_value": (args.warmup + args.steps) + pre + args.warmup,
                "ms_per_step": dt / args.steps * 1e3,
                "value_no_precondition": batch * world * args.steps / dt_cold,
                "ms_per_step_no_precondition": dt_cold / args.steps * 1e3,
                "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "batched xcorr, N=%d frames/sample (source 2N), %d pairs per GPU per step, "
                                       "48 kHz mono float32, planted delays, SNR 0 dB" % (n, batch),
                           "sample_len": n, "pairs_per_gpu": batch, "group": plan_group,
                           "split": split, "threads_cols_rows": list(plan_threads), "layout": plan_layout,
                           "pearson": ("spectral form with a per-pair error bound <= 1e-5, direct reduction otherwise "
                                       "(include/audiosync/xcorr_hip.h: asx_plan_set_pearson)" if spectral_modes is not None
                                       else "direct reduction over the segments"),
                           "exact": "every pair's lag is the float64 argmax by construction: overflowed near-tie lists are "
                                    "looked at again behind the last launch group (one host synchronisation per step, inside "
                                    "the timed region)",
                           "parallelism": "pairs sharded over %d GPU(s), one process per GPU, RCCL all_gather of results" % world},
                "world_size_seen": dist.get_world_size() if multi else 1,
                "results_ok": ok, "coef_check": coef_check, "pearson_modes": spectral_modes, "peak_overflows": overflows,
                "roofline": roofline,
            }
            if per_rank is not None:
                line["per_rank"] = per_rank
            base = load_baseline_line(args.baseline_json)
            if base is not None:
                # weak scaling: N times the pairs in (ideally) the same time; strong scaling: the same 8192 pairs in 1/N of it
                line["speedup_vs"] = {"baseline_n_gpus": base.get("n_gpus"), "value": value / base["value"] if base.get("value") else None}
                if cfg4 is not None and base.get("config4", {}).get("value"):
                    cfg4["speedup_vs"] = {"baseline_n_gpus": base.get("n_gpus"), "value": cfg4["value"] / base["config4"]["value"]}
            if cfg4 is not None:
                line["config4"] = cfg4
            if cfg4c is not None:
                line["config4_capi"] = cfg4c
            if single is not None:
                line["single_pair"] = single
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precondition", type=int, default=30,
                    help="untimed steps before the timed region in all (the W warm-up steps included): clock ramp after idle")
    ap.add_argument("--sample-len", type=int, default=1440000)
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = auto)")
    ap.add_argument("--noise-shift", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-config4", action="store_true", help="skip the fixed-batch 8192 x 480000 leg")
    ap.add_argument("--no-single", action="store_true", help="skip the single-pair latency leg")
    ap.add_argument("--no-capi", action="store_true", help="skip the C-ABI multi-device form of the config4 leg")
    ap.add_argument("--steps4", type=int, default=3, help="timed steps of the config4 leg")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="steps of the per-kernel HIP-event window (0 = max(20, steps) behind 10 lead steps); counter-collection "
                         "runs (tools/traffic.sh, tools/pmc.sh), where every launch is serialised, pass a small number")
    ap.add_argument("--split", default=None)
    ap.add_argument("--dry-run", action="store_true", help="rank/shard/gather path on CPU with gloo, no GPU")
    ap.add_argument("--baseline-json", default=None,
                    help="file with the JSON line of an earlier run (normally --gpus 1): adds `speedup_vs` to the line and to config4")
    ap.add_argument("--mode", default="batched", choices=["batched", "streaming", "single", "capi"],
                    help="batched = the headline workload (default); streaming = BASELINE config 5 "
                         "(growing window 3..30 s, plan reuse); single = config 2 (one pair, latency)")
    args = ap.parse_args()
    if args.mode == "capi":      # child of rank 0 (capi_leg_guarded): config4 through asx_comm_create / asx_xcorr_batch_multi_dev
        print(json.dumps(capi_leg(args.gpus, args.steps4)), flush=True)
        return 0
    if args.mode != "batched":
        return side_mode(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, sys.argv[1:])
    if args.dry_run:
        return dry_run(args)
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu:
        # rank 0 at N = 1 only, and FIRST: forked worker processes and a clean machine, before any GPU call
        cpu = cpu_baseline(args.sample_len)
        # the same leg also leaves the checker's answer for pair 0 of the bench workload (the oracle on the inputs k_synth
        # generates on the device, about a second); Workload.verify() compares the timed path's lag AND coefficient with it
        cpu["oracle_pair0"] = oracle_pair0(args.sample_len, args.noise_shift)
    return run_rank(args, cpu)


if __name__ == "__main__":
    sys.exit(main() or 0)
