"""The reference's C API exercised from C programs (tests/c/), on the GPU."""
import os
import subprocess

import pytest

from util import ROOT, graft

pytestmark = pytest.mark.gpu
CDIR = os.path.join(ROOT, "tests", "c")


@pytest.fixture(scope="module")
def built():
    graft.build()
    subprocess.check_call(["make", "-s", "-C", CDIR])
    return CDIR


def test_reference_known_answers_from_c(built):
    out = subprocess.run([os.path.join(built, "kat_runner"), os.path.join(ROOT, "tests", "golden", "reference_kat.txt")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "12 cases, 0 failures" in out.stdout


@pytest.mark.parametrize("delay", [12345, -4321, 100000])
def test_audiosync_run_with_memory_feed(built, delay):
    # clean tracks: the first interval (3 s) already reaches MIN_CONFIDENCE (src/audiosync.c:254-258)
    out = subprocess.run([os.path.join(built, "run_feed"), str(delay), "0.01"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    ret, lag_ms, status = out.stdout.split()
    assert int(ret) == 0 and status == "idle"
    assert int(lag_ms) == round(delay * 1000.0 / 48000.0)


def test_audiosync_run_gives_up_on_noise(built):
    # coefficient never reaches 0.95: all six intervals run, ret stays -1 (src/audiosync.c:166-284)
    out = subprocess.run([os.path.join(built, "run_feed"), "5000", "2.0"], capture_output=True, text=True, timeout=600)
    ret, lag_ms, status = out.stdout.split()
    assert int(ret) == -1 and status == "idle"


def test_bench_prints_exactly_one_json_line_on_stdout():
    """the driver's contract: `python bench.py --gpus 1 --steps K --warmup W` -> ONE JSON line on stdout, whatever the libraries
    underneath print when they come up (RCCL's version banner of the config4_capi leg went to stdout once, round 5)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--precondition", "2",
                        "--no-cpu", "--steps4", "1"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:-1]
    d = json.loads(lines[0])
    assert d["results_ok"] and d["config4"]["results_ok"] and d["config4_capi"].get("results_ok") is True, d.get("config4_capi")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert "cpu_baseline" not in d          # --no-cpu
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


def test_bench_rank_path_with_the_capi_leg_as_a_child_process():
    """the N > 1 form of bench.py on the one GPU there is: a rank under torch.distributed.run (ASX_BENCH_FORCE_DIST=1: process
    group, RCCL gather, the gloo group the idle ranks wait in) whose config4_capi leg runs as a CHILD process with a time limit
    (ASX_BENCH_CAPI_CHILD=1), as it does with more than one GPU; still ONE JSON line on stdout"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, ASX_BENCH_FORCE_DIST="1", ASX_BENCH_CAPI_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--precondition", "1", "--no-cpu", "--steps4", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:-1]
    d = json.loads(lines[0])
    assert d["results_ok"] and d["world_size_seen"] == 1 and d["config4"]["results_ok"]
    assert d["config4_capi"].get("results_ok") is True and d["config4_capi"]["n_gpus"] == 1, d["config4_capi"]
    # round 6: the rank's own record (gathered on the CPU with gloo) and the coefficient check of the timed path
    pr = d["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["pairs"] > 0 and pr[0]["ms_per_step"] > 0
    assert set(pr[0]["kernel_ms_per_step"]) >= {"fwd_cols", "rows", "inv_cols", "pearson"} and pr[0]["config4_ms_per_step"] > 0
    cc = d["coef_check"]
    assert cc["coef_pairs_compared"] == pr[0]["pairs"] and cc["coef_max_delta"] < cc["coef_tol"], cc
    assert d["config4"]["coef_check"]["coef_pairs_compared"] == 8192
