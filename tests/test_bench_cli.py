"""bench.py's N>1 launcher on CPU: `--gpus 2` must start two ranks (one process each) that form a
process group, shard both workloads and gather every shard's results -- here with gloo and fabricated
results (`--dry-run`), the same code path the GPU run takes with RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_launches_two_ranks():
    p, line = run_bench("--gpus", "2", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None, p.stdout[-2000:] + p.stderr[-2000:]
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["world_size_seen"] == 2
    assert line["results_ok"] is True
    assert line["shards_rank0"]["config4"] == {"total": 8192, "shard": [0, 4096]}


def test_gpus_3_uneven_shards():
    p, line = run_bench("--gpus", "3", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 3 and line["results_ok"] is True
    assert line["shards_rank0"]["config4"] == {"total": 8192, "shard": [0, 2731]}


def test_per_rank_records_and_a_baseline_line(tmp_path):
    """the N > 1 line carries every rank's own clock and kernel medians (gathered on the CPU with gloo, the object
    gather the GPU run uses), and --baseline-json reads an earlier run's line for `speedup_vs`"""
    base = tmp_path / "one_gpu.json"
    base.write_text("noise before the line\n" + json.dumps({"n_gpus": 1, "value": 50000.0, "config4": {"value": 160000.0}}) + "\n")
    p, line = run_bench("--gpus", "3", "--dry-run", "--baseline-json", str(base))
    assert p.returncode == 0, p.stderr[-2000:]
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1, 2]
    assert [r["ms_per_step"] for r in pr] == [1.0, 2.0, 3.0]
    assert all(set(r["kernel_ms_per_step"]) >= {"fwd_cols", "rows", "inv_cols", "pearson"} for r in pr)
    assert sum(r["pairs"] for r in pr) == 18
    assert line["baseline_seen"] == 1
    p, line = run_bench("--gpus", "2", "--dry-run", "--baseline-json", str(tmp_path / "missing.json"))
    assert p.returncode == 0 and line["baseline_seen"] is None


def test_a_failing_rank_fails_the_launcher():
    # a rank that cannot form the group (bad backend request) must surface as a non-zero exit code
    p, line = run_bench("--gpus", "2", "--dry-run", env_extra={"ASX_BENCH_DRYRUN_FAIL": "1"})
    assert p.returncode != 0


def test_gpus_8_the_node_size_of_baseline_config_4():
    """BASELINE.json configs[3]: 8192 pairs over the 8 GPUs of a node.  No 8-GPU node was available to rounds 1-4; this is
    the rank / shard / gather path at that world size on CPU (gloo), and the shape of the strong-scaling record a future
    SCALE run needs: config4.value per n_gpus with its speed-up basis stated."""
    p, line = run_bench("--gpus", "8", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["world_size_seen"] == 8 and line["results_ok"] is True
    assert line["shards_rank0"]["config4"] == {"total": 8192, "shard": [0, 1024]}
    assert line["shards_rank0"]["headline"] == {"total": 48, "shard": [0, 6]}
    c4 = line["config4"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 8 and c4["pairs_per_gpu"] == 1024 and c4["pairs_total"] == 8192
    assert "n_gpus = 1" in c4["speedup_basis"] and c4["record_bytes_per_rank"] == 1024 * 20
    assert line["scaling"] == "weak"
