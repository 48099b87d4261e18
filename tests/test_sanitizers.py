"""The host C code and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer, on the CPU
(the reference's Debug build is -fsanitize=undefined,address, CMakeLists.txt:15-16; GPU sanitizers are not available on
this pool).  host/cross_correlation.c and host/audiosync.c are linked against tests/c/asx_stub.c, a test-only CPU stand-in
for the asx_* entry points (the product itself has no CPU path: tests/test_abi.py)."""
import os
import subprocess

import pytest

from util import ROOT

CDIR = os.path.join(ROOT, "tests", "c")
KAT = os.path.join(ROOT, "tests", "golden", "reference_kat.txt")


@pytest.fixture(scope="module")
def san():
    subprocess.check_call(["make", "-s", "-C", CDIR, "sanitizers"])
    return os.path.join(CDIR, "san")


def run(cmd, **env):
    e = dict(os.environ)
    e.update({"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
              "TSAN_OPTIONS": "halt_on_error=1"})
    e.update(env)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_reference_known_answers_through_the_host_code_under_asan(san):
    assert "12 cases, 0 failures" in run([os.path.join(san, "kat_asan"), KAT])


def test_oracle_known_answers_under_asan(san):
    assert "12 cases, 0 failures" in run([os.path.join(san, "oracle_kat_asan"), KAT])


@pytest.mark.parametrize("binary", ["stress_asan", "stress_tsan"])
def test_plan_cache_under_contention(san, binary):
    """9 sample lengths on 8 cache slots from 8 threads while a ninth thread drops every cached plan again and again:
    the pin / doom reference counts of host/cross_correlation.c:46-118 (missing 6)"""
    assert ", 0 failures" in run([os.path.join(san, binary), "40"])


@pytest.mark.parametrize("binary", ["abort_asan", "abort_tsan"])
def test_producers_and_state_machine(san, binary, tmp_path):
    """pause / resume / abort of a slow memory feed, abort of a run whose FIFO writer never starts, feed setters refused
    during a run (host/audiosync.c)"""
    assert "0 failures" in run([os.path.join(san, binary), str(tmp_path)])


@pytest.mark.parametrize("binary", ["shard_tsan", "shard_asan"])
def test_sharded_batch_driver_without_hardware(san, binary):
    """csrc/shard_driver.cpp -- the thread / partition / record-layout logic of asx_xcorr_batch_multi_dev,
    which has never met a second GPU -- on host-memory stand-ins for the device operations: 1, 2, 3 and 8 shards, uneven
    counts, a failing shard (error returned, every thread joined, every stream drained before the call returns), a failing
    record allocation (no stale width), width changes"""
    out = run([os.path.join(san, binary)])
    assert ", 0 failures" in out and not out.startswith("0 cases")


@pytest.mark.parametrize("binary", ["narrow_tsan", "narrow_asan"])
def test_float32_exactness_check_of_the_double_abi(san, binary):
    """csrc/host_narrow.cpp -- the worker pool that converts the reference's double buffers to float32 while checking that
    nothing is lost (then 4 bytes per frame cross PCIe instead of 8) -- exact and inexact buffers, NaNs, ragged sizes,
    four concurrent callers, and (AddressSanitizer build only: ThreadSanitizer does not follow the fork of a threaded process) a
    forked child, where the pool's threads do not exist and the caller converts every chunk itself"""
    out = run([os.path.join(san, binary)] + (["nofork"] if binary.endswith("tsan") else []))
    assert ", 0 failures" in out and (binary.endswith("tsan") or out.startswith("50 cases"))
