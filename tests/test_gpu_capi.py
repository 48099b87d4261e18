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
