"""The CPython module `audiosync` of this build: same behaviour the reference's own binding test
checks (tests/test_bindings.py:11-54 there: debug round trip, setup(), run() on a thread with
pause -> 'paused', resume -> 'running', abort -> 'idle', a second run/abort), plus the two
additions (cross_correlation, set_feed)."""
import os
import sys
import threading
import time

import numpy as np
import pytest

import oracle
from util import graft

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def audiosync():
    graft.build()
    sys.path.insert(0, graft.PKG_DIR)
    import torch  # noqa: F401  (one HIP runtime per process: torch's, loaded first)
    import audiosync as mod
    return mod


def test_surface_matches_reference_module(audiosync):
    for name in ("run", "pause", "resume", "abort", "status", "setup", "get_debug", "set_debug"):
        assert callable(getattr(audiosync, name)), name


def test_debug_round_trip_and_setup(audiosync):
    assert not audiosync.get_debug()
    audiosync.set_debug(True)
    assert audiosync.get_debug()
    audiosync.set_debug(False)
    assert not audiosync.get_debug()
    assert audiosync.setup("test") is False      # no PulseAudio in this build
    assert audiosync.status() == "idle"


def test_cross_correlation_from_python(audiosync):
    n = 48000
    src, smp, true_lag = oracle.synth_pair(77, 1, n, 1)
    ret, lag, coef = audiosync.cross_correlation(src.astype(np.float64), smp.astype(np.float64))
    o_ret, o_lag, o_coef = oracle.cross_correlation(src, smp)
    assert (ret, lag) == (o_ret, o_lag) and lag == true_lag and abs(coef - o_coef) < 1e-5
    with pytest.raises(ValueError):
        audiosync.cross_correlation(np.zeros(10), np.zeros(4))
    with pytest.raises(TypeError):
        audiosync.cross_correlation(np.zeros(10, dtype=np.int32), np.zeros(5, dtype=np.int32))
    with pytest.raises(TypeError):
        audiosync.cross_correlation(np.zeros(10, dtype=np.float64), np.zeros(5, dtype=np.float32))
    # float32 buffers: the batched float32 path with one pair
    ret32, lag32, coef32 = audiosync.cross_correlation(src, smp)
    assert (ret32, lag32) == (o_ret, o_lag) and abs(coef32 - o_coef) < 1e-5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cross_correlation_batch_from_python(audiosync, dtype):
    """the batched variant over the buffer protocol, against the oracle pair by pair"""
    n, batch = 24000, 5
    pairs = [oracle.synth_pair(78, p, n, 0) for p in range(batch)]
    src = np.stack([p[0] for p in pairs]).astype(dtype)
    smp = np.stack([p[1] for p in pairs]).astype(dtype)
    smp[3] = 0                                     # one all-zero sample: ret -1, NaN coefficient
    rets, lags, coefs = audiosync.cross_correlation_batch(src, smp)
    assert len(rets) == len(lags) == len(coefs) == batch
    for b in range(batch):
        o_ret, o_lag, o_coef = oracle.cross_correlation(src[b], smp[b])
        assert (rets[b], lags[b]) == (o_ret, o_lag), b
        if o_ret == 0:
            assert abs(coefs[b] - o_coef) < 1e-5
        else:
            assert coefs[b] != coefs[b]
    # flat buffers with an explicit batch, and the shape errors
    r2, l2, c2 = audiosync.cross_correlation_batch(src.reshape(-1), smp.reshape(-1), batch)
    assert l2 == lags
    with pytest.raises(ValueError):
        audiosync.cross_correlation_batch(src.reshape(-1), smp.reshape(-1), 7)
    with pytest.raises(ValueError):
        audiosync.cross_correlation_batch(src[:, :-2].copy(), smp)


def test_run_pause_resume_abort_state_machine(audiosync):
    # a slow feed (frames_per_ms) keeps the run alive long enough to poke the state machine
    rng = np.random.default_rng(0)
    source = rng.uniform(-1, 1, 2 * 30 * 48000)
    sample = 0.5 * source[1000: 1000 + 30 * 48000] + 0.01 * rng.uniform(-1, 1, 30 * 48000)
    audiosync.set_feed(source, sample, 200)      # 200 frames/ms: the first interval needs > 1 s
    result = {}
    th = threading.Thread(target=lambda: result.update(r=audiosync.run("")))
    th.start()
    time.sleep(0.2)
    audiosync.pause()
    assert audiosync.status() == "paused"
    audiosync.resume()
    assert audiosync.status() == "running"
    audiosync.abort()
    th.join(timeout=60)
    assert not th.is_alive()
    assert audiosync.status() == "idle"
    assert result["r"][1] is False               # aborted: (lag, False)
    # second run, to completion this time: the planted delay in milliseconds
    audiosync.set_feed(source, sample, 0)
    lag_ms, ok = audiosync.run("again")
    assert ok is True and lag_ms == round(1000 * 1000.0 / 48000.0)
    assert audiosync.status() == "idle"


def test_run_from_f64le_files_and_a_fifo(audiosync, tmp_path):
    """the wire format of the reference's producers (`ffmpeg -f f64le`, src/ffmpeg_pipe.c:68-81): run() reads the two
    tracks from files; a short sample file is zero-filled (src/ffmpeg_pipe.c:139-149); a FIFO written by another
    thread works like the ffmpeg pipe; a missing file aborts the run"""
    rng = np.random.default_rng(5)
    source = rng.uniform(-1, 1, 2 * 30 * 48000)
    delay = 2400                                                    # 50 ms
    sample = 0.5 * source[delay: delay + 30 * 48000] + 0.01 * rng.uniform(-1, 1, 30 * 48000)
    fsrc, fsmp = tmp_path / "source.f64le", tmp_path / "sample.f64le"
    source.astype("<f8").tofile(fsrc)
    sample.astype("<f8").tofile(fsmp)
    audiosync.set_feed_files(str(fsrc), str(fsmp))
    lag_ms, ok = audiosync.run("files")
    assert ok is True and lag_ms == 50
    # 4 s of sample only: the first interval (3 s) still resolves the delay
    sample[: 4 * 48000].astype("<f8").tofile(fsmp)
    audiosync.set_feed_files(str(fsrc), str(fsmp))
    lag_ms, ok = audiosync.run("short")
    assert ok is True and lag_ms == 50
    # FIFO: a writer thread plays the role of the ffmpeg child
    fifo = tmp_path / "sample.fifo"
    os.mkfifo(fifo)
    def writer():
        try:
            with open(fifo, "wb") as f:
                f.write(sample.astype("<f8").tobytes())
        except BrokenPipeError:
            pass    # the run found the delay in the first interval and closed its end, like SIGKILL to ffmpeg
    th = threading.Thread(target=writer)
    th.start()
    audiosync.set_feed_files(str(fsrc), str(fifo))
    lag_ms, ok = audiosync.run("fifo")
    assert ok is True and lag_ms == 50
    th.join(timeout=30)
    assert not th.is_alive()
    # a file that does not exist: like a failed ffmpeg child, the job is aborted
    audiosync.set_feed_files(str(tmp_path / "missing.f64le"), str(fsmp))
    lag_ms, ok = audiosync.run("missing")
    assert ok is False and audiosync.status() == "idle"


def test_fifo_whose_writer_never_starts_can_be_aborted(audiosync, tmp_path):
    """open() of a FIFO used to block until a writer appeared and read() between chunks, so abort() was
    never seen and run() hung in pthread_join.  The producers now open O_NONBLOCK and poll() with a timeout, the status
    checked every round (host/audiosync.c read_chunk)."""
    rng = np.random.default_rng(6)
    source = rng.uniform(-1, 1, 2 * 30 * 48000)
    fsrc = tmp_path / "source.f64le"
    source.astype("<f8").tofile(fsrc)
    fifo = tmp_path / "nobody_writes.fifo"
    os.mkfifo(fifo)
    audiosync.set_feed_files(str(fsrc), str(fifo))
    result = {}
    th = threading.Thread(target=lambda: result.update(r=audiosync.run("stalled")))
    th.start()
    time.sleep(0.5)
    assert audiosync.status() == "running"
    with pytest.raises(RuntimeError):
        audiosync.set_feed_files(str(fsrc), str(fsrc))      # refused while the run is in progress
    audiosync.abort()
    th.join(timeout=20)
    assert not th.is_alive()
    assert result["r"][1] is False and audiosync.status() == "idle"
