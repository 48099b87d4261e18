"""Host planning arithmetic (csrc/plan_math.cpp) against the NumPy model of the device
algorithm (tests/model_fourstep.py): transform length, split, radix schedule, digit-reversal
tables, twiddle tables.  No GPU needed."""
import numpy as np
import pytest

import model_fourstep as model
from util import asx

PRODUCTION = [144000, 288000, 480000, 720000, 960000, 1440000]


@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 7, 11, 13, 100, 1000, 1001, 4096, 48000, 12345] + PRODUCTION)
def test_length_and_split(n):
    d = asx().planmath_describe(n)
    F, valid = model.embed_params(n, model.next_smooth_even)
    assert d["F"] == F and d["src_valid"] == valid
    M = F // 2
    assert d["M1"] * d["M2"] == M
    assert d["T"] >= 1 and d["T"] & (d["T"] - 1) == 0 and d["T"] <= 64
    # LDS: a column tile; a row block = four rows of the packed-sample kernel, or one row of each spectrum of the
    # real-column kernel (the 2400-point rows of the two longest reference lengths, csrc/rlayout.hip)
    assert d["M1"] * d["T"] * 8 <= 80 * 1024
    assert 4 * d["M2"] * 8 <= 64 * 1024 or (n in (960000, 1440000) and 2 * d["M2"] * 8 <= 64 * 1024)
    assert int(np.prod(d["radix1"], dtype=np.int64)) == d["M1"]
    assert int(np.prod(d["radix2"], dtype=np.int64)) == d["M2"]
    assert set(d["radix1"] + d["radix2"]) <= {2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16}


@pytest.mark.parametrize("n", PRODUCTION)
def test_production_lengths_are_not_embedded_and_use_wide_tiles(n):
    d = asx().planmath_describe(n)
    assert d["F"] == 2 * n          # true circular correlation of length 2N
    assert d["T"] >= 16             # sixteen real columns per tile: 64-byte input pieces, whole lines of the intermediates
    assert d["M1"] % 2 == 0 and d["M2"] % d["T"] == 0 and d["M2"] in (480, 1200, 2400)   # what csrc/rlayout.hip needs


@pytest.mark.parametrize("n", [6, 30, 1000, 48000, 144000])
def test_position_tables_match_model(n):
    mod = asx()
    d = mod.planmath_describe(n)
    pos1 = mod.planmath_table(n, 0)
    inv1 = mod.planmath_table(n, 1)
    pos2 = mod.planmath_table(n, 2)
    assert np.array_equal(pos1, model.position_table(d["M1"], d["radix1"]))
    assert np.array_equal(pos2, model.position_table(d["M2"], d["radix2"]))
    assert np.array_equal(inv1[pos1], np.arange(d["M1"]))
    assert sorted(pos2.tolist()) == list(range(d["M2"]))


def test_split_override_and_rejects():
    mod = asx()
    d = mod.planmath_describe(1000, "25x40x8")
    assert (d["M1"], d["M2"], d["T"]) == (25, 40, 8)
    for bad in ("25x41x8", "25x40x3", "0x1000x8", "nonsense"):
        with pytest.raises(mod.AsxError):
            mod.planmath_describe(1000, bad)


@pytest.mark.parametrize("n", [1000, 144000, 1440000])
def test_twiddle_tables(n):
    mod = asx()
    d = mod.planmath_describe(n)
    F, M1, M2 = d["F"], d["M1"], d["M2"]
    tw1 = mod.planmath_twiddles(n, 0)
    tw2 = mod.planmath_twiddles(n, 1)
    lo = mod.planmath_twiddles(n, 2)
    hi = mod.planmath_twiddles(n, 3)
    assert np.abs(tw1 - model.tw(M1, np.arange(M1))).max() < 1e-7
    assert np.abs(tw2 - model.tw(M2, np.arange(M2))).max() < 1e-7
    # the two-level table reproduces w_F^p for random p < F to float32 accuracy
    rng = np.random.default_rng(0)
    p = rng.integers(0, F, 4096)
    got = lo[p & 2047].astype(np.complex128) * hi[p >> 11].astype(np.complex128)
    assert np.abs(got - model.tw(F, p)).max() < 2.5e-7


def test_inplace_stage_model_round_trip():
    # the in-place DIF / DIT recipe the kernels use (lds_fft.h), checked in float64
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 12, 60, 360):
        rad = model.radix_list(n)
        x = rng.normal(size=n) + 1j * rng.normal(size=n)
        y = model.dif_forward_inplace(x, rad)
        pos = model.position_table(n, rad)
        assert np.abs(y[pos] - np.fft.fft(x)).max() < 1e-10
        assert np.abs(model.dit_inverse_inplace(y, rad) - n * x).max() < 1e-9


def test_unsupported_lengths_are_rejected_with_a_message():
    mod = asx()
    for bad in (0, (1 << 27) + 1):
        with pytest.raises(mod.AsxError):
            mod.planmath_describe(bad)
    # a huge prime sample_len embeds into a smooth length that must still split into LDS-sized factors
    d = mod.planmath_describe(3999971)
    assert d["F"] >= 3 * 3999971 - 1 and d["M1"] * d["M2"] * 2 == d["F"]


@pytest.mark.parametrize("n", [45, 1000, 44100, 96000, 1000000, 1440000])
def test_measured_mode_candidates_are_valid_splits(n):
    mod = asx()
    base = mod.planmath_describe(n)
    cands = mod.planmath_candidates(n, 16)
    assert 1 <= len(cands) <= 16 and len(set(cands)) == len(cands)
    for c in cands:
        d = mod.planmath_describe(n, c)          # every candidate must be accepted as an explicit split
        m1, m2, t = (int(x) for x in c.split("x"))
        assert (d["M1"], d["M2"], d["T"]) == (m1, m2, t)
        assert m1 * m2 * 2 == base["F"] and d["F"] == base["F"]
        assert t >= 2 and (t & (t - 1)) == 0 and m1 * t * 8 <= 80 * 1024 and 4 * m2 * 8 <= 64 * 1024


@pytest.mark.parametrize("m1,m2", [(4, 8), (6, 4), (10, 12), (12, 20), (30, 16)])
def test_real_column_decomposition_matches_the_reference_recipe(m1, m2):
    """csrc/rlayout.hip's algebra (model_fourstep.rlayout_*): r2c columns with the untangling inside the tile, rows
    k1 = 0..M1 independent, c2r columns -- equals rfft * conj(rfft) -> irfft of src/cross_correlation.c:204-239"""
    n = m1 * m2
    rng = np.random.default_rng(m1 * 100 + m2)
    src = rng.standard_normal(2 * n)
    smp = rng.standard_normal(n)
    r = model.rlayout_xcorr(src, smp, m1, m2)
    ref = model.reference_r(src, smp)
    assert np.abs(r - ref).max() < 1e-9 * np.abs(ref).max()


def test_spectral_pearson_algebra_matches_the_reference_reduction():
    """csrc/pearson_spectral.hip's identity, in float64: r[peak] (minus the products that did not wrap around, for a negative
    lag) and band + edge window sums give pearson_coefficient() of the segments src/cross_correlation.c:256-271 selects --
    for lags inside a band, on band borders, with windows that hold no whole band, and with offsets in both tracks."""
    import oracle
    M2, T, rows = 48, 16, 40                  # a [40][48] sample matrix for the sample, [80][48] for the source; bands of 8 rows
    N = rows * M2
    gs = 8 * M2
    rng = np.random.default_rng(17)
    source = rng.normal(size=2 * N) + 0.3
    sample = rng.normal(size=N) - 0.2
    peaks = [0, 1, 5, gs - 1, gs, gs + 1, 3 * gs, N - 1, N - gs, N, N + 1, N + 7, N + gs, 2 * N - gs, 2 * N - gs - 3, 2 * N - 5,
             2 * N - 1, N + N // 2]
    for peak in peaks:
        lag, coef, mode = model.spectral_pearson(source, sample, peak, M2, T)
        if peak >= N:
            want_lag = (peak % N) - N
            a, b = source[: N + want_lag], sample[-want_lag:]
        else:
            want_lag = peak
            a, b = source[peak: peak + N], sample
        assert lag == want_lag
        if len(a) < 2:
            continue
        want = oracle.pearson_coefficient(a.copy(), b.copy())
        assert abs(coef - want) < 1e-10, (peak, mode, coef, want)
