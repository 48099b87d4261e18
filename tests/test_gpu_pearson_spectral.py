"""The spectral form of the Pearson coefficient (csrc/pearson_spectral.hip).

Reference: pearson_coefficient() over the two segments the lag selects (src/cross_correlation.c:74-116, :256-276).
The float32 batched entry points of a real-column plan take the coefficient from r[peak] and window sums instead of
reading the inputs again; a pair whose error bound is not below 1e-5 takes the direct reduction by itself.  Either way
the contract is north_star's: |coefficient - reference's| < 1e-5, same lag, same ret."""
import numpy as np
import pytest

import oracle
from util import asx

pytestmark = pytest.mark.gpu

COEF_TOL = 1e-5      # north_star
FAST, CORR, DIRECT = 0, 1, 2


@pytest.fixture(scope="module")
def mod():
    m = asx()
    assert m.device_count() >= 1, "no MI355X visible"
    return m


def planted(rng, n, lag, gain=0.5, noise=0.25, sign=1.0):
    """source = uniform noise; sample[i] = sign * gain * source[i + lag] (where that exists) + noise"""
    big = rng.uniform(-1, 1, 4 * n)
    src = big[n: 3 * n]
    smp = sign * gain * big[n + lag: 2 * n + lag] + noise * rng.uniform(-1, 1, n)
    return src.astype(np.float32), smp.astype(np.float32)


def run_both(mod, n, srcs, smps):
    src, smp = np.stack(srcs), np.stack(smps)
    with mod.Plan(n, len(srcs), 0) as plan:
        assert plan.layout == "real-column"
        lag_s, coef_s, ret_s = plan.xcorr_batch_f32(src, smp)
        modes = plan.pearson_modes()
        plan.set_pearson(False)
        lag_d, coef_d, ret_d = plan.xcorr_batch_f32(src, smp)
        assert plan.pearson_modes() == modes        # the direct setting does not count
    return (lag_s, coef_s, ret_s), (lag_d, coef_d, ret_d), modes


@pytest.mark.parametrize("n", [144000, 288000, 480000, 720000, 960000, 1440000])
def test_spectral_form_matches_the_oracle_and_the_direct_form(mod, n):
    """positive lags (cross term = r[peak]), small negative lags (minus the wrap-around part), large negative lags (short
    segment: direct), a lag inside the first band, a lag that ends the window exactly on a band border, anti-correlation"""
    rng = np.random.default_rng(n)
    with mod.Plan(n, 1, 0) as probe:
        m1, m2, _ = probe.split
    band = 40 * m2   # a multiple of every band height in use (8 or 10 rows of M2 samples)
    lags = [int(rng.integers(1, n)), 17, band, 3 * band + 5, n - 1,                       # lag >= 0: the cross term is r[peak]
            -3, -band, -int(0.05 * n), -int(rng.integers(1, n // 10)),                   # small negative lags: minus the wrap-around part
            -(n // 2 - 1),                                                                # L = N/2 + 1: the bound (x N / L) exceeds 1e-5
            -int(rng.integers(n // 2 + 1, n - n // 8)), -(n - n // 8)]                   # lag < -N/2: the segment is the shorter read
    srcs, smps = [], []
    for i, lag in enumerate(lags):
        s_, t_ = planted(rng, n, lag, sign=-1.0 if i % 4 == 3 else 1.0, noise=[0.05, 0.25, 1.0][i % 3])
        srcs.append(s_); smps.append(t_)
    (lag_s, coef_s, ret_s), (lag_d, coef_d, ret_d), modes = run_both(mod, n, srcs, smps)
    assert sum(modes) == len(lags)
    assert modes[FAST] == 5 and modes[CORR] == 4 and modes[DIRECT] == 3, modes
    worst = 0.0
    for i, lag in enumerate(lags):
        o_ret, o_lag, o_coef = oracle.cross_correlation(srcs[i], smps[i])
        assert (o_ret, o_lag) == (0, lag), (i, lag, o_lag)
        assert (int(ret_s[i]), int(lag_s[i])) == (0, lag) == (int(ret_d[i]), int(lag_d[i])), (i, lag)
        assert abs(float(coef_s[i]) - o_coef) < COEF_TOL, (i, lag, float(coef_s[i]), o_coef)
        assert abs(float(coef_d[i]) - o_coef) < 1e-9, (i, lag, float(coef_d[i]), o_coef)   # the direct form is float64 all the way
        worst = max(worst, abs(float(coef_s[i]) - o_coef))
    print("N=%d: spectral form worst |delta| = %.3g (modes %s)" % (n, worst, modes))
    assert worst < 3e-6   # measured ~1e-7..1e-6: the float32 transforms' error at the peak, far inside the bound that gates it


def test_pairs_the_bound_does_not_cover_take_the_direct_form(mod):
    """quiet window of a loud source, offsets in either track, a constant sample (NaN like the reference), a silent sample:
    the error bound of the spectral form exceeds 1e-5 (or no peak exists), the pair is reduced directly, the answer is the oracle's"""
    n = 144000
    rng = np.random.default_rng(7)
    cases = {}
    s_, t_ = planted(rng, n, 5000)
    loud = s_.copy(); loud[5000 + n:] *= 40.0; loud[:5000] *= 40.0       # the matching window holds 1/800 of the source's energy
    cases["quiet window"] = (loud, t_)
    cases["source offset"] = (s_ + 30.0, t_)
    cases["sample offset"] = (s_, t_ + 20.0)
    cases["constant sample"] = (s_, np.full(n, 0.25, dtype=np.float32))
    cases["silent sample"] = (s_, np.zeros(n, dtype=np.float32))
    cases["ordinary"] = (s_, t_)
    names = list(cases)
    (lag_s, coef_s, ret_s), (lag_d, coef_d, ret_d), modes = run_both(mod, n, [cases[k][0] for k in names], [cases[k][1] for k in names])
    assert modes[DIRECT] >= 4 and modes[FAST] >= 1, modes
    for i, k in enumerate(names):
        o_ret, o_lag, o_coef, o_r, margin = oracle.cross_correlation(cases[k][0], cases[k][1], want_results=True)
        assert int(ret_s[i]) == o_ret == int(ret_d[i]), k
        if margin > 1.0 + 1e-9:
            assert int(lag_s[i]) == o_lag == int(lag_d[i]), k
        if o_ret == 0:
            assert abs(float(coef_s[i]) - o_coef) < COEF_TOL, (k, float(coef_s[i]), o_coef)
        else:
            assert np.isnan(coef_s[i]), k


def test_exact_peak_value_is_used_when_near_ties_were_re_evaluated(mod):
    """a tonal pair: thousands of near-ties, the winner's exact r (k_refine_dots) replaces the float32 value in the cross term"""
    n = 144000
    i = np.arange(2 * n, dtype=np.float64)
    src = (np.sin(0.05 * i) + 1e-3 * np.random.default_rng(1).normal(size=2 * n)).astype(np.float32)
    smp = (np.sin(0.05 * i[:n] + 2.0) + 1e-3 * np.random.default_rng(2).normal(size=n)).astype(np.float32)
    o_ret, o_lag, o_coef, o_r, margin = oracle.cross_correlation(src, smp, want_results=True)
    assert margin > 1.0 + 1e-12
    (lag_s, coef_s, ret_s), (lag_d, coef_d, ret_d), modes = run_both(mod, n, [src], [smp])
    assert (int(ret_s[0]), int(lag_s[0])) == (o_ret, o_lag)
    assert abs(float(coef_s[0]) - o_coef) < COEF_TOL and abs(float(coef_d[0]) - o_coef) < 1e-9


def test_same_pair_same_bits_alone_and_in_a_batch(mod):
    """the band sums, the window sums and the merge trees depend on the pair alone: a pair's coefficient has the same bits alone,
    anywhere in a batch, and on the device-resident entry point"""
    import torch
    n = 288000
    rng = np.random.default_rng(3)
    pairs = [planted(rng, n, lag) for lag in (12345, -777, -200000, 0, 250000)]
    src = np.stack([p[0] for p in pairs]); smp = np.stack([p[1] for p in pairs])
    with mod.Plan(n, len(pairs), 0) as plan:
        lag, coef, ret = plan.xcorr_batch_f32(src, smp)
        d_src = torch.from_numpy(src).cuda(); d_smp = torch.from_numpy(smp).cuda()
        d_lag = torch.zeros(len(pairs), dtype=torch.int64, device="cuda"); d_coef = torch.zeros(len(pairs), dtype=torch.float64, device="cuda")
        d_ret = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), len(pairs), d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr())
        plan.sync()
        assert np.array_equal(d_coef.cpu().numpy().view(np.uint64), coef.view(np.uint64))
    with mod.Plan(n, 1, 0) as one:
        for i in (4, 1, 2):
            l1, c1, r1 = one.xcorr_batch_f32(src[i: i + 1], smp[i: i + 1])
            assert int(l1[0]) == int(lag[i]) and c1.view(np.uint64)[0] == coef.view(np.uint64)[i], i


def test_double_abi_keeps_the_direct_form(mod):
    """cross_correlation(double*) promises the float64 reduction over the caller's values: identical segments give exactly 1.0
    (tests/test_cross_correlation.c:29) at a production length too, whether or not the frames crossed PCIe as float32"""
    n = 144000
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, 2 * n).astype(np.float32).astype(np.float64)
    with mod.Plan(n, 1, 0) as plan:
        before = plan.pearson_modes()
        ret, lag, coef = plan.xcorr_f64(x, x[1000: 1000 + n].copy())
        assert (ret, lag) == (0, 1000) and coef == 1.0
        assert plan.pearson_modes() == before


@pytest.mark.parametrize("n", [144000, 1440000])
def test_identical_segments_on_the_spectral_path(mod, n):
    """The ONE observable change the spectral form makes: the reference asserts
    `coefficient == 1.0` for identical segments (tests/test_cross_correlation.c:29).  The direct form gives exactly +-1.0 -- x and y go
    through the same operations -- and so do cross_correlation(double*), asx_xcorr_f64 and asx_stream_xcorr, which always reduce directly.
    The float32 batched entry points of a real-column plan take the cross term from the float32 transforms: there identical (or exactly
    negated) segments give a coefficient CLAMPED to [-1, 1] and within 1e-5 of +-1.0, not necessarily +-1.0 to the last bit;
    asx_plan_set_pearson(plan, 0) restores the reference's bits."""
    rng = np.random.default_rng(n + 5)
    big = rng.uniform(-1, 1, 2 * n).astype(np.float32)
    cases = [(0, 1.0), (12345, 1.0), (n - 1, -1.0), (n // 3, -1.0), (-777, 1.0), (-(n // 5), -1.0)]
    srcs, smps = [], []
    for lag, sign in cases:
        smp = (0.25 * rng.uniform(-1, 1, n)).astype(np.float32)       # what lies outside the overlap (negative lags only)
        if lag >= 0:
            smp[:] = sign * big[lag: lag + n]
        else:
            smp[-lag:] = sign * big[: n + lag]
        srcs.append(big); smps.append(smp)
    (lag_s, coef_s, ret_s), (lag_d, coef_d, ret_d), modes = run_both(mod, n, srcs, smps)
    assert modes[DIRECT] == 0, modes       # every case here is one the spectral form keeps
    for i, (lag, sign) in enumerate(cases):
        assert (int(ret_s[i]), int(lag_s[i])) == (0, lag) == (int(ret_d[i]), int(lag_d[i])), (i, lag)
        assert float(coef_d[i]) == sign, (i, lag, float(coef_d[i]))                       # the reference's exact +-1.0
        c = float(coef_s[i])
        assert abs(c) <= 1.0 and 1.0 - abs(c) < COEF_TOL and np.sign(c) == sign, (i, lag, c)
