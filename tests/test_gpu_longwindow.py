"""GPU: windows longer than any single transform: the reference's production
setting window_length = 86400 s @ 50 Hz (n = 4 320 000) with clip_lag = (0, 9 h)
(scripts/cc.py:25-29, examples/core/process.py:25-28) needs n + max|lag| = 5.94 M points; the
plan assembles it block-wise on the time-stacking engine (csrc/seg_plan.hip: seg_run).
Checked against the oracle's rfft restatement of util/cc.py (R.cc_fair, float64)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402  (checker only)
import xcorr_amd as X                      # noqa: E402

TOL = 1e-5


def _rel(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / np.abs(ref).max())


def test_production_day_window_clip_lag(ctx):
    n, fs = 4320000, 50.
    rng = np.random.default_rng(2024)
    c = rng.standard_normal(n + 30000).astype(np.float32)
    x = (c[:n] + 0.5 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)          # peak of cc(x, y) at +600 s,
    y = (c[30000:] + 0.5 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)     # inside the retained lags
    imin, imax = n - 1, n - 1 + int(9 * 3600 * fs)              # lags [0, 9 h)
    plan = X.CCPlan(ctx, n, imin, imax, normalize=True)
    assert plan.L == 1620000 and plan.N == 368640                # block transform of the segmented plan
    got = plan.cc(x[None], y[None])[0]
    ref = R.cc_fair(x, y, normalize=True, dtype=np.float64)[imin:imax]
    assert got.shape == ref.shape
    assert _rel(got, ref) < TOL
    assert int(np.argmax(got)) == int(np.argmax(ref)) == 30000 and ref[30000] > 0.7
    # both signs of lag, unbiased, device-resident rows
    imin2, imax2 = n - 1 - 400000, n - 1 + 250001
    plan2 = X.CCPlan(ctx, n, imin2, imax2, normalize=True, unbiased=True)
    dx, dy = ctx.array(np.stack([x, y])), ctx.array(np.stack([y, x]))
    out = plan2.cc(dx, dy)
    got2 = out.to_host()
    ref2 = R.cc_fair(x, y, normalize=True, dtype=np.float64) * R.weight_full(n)
    assert _rel(got2[0], ref2[imin2:imax2]) < TOL
    assert _rel(got2[1], ref2[::-1][imin2:imax2]) < TOL           # cc(y, x)[k] = cc(x, y)[-k]
    plan.close(); plan2.close()


def test_segmented_plan_matches_single_transform(ctx, monkeypatch):
    """Forced segmentation at small sizes (XCX_SEG_FORCE, block transform 4096) against the ordinary
    plan and the oracle: ragged last block, lag ranges on both sides, flags, the pairs form."""
    rng = np.random.default_rng(7)
    n = 10007
    x = (rng.standard_normal((3, n)) + 0.4).astype(np.float32)
    y = (np.roll(x, 37, axis=1) + rng.standard_normal((3, n))).astype(np.float32)
    cases = [dict(), dict(index_min=n - 1, index_max=n + 5000), dict(index_min=123, index_max=2 * n - 2 - 77),
             dict(index_min=n - 3000, index_max=n - 10), dict(normalize=False, demean=True),
             dict(onebit=True, demean=True), dict(unbiased=True, index_min=2000, index_max=17000)]
    for kw in cases:
        ref_plan = X.CCPlan(ctx, n, **kw)
        ref = ref_plan.cc(x, y)
        monkeypatch.setenv('XCX_SEG_FORCE', '1')
        monkeypatch.setenv('XCX_SEG_N', '4096')
        seg = X.CCPlan(ctx, n, **kw)
        monkeypatch.delenv('XCX_SEG_FORCE')
        monkeypatch.delenv('XCX_SEG_N')
        assert seg.N == 4096 and ref_plan.N != 4096
        got = seg.cc(x, y)
        scale = np.abs(X.CCPlan(ctx, n, **{k: v for k, v in kw.items() if not k.startswith('index')}).cc(x, y)).max()
        assert np.abs(got - ref).max() / scale < 2e-6, kw
        pr = np.array([(0, 1), (2, 0), (1, 1)], np.int32)
        T = n + 500
        tr = (rng.standard_normal((3, T))).astype(np.float32)
        a = seg.pairs(tr, pr, 2, 500)
        b = ref_plan.pairs(tr, pr, 2, 500)
        assert np.abs(a - b).max() / np.abs(b).max() < 2e-5 or kw.get('unbiased'), kw
        seg.close(); ref_plan.close()
    o = R.cc_fair(x[0].astype(np.float64), y[0].astype(np.float64), normalize=True, dtype=np.float64)
    monkeypatch.setenv('XCX_SEG_FORCE', '1')
    monkeypatch.setenv('XCX_SEG_N', '4096')
    seg = X.CCPlan(ctx, n)
    assert _rel(seg.cc(x[:1], y[:1])[0], o) < TOL
    from xcorr_amd._ffi import XcxError
    with pytest.raises(XcxError):
        X.CCPlan(ctx, n, stack=True).stack_add if False else X.CCPlan(ctx, 4320000, stack=True)
    with pytest.raises(XcxError):
        X.CCPlan(ctx, 4320000, whiten=(50., 1., 10., .5))


def test_timeshift_of_long_rows(ctx):
    """Post-processing of the production lag axis (clip_lag = (0, 9 h) -> 1.62 M lags): the
    period-n Dirichlet convolution runs through the block-decomposed correlation."""
    from xcorr_amd import ops
    rng = np.random.default_rng(21)
    fs = 50.
    for size in (500001, 1620000):
        x = (rng.standard_normal((2, size)) * np.hanning(size)).astype(np.float32)
        delay = np.array([0.0071, -0.0183])
        got = ops.timeshift(ctx, x, delay, fs)
        ref = R.timeshift(x, delay, fs)
        assert got.shape == ref.shape == (2, size - (size & 1))
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_timeshift_at_the_single_transform_boundary(ctx):
    """Rows up to 393 216 samples are shifted by ONE correlation against two periods of the Dirichlet kernel
    (2n - 1 <= 786 432, the longest transform; the pair pass writes the shifted row), longer ones by the
    one-pass block convolution with that kernel (seg_run's convolution mode): both sides of the switch, odd sizes too."""
    from xcorr_amd import ops
    rng = np.random.default_rng(23)
    fs = 50.
    for size in (393216, 393217, 393218, 393219):
        x = (rng.standard_normal((2, size)) * np.hanning(size)).astype(np.float32)
        delay = np.array([0.0113, -0.75])
        got = ops.timeshift(ctx, x, delay, fs)
        ref = R.timeshift(x, delay, fs)
        assert got.shape == ref.shape == (2, size - (size & 1))
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), size


def test_production_postprocess_chain(ctx):
    """core/postprocess.py:133-141 on the production container layout: day-long windows
    (npts = 4 320 000), clip_lag = (0, 9 h) -> rows of 1 620 000 lags, device-resident."""
    from xcorr_amd.core.postprocess import postprocess_rows
    n, fs = 4320000, 50.
    i0, i1 = n - 1, n - 1 + 1620000
    rng = np.random.default_rng(22)
    rows = (1e-3 * rng.standard_normal((2, i1 - i0))).astype(np.float32)
    rows[:, 5000:5400] += (1e-2 * np.hanning(400)).astype(np.float32)
    delay = np.array([0.0123, -0.0042])
    d = ctx.array(rows.copy())
    y = postprocess_rows(d, delay, fs, n, i0, i1, ctx=ctx)
    got = y.to_host()
    y.free(); d.free()
    ref = R.postprocess_chain(rows.astype(np.float64), n, fs, i0, i1, delay)
    assert got.shape == ref.shape == (2, 1620000)
    assert _rel(got, ref) < 2 * TOL                 # six chained float32 stages


def test_segmented_rows_in_several_chunks(ctx, monkeypatch):
    """A call whose rows do not fit the workspace budget runs in chunks of rows, each transforming the windows its
    rows meet (XCX_SEG_BUDGET_MB; by default half of the free device memory, 4-64 GB): all pairs of 7 stations and
    a batch of unrelated rows, 1 MB at a time, give bit for bit what one chunk gives; rows that share a delay kernel
    in the time shift too."""
    from xcorr_amd import ops
    rng = np.random.default_rng(31)
    n, S = 10007, 7
    monkeypatch.setenv('XCX_SEG_FORCE', '1')
    monkeypatch.setenv('XCX_SEG_N', '4096')
    seg = X.CCPlan(ctx, n, index_min=n - 1 - 300, index_max=n + 6000)
    assert seg.N == 4096
    tr = rng.standard_normal((S, n + 64)).astype(np.float32)
    pr = np.array([(i, j) for i in range(S) for j in range(i, S)], np.int32)
    x = rng.standard_normal((9, n)).astype(np.float32)
    y = rng.standard_normal((9, n)).astype(np.float32)
    t = (rng.standard_normal((11, 9107)) * np.hanning(9107)).astype(np.float32)
    delay = np.array([0.003, -0.011, 0.02])[rng.integers(0, 3, 11)]
    one = seg.pairs(tr, pr, 2, 64).copy(), seg.cc(x, y).copy(), ops.timeshift(ctx, t, delay, 50.).copy()
    monkeypatch.setenv('XCX_SEG_BUDGET_MB', '1')
    many = seg.pairs(tr, pr, 2, 64), seg.cc(x, y), ops.timeshift(ctx, t, delay, 50.)
    for a, b in zip(one, many):
        assert np.array_equal(a, b)
    monkeypatch.delenv('XCX_SEG_FORCE'); monkeypatch.delenv('XCX_SEG_N'); monkeypatch.delenv('XCX_SEG_BUDGET_MB')
    ref = X.CCPlan(ctx, n, index_min=n - 1 - 300, index_max=n + 6000)
    b = ref.pairs(tr, pr, 2, 64)
    assert np.abs(many[0] - b).max() / np.abs(b).max() < 2e-5
    assert _rel(many[2], R.timeshift(t, delay, 50.)) < TOL
    seg.close(); ref.close()


def test_segmented_plan_random_sweep(ctx):
    """tools/fuzz_seg.py for a few seconds: random window lengths (ragged last blocks), lag slices on either side of
    zero lag, flags and batches, segmentation forced at block transforms of 4096 / 8192 points -- against the
    single-transform plan and the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'fuzz_seg', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_seg.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases, worst = mod.run(5.0, seed=4)
    assert cases >= 3 and worst < 1e-5
