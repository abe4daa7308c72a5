"""CPU: the oracle restatement against vectors frozen from the reference's own
xcorr/util/cc.py (oracle/make_golden.py) and against scipy/numpy identities."""
import numpy as np
import pytest
import scipy.signal

from oracle import ref_numpy as R

TOL = 1e-12


def test_kat_reference(golden):
    x, y = golden['kat_x'], golden['kat_y']
    np.testing.assert_allclose(R.cc(x, y, normalize=False), golden['kat_raw'], atol=TOL)
    np.testing.assert_allclose(R.cc(x, y, normalize=True), golden['kat_norm'], atol=TOL)
    np.testing.assert_allclose(R.cc(x, y, normalize=False, pad=False), golden['kat_nopad'], atol=TOL)
    np.testing.assert_allclose(golden['kat_raw'], [0, .5, 2, 3.5, 5, 4, 0], atol=1e-12)
    np.testing.assert_allclose(R.lag(4, .02), golden['kat_lag'], atol=TOL)
    np.testing.assert_allclose(R.lag(4, .02, pad=False), golden['kat_lag_nopad'], atol=TOL)


def test_cc_matches_reference_all_sizes(golden):
    for n in golden['sizes']:
        x, y = golden[f'x_{n}'], golden[f'y_{n}']
        for nrm in (0, 1):
            ref = golden[f'cc_{n}_norm{nrm}']
            got = R.cc(x, y, normalize=bool(nrm), dtype=np.float64)
            scale = max(np.abs(ref).max(), 1e-300)
            assert np.abs(got - ref).max() / scale < 1e-12, (n, nrm)
            # the three formulations of the same linear CC agree
            fair = R.cc_fair(x, y, normalize=bool(nrm), dtype=np.float64)
            assert np.abs(fair - ref).max() / scale < 1e-11, (n, nrm)
            c1d = R.correlate1d(x, y, normalize=bool(nrm), dtype=np.float64)
            assert np.abs(c1d - ref).max() / scale < 1e-11, (n, nrm)
        np.testing.assert_allclose(R.lag(int(n), 0.02), golden[f'lag_{n}'], atol=TOL)


def test_cc_equals_numpy_correlate(golden):
    x, y = golden['x_257'].astype(np.float64), golden['y_257'].astype(np.float64)
    np.testing.assert_allclose(R.cc(x, y, normalize=False), np.correlate(x, y, 'full'), atol=1e-9)


def test_weight_quirk_and_closed_form(golden):
    for n in golden['sizes']:
        n = int(n)
        if n < 3:
            continue
        ref = golden[f'weight_{n}']
        assert ref.shape == (2 * n - 3,)              # reference is two short
        np.testing.assert_allclose(R.weight(n), ref, rtol=1e-15)
        np.testing.assert_allclose(R.weight_full(n)[:2 * n - 3], ref, rtol=1e-15)
        np.testing.assert_allclose(R.weight(n, clip=3.0), golden[f'weight_{n}_clip'], rtol=1e-15)
        assert R.weight_full(n).shape == (2 * n - 1,)
        np.testing.assert_allclose(R.weight_full(n), R.weight_full(n)[::-1])


def test_big_digest(golden):
    n = int(golden['big_n'])
    rng = np.random.default_rng(12345)
    a = rng.standard_normal(n).astype(np.float32)
    b = (np.roll(a, 250) + 0.5 * rng.standard_normal(n).astype(np.float32)).astype(np.float32)
    r = R.cc_fair(a, b, dtype=np.float64)              # fast path (12 ms)
    assert r.shape == tuple(golden['big_shape'])
    assert int(np.argmax(r)) == int(golden['big_argmax']) == 179749
    # the reference ran with dtype=float32: its two L2 norms are float32
    # reductions (util/cc.py:68-77), good to ~1e-6 relative; the oracle keeps
    # float64 norms.  Tolerance 2e-6 of max|cc| (the parity bar is 1e-5).
    tol = 2e-6 * float(golden['big_max'])
    assert abs(r.max() - float(golden['big_max'])) < tol
    np.testing.assert_allclose(r[golden['big_probe_idx']], golden['big_probe_val'], atol=tol)
    assert abs(np.sqrt((r ** 2).sum()) - float(golden['big_l2'])) < 1e-5


def test_lag_indices_and_clip():
    n, delta = 180000, 0.02
    lg = R.lag(n, delta)
    assert R.lag_indices(n, delta, lg[0], lg[-1]) == (0, 2 * n - 1)
    imin, imax = R.clip_lag_indices(n, delta, (-10., 10.))
    assert (imin, imax) == (179499, 180499)
    assert R.lag_indices(n, delta, lg[imin], lg[imax - 1]) == (imin, imax)


def test_filter_restatement_matches_scipy_internals():
    # sosfiltfilt edge rule stated in SURVEY section 7
    sos = R.butter_sos(2, 3., 'highpass', 50.)
    assert sos.shape == (1, 6)
    sos4 = R.butter_sos(4, 1.5, 'highpass', 50.)
    assert sos4.shape == (2, 6)
    x = np.random.default_rng(0).standard_normal((3, 500)).astype(np.float32)
    for sos_ in (sos, sos4, R.butter_sos(4, (0.1, 10.), 'bandpass', 50.)):
        y = R.sosfiltfilt(sos_, x)
        assert y.dtype == np.float32 and y.shape == x.shape
        # explicit restatement of scipy's sosfiltfilt (the recipe the HIP
        # kernels follow): odd extension by edge = 3*ntaps, zi scaled by the
        # first sample, forward, reverse, backward, reverse, strip
        ns = sos_.shape[0]
        ntaps = 2 * ns + 1 - min((sos_[:, 2] == 0).sum(), (sos_[:, 5] == 0).sum())
        edge = 3 * ntaps
        zi = scipy.signal.sosfilt_zi(sos_)
        for r in range(x.shape[0]):
            xr = x[r].astype(np.float64)
            ext = np.concatenate([2 * xr[0] - xr[edge:0:-1], xr,
                                  2 * xr[-1] - xr[-2:-(edge + 2):-1]])
            yf, _ = scipy.signal.sosfilt(sos_, ext, zi=zi * ext[0])
            yb, _ = scipy.signal.sosfilt(sos_, yf[::-1], zi=zi * yf[-1])
            np.testing.assert_allclose(yb[::-1][edge:-edge], y[r], atol=2e-6)


def test_window_taper_detrend():
    w = R.window(1000, 50., max_length=5.)
    assert w[0] == 0 and w[250] == 1 and np.all(w[250:750] == 1)
    win = scipy.signal.windows.hann(500, False)
    np.testing.assert_allclose(w[:250], win[:250])
    np.testing.assert_allclose(w[-250:], win[-250:])
    x = (np.arange(100) * 0.5 + 3).astype(np.float32)[None]
    assert np.abs(R.detrend_linear(x)).max() < 1e-5
    assert abs(R.demean(x).mean()) < 1e-5


def test_pairs_engine_semantics():
    S, n, hop, W = 4, 400, 200, 3
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    out = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64)
    assert out.shape == (6, W, 2 * n - 1)
    for p, (i, j) in enumerate(pairs):
        for w in range(W):
            ref = R.cc(tr[i, w * hop:w * hop + n], tr[j, w * hop:w * hop + n], dtype=np.float64)
            np.testing.assert_allclose(out[p, w], ref, atol=1e-12)
    # stacking commutes with the inverse transform (linearity)
    np.testing.assert_allclose(R.stack(out), out.sum(axis=1), atol=1e-12)


def test_timeshift_integer_delay_is_roll():
    x = np.random.default_rng(3).standard_normal((2, 400))
    y = R.timeshift(x, 4 / 50., 50.)
    # f grid uses fN rounded to integer Hz (fft.py:358): 25 Hz exactly here
    np.testing.assert_allclose(y, np.roll(x, 4, axis=-1), atol=1e-9)


def test_correlate2d_restatement_is_the_defining_sum():
    """correlate.py:230-246 (fft2 / conj / ifft2 / fftshift) == the double sum,
    and == scipy.signal.correlate2d(mode='full')."""
    import scipy.signal
    rng = np.random.default_rng(8)
    for F, L in [(1, 1), (2, 5), (4, 6), (7, 3)]:
        f = rng.standard_normal((F, L))
        g = rng.standard_normal((F, L))
        a = R.correlate2d(f, g, normalize=False)
        assert a.shape == (2 * F - 1, 2 * L - 1)
        assert np.allclose(a, R.correlate2d_direct(f, g), atol=1e-12)
        assert np.allclose(a, scipy.signal.correlate2d(f, g, mode='full'), atol=1e-12)
    f = rng.standard_normal((3, 5, 8))
    n = R.norm2d(f)
    assert np.allclose((n ** 2).sum(axis=(-2, -1)), 1.0)
    assert np.isclose(R.correlate2d(f, f)[..., 4, 7], 1.0).all()
