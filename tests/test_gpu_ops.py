"""GPU: row-wise signal ops and SOS filters against the oracle
(scipy/numpy restatements of xcorr/signal/*)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
from xcorr_amd import ops                  # noqa: E402

TOL = 1e-5


def relerr(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.fixture(scope='module')
def x():
    rng = np.random.default_rng(1)
    return (rng.standard_normal((7, 5001)) + 3.0 + 1e-3 * np.arange(5001)).astype(np.float32)


def test_norm_demean_detrend(ctx, x):
    assert relerr(ops.norm1d(ctx, x), R.norm1d(x)) < TOL
    assert relerr(ops.demean(ctx, x), R.demean(x)) < TOL
    assert relerr(ops.detrend_linear(ctx, x), R.detrend_linear(x)) < 2 * TOL
    one = x[:1, :1]
    assert ops.demean(ctx, one)[0, 0] == 0
    assert np.all(np.isnan(ops.norm1d(ctx, np.zeros((2, 9), np.float32))))


def test_taper_and_window_rules(ctx, x):
    assert relerr(ops.taper(ctx, x, 250), R.taper(x, 50.0, max_length=5.0)) < 1e-6
    assert relerr(ops.taper(ctx, x, 75), R.taper(x, 50.0, max_length=1.5)) < 1e-6
    assert relerr(ops.taper(ctx, x, 500), R.taper(x, 50.0, max_percentage=0.1 - 1e-9)) < 1e-6
    # half-length flanks (the reference default nmax = 0.5*len, odd length: overlap)
    nm = int(round(0.5 * x.shape[-1]))
    assert relerr(ops.taper(ctx, x, nm), x * R.window(x.shape[-1], 50.).astype(np.float32)) < 1e-6
    cos = R.cosine_taper_obspy(5001, 50.0, 0.05)
    assert relerr(ops.taper(ctx, x, 250, 'cosine'), x * cos.astype(np.float32)) < 1e-6
    from xcorr_amd._ffi import XcxError
    with pytest.raises(XcxError, match='nmax'):
        ops.taper(ctx, x, 0)


def test_unbias_and_onebit(ctx):
    rng = np.random.default_rng(2)
    n = 2501
    c = rng.standard_normal((4, 2 * n - 1)).astype(np.float32)
    assert relerr(ops.unbias(ctx, c, n), R.unbias(c, n)) < 1e-6
    sl = c[:, 100:4000]
    assert relerr(ops.unbias(ctx, sl, n, 100, 4000), R.unbias(sl, n, 100, 4000)) < 1e-6
    with pytest.raises(ValueError):
        ops.unbias(ctx, sl, n, 100, 3999)
    v = np.array([[-2., 0., 3., -0., 1e-30]], np.float32)
    np.testing.assert_array_equal(ops.onebit(ctx, v), [[-1, 0, 1, 0, 1]])


FILTERS = [(2, 3.0, 'highpass'), (4, 1.5, 'highpass'), (4, (0.1, 10.0), 'bandpass'),
           (4, 20.0, 'lowpass'), (8, (1.0, 5.0), 'bandpass'), (1, 2.0, 'lowpass')]


@pytest.mark.parametrize('order,freq,btype', FILTERS)
def test_sos_filters_rows(ctx, x, order, freq, btype):
    sos = R.butter_sos(order, freq, btype, 50.0)
    assert relerr(ops.sosfiltfilt(ctx, x, sos), R.sosfiltfilt(sos, x)) < TOL
    assert relerr(ops.sosfilt(ctx, x, sos), R.sosfilt(sos, x)) < TOL


def test_sos_long_rows_chunked_scan(ctx):
    """Few long rows take the exact chunked-scan path (day-long traces)."""
    rng = np.random.default_rng(3)
    xl = (rng.standard_normal((3, 300001)) + 1.0).astype(np.float32)
    for order, freq, btype in [(4, (0.1, 10.0), 'bandpass'), (2, 0.05, 'highpass')]:
        sos = R.butter_sos(order, freq, btype, 50.0)
        assert relerr(ops.sosfilt(ctx, xl, sos), R.sosfilt(sos, xl)) < TOL
        assert relerr(ops.sosfiltfilt(ctx, xl, sos), R.sosfiltfilt(sos, xl)) < TOL


def test_sos_many_short_rows_and_errors(ctx):
    rng = np.random.default_rng(4)
    xm = rng.standard_normal((300, 999)).astype(np.float32)
    sos = R.butter_sos(2, 3.0, 'highpass', 50.0)
    assert relerr(ops.sosfiltfilt(ctx, xm, sos), R.sosfiltfilt(sos, xm)) < TOL
    from xcorr_amd._ffi import XcxError
    with pytest.raises(XcxError, match='greater than padlen'):     # scipy's message
        ops.sosfiltfilt(ctx, xm[:, :9], sos)
    with pytest.raises(ValueError, match='sos array must be shape'):
        ops.sosfilt(ctx, xm, np.zeros((2, 5)))


def test_postprocess_chain_stages(ctx):
    """core/postprocess.py:133-141 without the timeshift stage (not built):
    unbias -> demean -> taper(5 s) -> highpass sosfiltfilt -> taper(1.5 s)."""
    rng = np.random.default_rng(5)
    n, fs = 3000, 50.
    c = (rng.standard_normal((6, 2 * n - 1)) * 0.01).astype(np.float32)
    y = ops.unbias(ctx, c, n)
    y = ops.demean(ctx, y)
    y = ops.taper(ctx, y, 250)
    sos = R.butter_sos(2, 3., 'highpass', fs)
    y = ops.sosfiltfilt(ctx, y, sos)
    y = ops.taper(ctx, y, 75)
    r = R.unbias(c, n)
    r = R.demean(r)
    r = R.taper(r, fs, max_length=5.)
    r = R.sosfiltfilt(sos, r)
    r = R.taper(r, fs, max_length=1.5)
    assert relerr(y, r) < TOL


@pytest.mark.parametrize('size', [64, 1001, 5999, 60001])
def test_timeshift_matches_reference_semantics(ctx, size):
    """signal/timeshift.py:90-96 (rfft drops the last sample of odd axes,
    integer-Hz Nyquist rounding) as a Dirichlet-kernel circular convolution."""
    rng = np.random.default_rng(size)
    fs = 50.
    x = rng.standard_normal((5, size)).astype(np.float32)
    x *= np.hanning(size).astype(np.float32)          # tapered like a post-processed CCF
    delay = np.array([0.0, 0.02, -0.013, 0.25 / fs, 1.37])
    got = ops.timeshift(ctx, x, delay, fs)
    ref = R.timeshift(x, delay, fs)
    assert got.shape == ref.shape == (5, size - (size & 1))
    assert relerr(got, ref) < TOL
    # zero delay is the identity on the first n samples; one sample = roll
    n = size - (size & 1)
    assert relerr(got[0], x[0, :n]) < TOL
    assert relerr(got[1], np.roll(x[1, :n], 1)) < TOL
    # scalar delay broadcasts
    assert relerr(ops.timeshift(ctx, x, -0.013, fs), R.timeshift(x, -0.013, fs)) < TOL


@pytest.mark.parametrize('size', [64, 1001, 5999, 60000])
def test_timeshift_fft_path(ctx, size):
    """fast=False (signal/timeshift.py:90-96 over fft.py:104-113,239-241): every sample kept; an
    odd axis comes back times cos(2 pi m / n) exactly as the reference's fftshift-twice ifft gives."""
    rng = np.random.default_rng(size + 1)
    fs = 50.
    x = rng.standard_normal((5, size)).astype(np.float32)
    x *= np.hanning(size).astype(np.float32)
    delay = np.array([0.0, 0.02, -0.013, 0.25 / fs, 1.37])
    got = ops.timeshift(ctx, x, delay, fs, fast=False)
    ref = R.timeshift_fft(x, delay, 1. / fs)
    assert got.shape == ref.shape == (5, size)
    assert relerr(got, ref) < TOL
    if size % 2 == 0:           # the two paths agree on an even axis
        assert relerr(got, ops.timeshift(ctx, x, delay, fs)) < TOL
        assert relerr(got[1], np.roll(x[1], 1)) < TOL
    else:                       # the quirk is really there
        m = np.arange(size)
        assert relerr(got[0], x[0] * np.cos(2 * np.pi * m / size)) < TOL
    from xcorr_amd import signal as S
    assert relerr(S.timeshift(x, delay, fast=False, sampling_rate=fs), ref) < TOL


@pytest.mark.parametrize('size,fast', [(6000, True), (5999, False), (40000, True)])
def test_timeshift_rows_that_share_a_shift(ctx, size, fast):
    """Rows of a launch group with the same delay share one Dirichlet kernel (generated and transformed
    once): repeated delays in any order -- the reference's -(pair_offset + time_offset) pattern,
    core/postprocess.py:118 -- over more than one launch group (256 rows), against the per-row oracle."""
    rng = np.random.default_rng(size)
    fs = 50.
    rows = 300 if size < 10000 else 12
    x = (rng.standard_normal((rows, size)) * np.hanning(size)).astype(np.float32)
    vals = np.array([0.0, 0.0049, -0.0131, 0.02, 0.25 / fs])
    delay = vals[rng.integers(0, len(vals), rows)]
    delay[7] = 0.777                                   # one shift of its own in the middle of the group
    got = ops.timeshift(ctx, x, delay, fs, fast=fast)
    ref = R.timeshift(x, delay, fs) if fast else R.timeshift_fft(x, delay, 1. / fs)
    assert got.shape == ref.shape
    worst = max(relerr(got[r], ref[r]) for r in range(rows))
    assert worst < TOL, worst
    # every row of one shift: one kernel for the whole call
    one = ops.timeshift(ctx, x[:40], 0.0049, fs, fast=fast)
    ref1 = R.timeshift(x[:40], 0.0049, fs) if fast else R.timeshift_fft(x[:40], np.full(40, 0.0049), 1. / fs)
    assert relerr(one, ref1) < TOL


def test_timeshift_full_lag_axis(ctx):
    """L = 359999 lags -> n = 359998 = 2 * 179999 (prime): no FFT of that length
    exists in the engine; the circular convolution runs at N = 737280."""
    rng = np.random.default_rng(11)
    fs, size = 50., 359999
    x = (rng.standard_normal((2, size)) * np.hanning(size)).astype(np.float32)
    delay = np.array([0.0071, -0.0183])
    got = ops.timeshift(ctx, x, delay, fs)
    ref = R.timeshift(x, delay, fs)
    assert got.shape == (2, 359998)
    assert relerr(got, ref) < TOL


def test_postprocess_chain_complete(ctx):
    """core/postprocess.py:133-141: unbias -> demean -> taper(5 s) ->
    timeshift(-(pair_offset+time_offset)) -> highpass sosfiltfilt -> taper(1.5 s)."""
    rng = np.random.default_rng(6)
    n, fs = 3000, 50.
    c = (rng.standard_normal((6, 2 * n - 1)) * 0.01).astype(np.float32)
    delay = rng.uniform(-0.02, 0.02, 6)
    y = ops.unbias(ctx, c, n)
    y = ops.demean(ctx, y)
    y = ops.taper(ctx, y, 250)
    y = ops.timeshift(ctx, y, delay, fs)
    sos = R.butter_sos(2, 3., 'highpass', fs)
    y = ops.sosfiltfilt(ctx, y, sos)
    y = ops.taper(ctx, y, 75)
    ref = R.postprocess_chain(c, n, fs, 0, 2 * n - 1, delay)
    assert y.shape == ref.shape
    assert relerr(y, ref) < TOL


def test_snr_pieces(ctx):
    """signal/snr.py:107-116: rms of the noise window, max |x| (and where) in
    the signal window."""
    import xcorr_amd.signal as sig
    rng = np.random.default_rng(8)
    x = rng.standard_normal((4, 3, 5000)).astype(np.float32)
    x[..., 2600] = 9.0
    x[1, 2, 2600] = 1.0
    x[1, 2, 3100] = -7.5
    lag = (np.arange(5000) - 2500) / 50.
    signal = (lag >= 0.) & (lag <= 20.)
    noise = (lag >= 30.) | (lag <= -30.)
    r, s, n, a = sig.snr(x, signal, noise, extend=True)
    rr, rs, rn, ra = R.snr(x, signal, noise)
    assert r.shape == (4, 3)
    np.testing.assert_array_equal(a, ra)
    assert a[1, 2] == 3100 and a[0, 0] == 2600
    np.testing.assert_allclose(s, rs, rtol=1e-6)
    np.testing.assert_allclose(n, rn, rtol=1e-5)
    np.testing.assert_allclose(sig.snr(x, signal, noise, decibels=True), 20 * np.log10(rr), rtol=1e-5)
    re_, rse, rne, _ = R.snr_envelope(x, signal, noise)
    e, se, ne, _ = sig.snr(x, signal, noise, envelope=True, extend=True)
    np.testing.assert_allclose(se, rse, rtol=1e-5)
    np.testing.assert_allclose(e, re_, rtol=2e-5)
    with pytest.raises(ValueError):
        sig.snr(x, np.zeros(5000, bool), noise)


@pytest.mark.parametrize('h', [0, 1, 7, 250, 2500, 3072])
def test_running_absolute_mean_normalisation(ctx, h):
    """Self-defined op named by the north star (Bensen et al. 2007 eq. 1)."""
    rng = np.random.default_rng(h)
    x = (rng.standard_normal((3, 20011)) * (1 + 5 * (rng.random((3, 20011)) > 0.999))).astype(np.float32)
    x[1, 5000:5010] *= 100.                      # an "earthquake"
    x[2, :4000] = 0.                             # dead channel start: w = 0 -> y = 0
    y = ops.ram_normalize(ctx, x, h)
    ref = R.ram_normalize(x, h)
    assert relerr(y, ref) < TOL
    if h == 0:
        assert np.all(np.abs(y[0]) == 1)
    if h >= 250:
        assert np.all(y[2, :4000 - h] == 0)
    short = x[:, :37]
    assert relerr(ops.ram_normalize(ctx, short, h), R.ram_normalize(short, h)) < TOL
    from xcorr_amd._ffi import XcxError
    with pytest.raises(XcxError):
        ops.ram_normalize(ctx, x, 3073)


def test_rms_along_last_axis(ctx):
    import xcorr_amd.signal as sig
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((3, 4, 7001)) + 0.5).astype(np.float32)
    got = sig.rms(x)
    assert got.shape == (3, 4)
    assert relerr(got, np.sqrt(np.mean(np.square(x.astype(np.float64)), axis=-1))) < TOL
    assert np.array_equal(sig.absolute(x), np.abs(x))


def test_post_prep_is_bitwise_the_three_ops(ctx):
    """xcx_post_prep (fused unbias -> demean -> taper, f1) == xcx_unbias + xcx_demean + xcx_taper, bit for bit,
    on full and clipped lag slices, device and host rows."""
    rng = np.random.default_rng(31)
    for npts, imin, imax, nmax in ((3000, 0, 5999, 250), (3000, 1000, 4500, 75), (500, 499, 999, 10), (2000, 0, 3999, 0),
                                   (40000, 0, 79999, 250), (50000, 1000, 90001, 75)):      # the last two: rows split over workgroups
        L = imax - imin
        x = (rng.standard_normal((7, L)) * 0.01 + 0.003).astype(np.float32)
        ref = ops.unbias(ctx, x, npts, imin, imax)
        ref = ops.demean(ctx, ref)
        if nmax:
            ref = ops.taper(ctx, ref, nmax)
        got = ops.post_prep(ctx, x, npts, imin, imax, unbiased=True, taper_nmax=nmax)
        assert np.array_equal(got, ref)
        d = ctx.array(x)
        ops.post_prep(ctx, d, npts, imin, imax, unbiased=True, taper_nmax=nmax)
        assert np.array_equal(d.to_host(), ref)
        r2 = R.unbias(x.astype(np.float64), npts, imin, imax)
        r2 = R.demean(r2)
        if nmax:
            r2 = r2 * R.window(L, 50., max_length=nmax / 50.)
        assert relerr(got, r2) < TOL
    nb = ops.post_prep(ctx, x, npts, imin, imax, unbiased=False, taper_nmax=0)
    assert np.array_equal(nb, ops.demean(ctx, x))


def test_postprocess_rows_device_resident(ctx):
    """core.postprocess_rows: the whole chain on a DeviceArray (nothing crosses PCIe) equals the
    host-array form and the oracle chain."""
    from xcorr_amd.core.postprocess import postprocess_rows
    rng = np.random.default_rng(32)
    npts, fs = 3001, 50.
    L = 2 * npts - 1
    x = (rng.standard_normal((5, L)) * 0.01).astype(np.float32)
    delay = rng.uniform(-0.03, 0.03, 5)
    host = postprocess_rows(x, delay, fs, npts, 0, L, ctx=ctx)
    d = ctx.array(x)
    dev = postprocess_rows(d, delay, fs, npts, 0, L, ctx=ctx)
    assert dev.shape == (5, L - 1) and np.array_equal(dev.to_host(), host)
    ref = np.stack([R.postprocess_chain(x[i].astype(np.float64), npts, fs, 0, L, delay[i]) for i in range(5)])
    assert relerr(host, ref) < 2 * TOL


def test_demean_out_of_place(ctx):
    """xcx_detrend_to: device rows src -> dst, source untouched; long-row and short-row paths."""
    rng = np.random.default_rng(31)
    for rows, n in ((3, 100001), (40, 5000)):
        x = (rng.standard_normal((rows, n)) + 2.5).astype(np.float32)
        d, o = ctx.array(x), ctx.empty(x.shape)
        ops.demean(ctx, d, out=o)
        assert np.array_equal(d.to_host(), x)
        assert np.array_equal(o.to_host(), ops.demean(ctx, x.copy()))          # bit-identical to the in-place op
        assert relerr(o.to_host(), R.demean(x)) < TOL
    with pytest.raises(ValueError):
        ops.demean(ctx, x, out=o)                                              # host source
