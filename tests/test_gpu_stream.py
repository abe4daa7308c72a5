"""GPU: sample-rate changing preprocess ops (decimate / lanczos interpolate /
running rms) and the ObsPy-style operation list, against the numpy/scipy oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
from xcorr_amd import ops, stream          # noqa: E402
from xcorr_amd._ffi import XcxError        # noqa: E402

TOL = 1e-5


def relerr(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / np.abs(ref).max())


@pytest.fixture(scope='module')
def raw():
    """250 Hz records (the reference's EDH hydrophone rate) with offset, trend and tones."""
    rng = np.random.default_rng(21)
    n = 250 * 400 + 3
    t = np.arange(n) / 250.
    x = rng.standard_normal((5, n)) + 2.0 + 1e-3 * t + np.sin(2 * np.pi * 3.1 * t) + 0.5 * np.sin(2 * np.pi * 60. * t)
    return x.astype(np.float32)


@pytest.mark.parametrize('factor', [1, 2, 5, 10, 16])
def test_decimate(ctx, raw, factor):
    y, fs = stream.decimate(raw, 250., factor, ctx=ctx)
    ref = R.decimate(raw, 250., factor)
    assert fs == 250. / factor and y.shape == ref.shape == (5, -(-raw.shape[1] // factor))
    assert relerr(y, ref) < TOL
    y2, _ = stream.decimate(raw, 250., factor, no_filter=True, ctx=ctx)
    assert np.array_equal(y2, raw[:, ::factor])
    # the design is ObsPy's: 96 dB stop band at the new Nyquist, order <= 12
    sos = stream.cheby2_lowpass_sos(0.5 * 250. / factor, 250.)
    assert sos.shape[0] <= 6 and np.allclose(sos, R.cheby2_lowpass_sos(0.5 * 250. / factor, 250.))
    with pytest.raises(ArithmeticError):
        stream.decimate(raw, 250., 17, ctx=ctx)


@pytest.mark.parametrize('new_rate,a', [(50., 20), (100., 20), (40., 8), (250., 1), (333.3, 12)])
def test_lanczos_interpolate(ctx, raw, new_rate, a):
    x = raw[:, :50000]
    y, fs = stream.interpolate(x, 250., new_rate, a=a, ctx=ctx)
    n_out = int(np.floor((x.shape[1] - 1) / 250. * new_rate)) + 1
    ref = R.lanczos_interpolation(x, 0.0, 250. / new_rate, n_out, a)
    assert y.shape == ref.shape and fs == new_rate
    assert relerr(y, ref) < TOL
    if new_rate == 250.:
        assert relerr(y, x) < TOL                      # identity at integer positions
    # shifted start, explicit npts
    y2, _ = stream.interpolate(x, 250., new_rate, a=a, starttime_offset=0.0137, npts=1000, ctx=ctx)
    ref2 = R.lanczos_interpolation(x, 0.0137 * 250., 250. / new_rate, 1000, a)
    assert relerr(y2, ref2) < TOL
    with pytest.raises(ValueError):
        stream.interpolate(x, 250., new_rate, a=a, npts=10 ** 9, ctx=ctx)


def test_lanczos_positions_next_to_a_sample(ctx, raw):
    """Interpolation points a rounding error away from an input sample (250 -> 333.3 Hz: k * 0.750075 passes
    30003.000000000004): the kernel's sines come from a rotation per tap, the centre taps from sinpi itself."""
    x = raw[:, :50000]
    for step, off, a in ((0.750075, 0.0, 12), (1.0 + 1e-13, 1e-12, 20), (2.5, 3.0000000001, 8), (0.2 + 1e-14, 2.0 - 1e-13, 3)):
        nout = int((x.shape[1] - 1 - off) / step)
        got = ops.lanczos(ctx, x, nout, off, step, a)
        ref = R.lanczos_interpolation(x, off, step, nout, a)
        assert relerr(got, ref) < TOL, (step, off, a)


def test_lanczos_rational_steps_take_the_polyphase_kernel(ctx, raw):
    """Steps P / Q with Q <= 64 (every resampling between round rates: 40 -> 50 Hz = 4/5, 50 -> 20 Hz = 5/2) repeat the
    fractional position with period Q: the weights come from a per-call table and the kernel is a polyphase FIR
    (resample.hip: k_lanczos_table / k_lanczos_poly).  Against the oracle, which evaluates every weight from its own
    position: offsets, odd numbers of outputs, phases that land on samples, taps clipped at both ends of the row, a
    kernel wider than the four-sample loads' stride, too few outputs for a table."""
    x = raw[:, :40001]
    n = x.shape[1]
    for step, off, a in ((0.8, 0.0, 20), (2.5, 0.0, 20), (10.0 / 3.0, 1.25, 7), (1.0, 0.5, 3), (0.8, 2.0 - 1e-13, 20),
                         (1.25, 0.0, 1), (64.0 / 63.0, 0.3, 5), (0.5, 0.0, 13)):
        nout = int((n - 1 - off) / step) + 1
        for m in (nout, nout - 1, 3):
            got = ops.lanczos(ctx, x, m, off, step, a)
            ref = R.lanczos_interpolation(x, off, step, m, a)
            assert got.shape == ref.shape and relerr(got, ref) < TOL, (step, off, a, m)
    d = ctx.array(x)                                     # device rows, many rows (grid z)
    y = ops.lanczos(ctx, d, 30000, 0.0, 0.8, 20)
    assert relerr(y.to_host(), R.lanczos_interpolation(x, 0.0, 0.8, 30000, 20)) < TOL
    y.free(); d.free()


def test_lanczos_bandlimited_signal_is_recovered(ctx):
    t = np.arange(20000) / 250.
    x = np.sin(2 * np.pi * 2.3 * t).astype(np.float32)[None]
    y, _ = stream.interpolate(x, 250., 50., a=20, ctx=ctx)
    tn = np.arange(y.shape[1]) / 50.
    inner = slice(50, -50)
    assert np.abs(y[0, inner] - np.sin(2 * np.pi * 2.3 * tn)[inner]).max() < 2e-3


def test_running_rms(ctx, raw):
    for window in (1.0, 10.0):
        y, fs = stream.running_rms(raw, 250., window, ctx=ctx)
        ref = R.running_rms(raw, 250., window)
        assert y.shape == ref.shape and fs == 1.0 / window
        assert relerr(y, ref) < TOL
    assert relerr(ops.block_rms(ctx, raw, 1000), np.sqrt((raw[:, :100000].astype(np.float64).reshape(5, 100, 1000) ** 2).mean(-1))) < TOL
    with pytest.raises(XcxError):
        ops.block_rms(ctx, raw, 1000, 1000, 10 ** 6)


def test_operation_list_like_scripts_cc(ctx, raw):
    """scripts/cc.py:46-113 style chain on a 250 Hz record."""
    chain = [
        ('detrend', {'type': 'demean'}),
        ('filter', {'type': 'highpass', 'freq': .05}),
        ('decimate', {'factor': 5}),
        ('filter', {'type': 'lowpass', 'freq': 20.}),
        ('detrend', {'type': 'demean'}),
        ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.}),
    ]
    y, fs = stream.apply_operations(raw, 250., chain, ctx=ctx)
    r = R.demean(raw)
    r = R.sosfilt(stream.butter_sos('highpass', 250., 4, freq=.05), r)
    r = R.decimate(r, 250., 5)
    r = R.sosfilt(stream.butter_sos('lowpass', 50., 4, freq=20.), r)
    r = R.demean(r)
    r = r * R.cosine_taper_obspy(r.shape[-1], 50., 0.05, 30.)
    assert fs == 50. and y.shape == r.shape
    assert relerr(y, r) < 5 * TOL                       # five chained float32 stages
    chain2 = [('interpolate', {'sampling_rate': 50., 'method': 'lanczos', 'a': 20})]
    y2, fs2 = stream.apply_operations(raw[:, :30000], 250., chain2, ctx=ctx)
    assert fs2 == 50. and y2.shape[1] == 6000
    with pytest.raises(NotImplementedError):
        stream.apply_operations(raw, 250., [('remove_response', {})], ctx=ctx)


def test_bhr_operation_list_of_scripts_cc(ctx):
    """The BHR list of xcorr/scripts/cc.py:70-90 minus `remove_response` (needs an ObsPy inventory), on a component
    set {BHZ, BHN, BHE} at 40 Hz: merge -> demean -> highpass -> rotate(->ZNE) -> rotate(NE->RT, 250.39 deg) ->
    select(BHR) -> Lanczos to 50 Hz -> lowpass -> trim -> demean -> cosine taper, host rows and device-resident
    rows, against the oracle restatement.  Unverified against obspy (no obspy here), like every a13 op."""
    rng = np.random.default_rng(31)
    npts = 40 * 900
    comps = {c: (rng.standard_normal(npts).cumsum() * 0.01 + rng.standard_normal(npts)).astype(np.float32)
             for c in ('BHZ', 'BHN', 'BHE')}
    baz = 250.39
    ops_list = [
        ('merge', {'method': 1, 'fill_value': 'interpolate', 'interpolation_samples': 0}),
        ('detrend', {'type': 'demean'}),
        ('filter', {'type': 'highpass', 'freq': .05}),
        ('rotate', {'method': '->ZNE'}),
        ('rotate', {'method': 'NE->RT', 'back_azimuth': baz}),
        ('select', {'channel': 'BHR'}),
        ('interpolate', {'sampling_rate': 50, 'method': 'lanczos', 'a': 20}),
        ('filter', {'type': 'lowpass', 'freq': 20.}),
        ('trim', {}),
        ('detrend', {'type': 'demean'}),
        ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.}),
    ]
    t0, t1 = 100.0, 100.0 + 800.0
    y, fs, ts = stream.apply_operations(comps, 40., ops_list, ctx=ctx, starttime=t0 + 20., endtime=t1,
                                        t0=t0, return_t0=True)
    # oracle: the same list in float64
    hp = stream.butter_sos('highpass', 40., 4, freq=.05)
    pre = {c: R.sosfilt(hp, R.demean(v.astype(np.float64)[None]))[0] for c, v in comps.items()}
    ba = np.radians(baz)
    r = -pre['BHE'] * np.sin(ba) - pre['BHN'] * np.cos(ba)
    span = (npts - 1) / 40.
    nnew = int(np.floor(span * 50.)) + 1
    r = R.lanczos_interpolation(r[None], 0.0, 40. / 50., nnew, 20)[0]
    r = R.sosfilt(stream.butter_sos('lowpass', 50., 4, freq=20.), r[None])[0]
    i0 = int(round((t0 + 20. - t0) * 50.))
    i1 = int(round((t1 - t0) * 50.))
    r = r[i0:i1 + 1]
    r = R.demean(r[None])[0]
    r = r * R.cosine_taper_obspy(r.size, 50., 0.05, 30.)
    assert fs == 50. and abs(ts - (t0 + 20.)) < 1e-9 and y.shape == r.shape
    assert relerr(y, r) < 5 * TOL
    # device-resident components: same numbers, the result stays on the device
    dcomps = {c: ctx.array(v[None]) for c, v in comps.items()}
    yd, fsd = stream.apply_operations(dcomps, 40., ops_list[1:], ctx=ctx, starttime=t0 + 20., endtime=t1, t0=t0)
    assert hasattr(yd, 'to_host') and fsd == 50.
    assert relerr(yd.to_host()[0], r) < 5 * TOL
    # the rotation alone, both directions of storage, against the defining formula
    n32, e32 = comps['BHN'], comps['BHE']
    rr, tt = stream.rotate_ne_rt(n32, e32, 33.0, ctx=ctx)
    b = np.radians(33.0)
    assert relerr(rr, -e32.astype(np.float64) * np.sin(b) - n32.astype(np.float64) * np.cos(b)) < 1e-7
    assert relerr(tt, -e32.astype(np.float64) * np.cos(b) + n32.astype(np.float64) * np.sin(b)) < 1e-7
    assert np.array_equal(n32, comps['BHN'])                  # host inputs are not modified


def test_normalize_propagates_nan_on_device_rows_like_the_host_path(ctx):
    """a gappy row (NaN sample) normalises to NaN on the host path (np.abs(x).max()); device rows agree."""
    x = np.random.default_rng(5).standard_normal((3, 70000)).astype(np.float32)
    x[1, 12345] = np.nan
    m = ops.absmax_rows(ctx, ctx.array(x))
    assert np.isnan(m[1]) and m[0] == np.abs(x[0]).max() and m[2] == np.abs(x[2]).max()
    assert np.isnan(ops.absmax_rows(ctx, x)[1])
    yd = stream.normalize(ctx.array(x)).to_host()
    yh = stream.normalize(x)
    assert np.isnan(yd[1]).all() and np.isnan(yh[1]).all()
    assert np.allclose(yd[0], yh[0], rtol=1e-6) and np.allclose(yd[2], yh[2], rtol=1e-6)


def test_device_resident_rows(ctx, raw):
    d = ctx.array(raw)
    y = ops.decimate(ctx, d, 5, stream.cheby2_lowpass_sos(25., 250.))
    assert relerr(y.to_host(), R.decimate(raw, 250., 5)) < TOL
    assert np.array_equal(d.to_host(), raw)            # input untouched
    z = ops.lanczos(ctx, d, 1000, 3.25, 5.0, 20)
    assert relerr(z.to_host(), R.lanczos_interpolation(raw, 3.25, 5.0, 1000, 20)) < TOL


def test_operation_list_on_device_resident_rows(ctx, raw):
    """The whole list of scripts/cc.py's EDH channel (without remove_response) with the rows staying on the device:
    trim (a column slice), remove_sensitivity and normalize (row scaling) included -- against the same list on the
    host arrays, op for op the same kernels, and against the oracle."""
    chain = [
        ('detrend', {'type': 'demean'}),
        ('filter', {'type': 'highpass', 'freq': .05}),
        ('remove_sensitivity', {}),
        ('decimate', {'factor': 5}),
        ('trim', {}),
        ('detrend', {'type': 'demean'}),
        ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.}),
        ('normalize', {}),
    ]
    n = raw.shape[-1]
    t1, t2 = 3.0, (n - 1) / 250. - 5.0                  # cut 3 s at the start and 5 s at the end
    kw = dict(ctx=ctx, t0=0.0, starttime=t1, endtime=t2, sensitivity=6.1e8, return_t0=True)
    yh, fh, t0h = stream.apply_operations(raw, 250., chain, **kw)
    d = ctx.array(raw)
    yd, fd, t0d = stream.apply_operations(d, 250., chain, **kw)
    assert fd == fh == 50. and t0d == t0h and tuple(yd.shape) == yh.shape
    got = yd.to_host()
    assert relerr(got, yh) < TOL
    assert np.allclose(np.abs(got).max(axis=-1), 1.0, rtol=1e-6)      # normalize: unit absolute maximum per row
    r = R.demean(raw)
    r = R.sosfilt(stream.butter_sos('highpass', 250., 4, freq=.05), r) / 6.1e8
    r = R.decimate(r, 250., 5)
    i0 = int(round(t1 * 50.)); i1 = int(round(t2 * 50.)) + 1
    r = R.demean(r[:, i0:i1])
    r = r * R.cosine_taper_obspy(r.shape[-1], 50., 0.05, 30.)
    r = r / np.abs(r).max(axis=-1, keepdims=True)
    assert got.shape == r.shape and relerr(got, r) < 5 * TOL
    # the row ops by themselves
    x = (np.random.default_rng(5).standard_normal((7, 50001)) * 3).astype(np.float32)
    dx = ctx.array(x)
    assert np.array_equal(ops.absmax_rows(ctx, dx), np.abs(x).max(axis=-1))
    f = np.linspace(0.5, 2.0, 7)
    assert np.array_equal(ops.scale_rows(ctx, dx, f).to_host(), (x.astype(np.float64) * f[:, None]).astype(np.float32))
    assert np.array_equal(ops.scale_rows(ctx, x, 0.25), x * np.float32(0.25))          # host rows, scalar factor


def test_filter_zerophase_is_obspys_two_pass_form(ctx, raw):
    """Trace.filter(zerophase=True) = sosfilt forward from rest, then sosfilt of the
    reversed signal from rest -- NOT scipy's sosfiltfilt (odd padding + steady-state start)."""
    x = raw[:, :60001]
    for kw in (dict(type='bandpass', freqmin=1., freqmax=20.), dict(type='highpass', freq=.5)):
        sos = stream.butter_sos(kw['type'], 250., 4, **{k: v for k, v in kw.items() if k != 'type'})
        y = stream.filter(x, 250., zerophase=True, ctx=ctx, **kw)
        ref = R.sosfilt_zerophase(sos, x)
        assert relerr(y, ref) < 2 * TOL                 # two float32 passes
        # and it does differ from sosfiltfilt at the edges
        assert relerr(R.sosfiltfilt(sos, x.astype(np.float64)), ref) > 1e-3
    d = ctx.array(x)                                    # device rows, day-long chunked scan path
    xl = np.ascontiguousarray(np.tile(x[:2], (1, 9))[:, :500001])
    sos = stream.butter_sos('bandpass', 250., 4, freqmin=1., freqmax=20.)
    assert relerr(ops.sosfilt_zerophase(ctx, xl, sos), R.sosfilt_zerophase(sos, xl)) < 2 * TOL
    assert relerr(ops.sosfilt_zerophase(ctx, d, sos).to_host(), R.sosfilt_zerophase(sos, x)) < 2 * TOL


@pytest.mark.parametrize('new_rate', [50., 100., 62.5, 40., 500.])
def test_fourier_resample(ctx, raw, new_rate):
    """stream.resample = ObsPy Trace.resample (Fourier method; registry entry 'resample',
    xcorr/stream/process.py:29-94) against the numpy restatement; odd / even lengths, integer and
    non-integer factors, window types, the anti-alias option, device rows, the operation list."""
    for npts in (20000, 20001, 7919):
        x = raw[:3, :npts]
        y, fs = stream.resample(x, 250., new_rate, ctx=ctx)
        ref = R.resample_fourier(x, 250., new_rate)
        assert fs == new_rate and y.shape == ref.shape == (3, int(npts / (250. / new_rate)))
        assert relerr(y, ref) < TOL
    x = raw[:2, :12000]
    for w in (None, 'blackman'):
        assert relerr(stream.resample(x, 250., new_rate, window=w, ctx=ctx)[0], R.resample_fourier(x, 250., new_rate, window=w)) < TOL
    d = ctx.array(x)
    import scipy.signal
    wh = np.fft.ifftshift(scipy.signal.get_window('hann', 12000))[:6001]
    yd = ops.resample(ctx, d, int(12000 / (250. / new_rate)), 1.0, wh)       # 12000 / factor is whole: bins coincide
    assert relerr(yd.to_host(), R.resample_fourier(x, 250., new_rate)) < TOL
    if new_rate <= 100.:
        yf, _ = stream.resample(x, 250., new_rate, no_filter=False, ctx=ctx)
        xf = R.sosfilt(stream.cheby2_lowpass_sos(0.5 * new_rate, 250.), x.astype(np.float64))
        assert relerr(yf, R.resample_fourier(xf, 250., new_rate)) < 5 * TOL
        y2, fs2 = stream.apply_operations(x, 250., [('resample', {'sampling_rate': new_rate})], ctx=ctx)
        assert fs2 == new_rate and relerr(y2, R.resample_fourier(x, 250., new_rate)) < TOL
    with pytest.raises(ValueError):
        stream.resample(raw[:1, :1001], 250., 100., strict_length=True, ctx=ctx)


def test_reference_edh_chain_with_merge_and_trim(ctx):
    """xcorr/scripts/cc.py:96-113 (EDH): merge(method 1, interpolate) -> demean -> highpass ->
    decimate(5) -> trim to the window -> demean -> taper; two segments with a 40-sample gap,
    window bounds injected like stream/process.py:153-196."""
    rng = np.random.default_rng(5)
    fs = 250.
    full = (rng.standard_normal(61000) + 2.0).astype(np.float32)
    t_start = 1000.0
    segs = [(full[30020:], t_start + 30020 / fs), (full[:29980], t_start)]      # out of order, gap 29980..30019
    chain = [
        ('merge', {'method': 1, 'fill_value': 'interpolate', 'interpolation_samples': 0}),
        ('detrend', {'type': 'demean'}),
        ('filter', {'type': 'highpass', 'freq': .05}),
        ('decimate', {'factor': 5}),
        ('trim', {}),
        ('detrend', {'type': 'demean'}),
        ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.}),
    ]
    w0, w1 = t_start + 4.0, t_start + 204.0
    y, fs2, t0 = stream.apply_operations(segs, fs, chain, ctx=ctx, starttime=w0, endtime=w1, return_t0=True)
    ref = full.astype(np.float64).copy()
    ref[29980:30020] = np.linspace(full[29979], full[30020], 42)[1:-1]
    r = R.demean(ref[None])
    r = R.sosfilt(stream.butter_sos('highpass', fs, 4, freq=.05), r)
    r = R.decimate(r, fs, 5)
    r = r[:, 200:200 + 10001]                       # 4 s .. 204 s at 50 Hz, both ends included
    r = R.demean(r)
    r = r * R.cosine_taper_obspy(r.shape[-1], 50., 0.05, 30.)
    assert fs2 == 50. and abs(t0 - w0) < 1e-9 and y.shape[-1] == 10001
    assert relerr(np.asarray(y).reshape(1, -1), r) < 5 * TOL
    with pytest.raises(ValueError):                 # a gap without fill_value leaves masked samples
        stream.apply_operations(segs, fs, [('merge', {'method': 1})], ctx=ctx)


def test_resample_long_rows(ctx):
    """Fourier resample beyond one chirp convolution (npts 500 000 -> 200 000): split transforms."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((1, 500000)).astype(np.float32)
    y, fs2 = stream.resample(x, 250., 100., ctx=ctx)
    ref = R.resample_fourier(x, 250., 100.)
    assert fs2 == 100. and y.shape == ref.shape == (1, 200000)
    assert relerr(y, ref) < TOL
