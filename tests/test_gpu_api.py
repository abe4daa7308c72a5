"""GPU: the host-side mirrors of the reference interface (util.cc.cc,
signal.*, core.process) -- these read like the reference's own call sites."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
import xcorr_amd.util.cc as ucc            # noqa: E402
import xcorr_amd.signal as sig             # noqa: E402
import xcorr_amd.core as core              # noqa: E402

TOL = 1e-5


def relerr(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300))


def test_util_cc_signature_and_values(ctx, golden):
    x, y = golden['x_1000'], golden['y_1000']
    r = ucc.cc(x, y, normalize=True, pad=True, unbiased=False)
    assert r.dtype == np.float32 and r.shape == (1999,)
    assert relerr(r, golden['cc_1000_norm1']) < TOL
    r64 = ucc.cc(x, y, normalize=False, pad=True, unbiased=False, dtype=np.float64)
    assert r64.dtype == np.float64 and relerr(r64, golden['cc_1000_norm0']) < TOL
    # pad=False: circular correlation with fftshift ordering (util/cc.py:72-78)
    assert relerr(ucc.cc(x, y, normalize=False, pad=False, unbiased=False),
                  R.cc(x, y, normalize=False, pad=False)) < TOL
    xs = golden['kat_x'].astype(np.float32)
    ys = golden['kat_y'].astype(np.float32)
    np.testing.assert_allclose(ucc.cc(xs, ys, False, False, False), golden['kat_nopad'], atol=1e-5)
    # unbiased=True (default): closed-form weights over all 2n-1 lags
    ru = ucc.cc(x, y, normalize=True)
    assert relerr(ru / R.weight_full(1000), golden['cc_1000_norm1']) < TOL
    s, m = ucc.compute_shift_and_max(x, y, delta=0.02, unbiased=False)
    sr, mr = ucc.extract_shift_and_max(golden['cc_1000_norm1'], 0.02)
    assert s == sr and abs(m - mr) < TOL


def test_signal_functions(ctx):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 4, 600)).astype(np.float32) + 2
    b = rng.standard_normal((3, 4, 600)).astype(np.float32)
    c = sig.correlate1d(a, b)
    assert c.shape == (3, 4, 1199)
    assert relerr(c, R.correlate1d(a, b, dtype=np.float64)) < TOL
    assert relerr(sig.correlate1d(a, b, normalize=False), R.correlate1d(a, b, False, np.float64)) < TOL
    with pytest.raises(ValueError, match='same shape'):
        sig.correlate1d(a, b[:, :, :-1])
    assert relerr(sig.norm1d(a), R.norm1d(a)) < TOL
    assert relerr(sig.demean(a), R.demean(a)) < TOL
    assert relerr(sig.detrend(a, type='linear'), R.detrend_linear(a)) < 2 * TOL
    assert relerr(sig.taper(a, max_length=2., sampling_rate=50.), R.taper(a, 50., max_length=2.)) < TOL
    assert relerr(sig.filter(a, 3., btype='highpass', order=2, sampling_rate=50.),
                  R.sosfiltfilt(R.butter_sos(2, 3., 'highpass', 50.), a)) < TOL
    assert relerr(sig.filter(a, (1., 5.), btype='bandpass', order=4, sampling_rate=50.),
                  R.sosfiltfilt(R.butter_sos(4, (1., 5.), 'bandpass', 50.), a)) < TOL
    cc = R.correlate1d(a, b)
    assert relerr(sig.unbias(cc, npts=600), R.unbias(cc, 600)) < 1e-6


def test_core_process_matches_reference_loop(ctx):
    """init + process on an in-memory client == the reference's double loop
    of util.cc.cc calls (core/process.py:82-165), incl. status bookkeeping."""
    fs, wl = 50., 40.
    n = int(wl * fs)
    S, T = 3, 9000
    tr = R.synthetic_traces(S, T)
    names = ['A', 'B', 'C']
    client = core.ArrayClient(dict(zip(names, tr)), fs, t0=0.)
    pairs = ['A-B', 'A-C', 'B-C']
    ds = core.init(pairs, 0., T / fs, fs, window_length=wl, window_overlap=.5, clip_lag=(-5., 5.))
    imin, imax = ds.lag_attrs['index_min'], ds.lag_attrs['index_max']
    core.process(ds, client)
    nproc = 0
    for ip, p in enumerate(pairs):
        i, j = names.index(p[0]), names.index(p[2])
        for it, t in enumerate(ds.time):
            i0 = int(round((t - wl / 2) * fs))
            if i0 < 0 or i0 + n > T:
                assert ds.status[ip, it] == -1 and np.all(ds.cc[ip, it] == 0)
                continue
            assert ds.status[ip, it] == 1
            ref = R.cc(tr[i, i0:i0 + n], tr[j, i0:i0 + n], normalize=True, dtype=np.float64)[imin:imax]
            assert relerr(ds.cc[ip, it], ref) < TOL
            assert ds.pair_offset[ip, it] == 0 and abs(ds.time_offset[ip, it]) < 1e-9
            nproc += 1
    assert nproc == 3 * 8
    # resume: nothing left to do; retry_missing re-tries the -1 steps
    before = ds.cc.copy()
    core.process(ds, client)
    np.testing.assert_array_equal(ds.cc, before)
    core.process(ds, client, retry_missing=True)
    assert (ds.status == -1).sum() == 3 * (len(ds.time) - 8)
    # bias correction at the end (core/process.py:178-180)
    ds2 = core.init(pairs, 0., T / fs, fs, wl, .5, unbiased_cc=True)
    core.process(ds2, client)
    ok = ds2.status[0] == 1
    ref = R.cc(tr[0, 0:n], tr[1, 0:n], dtype=np.float64)
    it = int(np.argmax(ok))
    assert relerr(ds2.cc[0, it] / R.weight_full(n), ref) < TOL
    assert ds2.cc_attrs['unbiased'] == 1


def test_process_batched_equals_process(ctx):
    fs, wl = 50., 60.
    n = int(wl * fs)
    tr = R.synthetic_traces(4, 5 * n // 2 + n)
    out, lag = core.process_batched(tr, fs, wl, 0.5, clip_lag=(-10., 10.), demean=True)
    assert out.shape == (6, 6, 1000) and lag.shape == (1000,)
    imin = 179499 - (180000 - n)
    ref = R.cc_pairs(tr, R.all_pairs(4), n, n // 2, demean_=True, dtype=np.float64,
                     index_min=n - 1 - 500, index_max=n - 1 + 500)
    assert relerr(out, ref) < TOL


def test_process_hashes_waveforms(ctx):
    """core/process.py:78-79,146-149: sha256 of the two preprocessed traces per processed step."""
    from xcorr_amd.util import hasher
    fs, wl = 50., 40.
    tr = R.synthetic_traces(2, 7000)
    client = core.ArrayClient({'IM.H10N1..EDH': tr[0], 'IU.RAR.10.BHZ': tr[1]}, fs)
    pair = 'IM.H10N1..EDH-IU.RAR.10.BHZ'
    ds = core.init(pair, 0., 7000 / fs, sampling_rate=fs, window_length=wl, window_overlap=.5, hash_waveforms=True)
    core.process(ds, client, hash_waveforms=True)
    done = ds.status[0] == 1
    assert done.any() and all(len(h) == 64 for h in ds.hash[0][done]) and (ds.hash[0][~done] == 'n/a').all()
    it = int(np.flatnonzero(done)[0])
    (xa, ta), (xb, tb) = client.get_pair_processed_waveforms(pair=pair, time=ds.time[it], duration=wl, sampling_rate=fs)
    want = hasher.hash_stream([(hasher.trace_stats('IM.H10N1..EDH', ta, fs, len(xa)), xa),
                               (hasher.trace_stats('IU.RAR.10.BHZ', tb, fs, len(xb)), xb)])
    assert ds.hash[0, it] == want and len(set(ds.hash[0][done])) == done.sum()
    ds2 = core.init(pair, 0., 7000 / fs, sampling_rate=fs, window_length=wl, window_overlap=.5)
    core.process(ds2, client, hash_waveforms=True)                 # no hash variable: nothing to fill
    assert ds2.hash is None and np.array_equal(ds2.cc, ds.cc)


def test_signal_timeshift_and_core_postprocess(ctx):
    """signal.timeshift (array form) and the whole core.postprocess chain on a
    processed dataset, against the oracle chain (core/postprocess.py:133-141)."""
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((2, 3, 801)) * np.hanning(801)).astype(np.float32)
    d = rng.uniform(-0.05, 0.05, (2, 3))
    y = sig.timeshift(a, d, sampling_rate=50.)
    assert y.shape == (2, 3, 800) and y.dtype == np.float32
    assert relerr(y, R.timeshift(a, d, 50.)) < TOL
    yf = sig.timeshift(a, d, sampling_rate=50., fast=False)        # fft/ifft path keeps all 801 samples
    assert yf.shape == (2, 3, 801) and relerr(yf, R.timeshift_fft(a, d, 1 / 50.)) < TOL
    with pytest.raises(TypeError):
        sig.timeshift(a, 1, sampling_rate=50.)
    # process + postprocess on an in-memory client with sub-sample start offsets
    fs, wl = 50., 40.
    n = int(wl * fs)
    tr = R.synthetic_traces(3, 9000)
    names = ['A', 'B', 'C']
    client = core.ArrayClient(dict(zip(names, tr)), fs)
    ds = core.init(['A-B', 'B-C'], 0., 9000 / fs, fs, window_length=wl, window_overlap=.5)
    core.process(ds, client)
    ds.pair_offset[...] = rng.uniform(-0.01, 0.01, ds.pair_offset.shape)    # as if starts were offset
    pp = core.postprocess(ds, lag_lim=(-20., 20.), filter_kwargs={'frequency': 3., 'order': 2})
    valid = ds.status == 1
    lag = ds.lag
    m = (lag >= -20.) & (lag <= 20.)
    i0 = int(np.flatnonzero(m)[0]); i1 = int(np.flatnonzero(m)[-1]) + 1
    rows = ds.cc[valid][:, i0:i1]
    delay = -(ds.pair_offset + ds.time_offset)[valid]
    ref = R.postprocess_chain(rows, n, fs, i0, i1, delay)
    assert pp.cc.shape[-1] == ref.shape[-1] == (i1 - i0) - ((i1 - i0) & 1)
    assert relerr(pp.cc[valid], ref) < TOL
    assert np.all(pp.cc[~valid] == 0) and pp.cc_attrs['unbiased'] == 1


def test_bench_contract_small(ctx):
    """bench.py prints ONE JSON line with the contract's keys (tiny workload)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--stations', '6', '--seconds', '14400',
                          '--steps', '1', '--warmup', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 1 and d['vs_baseline'] is None and d['value'] > 0
    assert d['config']['workload'].startswith('6 stations') and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'frac_end_to_end',
              'frac_pair_phase', 'frac_of_measured_copy'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and 0 < r['frac'] < 1
    # PMC traffic belongs to the profiled workload only: a 6-station run must not inherit it
    assert r['traffic'] is None and 'workload' in r['traffic_source'] or 'build' in r['traffic_source'] \
        or 'no profiles' in r['traffic_source']
    assert len(d['lib_sha256']) == 64


def test_bench_stack_days_leaves_a_stack_behind(ctx):
    """bench.py --stack --days 2 on one rank: the accumulate loop's result survives the step (an `else:` on
    the wrong block once zeroed it after every successful stack_finish) -- the line carries `stack_check`
    and the bench refuses to print a line for an all-zero stack."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--stations', '6', '--seconds', '14400',
                          '--stack', '--days', '2', '--steps', '1', '--warmup', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    chk = d['stack_check']
    assert chk['row0_absmax'] > 0 and chk['row0_argmax'] == chk['row0_argmax_expected']
    assert d['config']['station_days'] == 2


def test_prepare_pipelines_the_forward_transforms(ctx):
    """xcx_cc1d_prepare: spectra computed ahead on a side stream are consumed by the
    matching pairs() call (bit-identical results); a non-matching call discards them."""
    import numpy as np
    from xcorr_amd.engine import CCPlan
    from xcorr_amd._ffi import XcxError
    rng = np.random.default_rng(77)
    S, n, hop, W = 6, 3000, 1500, 5
    T = (W - 1) * hop + n
    pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
    a = rng.standard_normal((S, T)).astype(np.float32)
    b = rng.standard_normal((S, T)).astype(np.float32)
    plan = CCPlan(ctx, n, normalize=True, demean=True)
    da, db = ctx.array(a), ctx.array(b)
    ld = (plan.L + 31) // 32 * 32
    out = ctx.empty((len(pairs) * W, ld))
    ref_a = plan.pairs(da, pairs, W, hop, out=out, ld_out=ld).to_host().copy()
    ref_b = plan.pairs(db, pairs, W, hop, out=out, ld_out=ld).to_host().copy()
    # pipelined: prepare b while a's pair phase runs, and so on
    plan.prepare(da, W, hop)
    got_a = plan.pairs(da, pairs, W, hop, out=out, ld_out=ld)
    plan.prepare(db, W, hop)                       # overlaps with the pair phase of a
    got_a = got_a.to_host().copy()
    got_b = plan.pairs(db, pairs, W, hop, out=out, ld_out=ld).to_host().copy()
    assert np.array_equal(got_a[:, :plan.L], ref_a[:, :plan.L])
    assert np.array_equal(got_b[:, :plan.L], ref_b[:, :plan.L])
    # stale preparation (other traces) is dropped, the result is that of the call's own traces
    plan.prepare(da, W, hop)
    got = plan.pairs(db, pairs, W, hop, out=out, ld_out=ld).to_host()
    assert np.array_equal(got[:, :plan.L], ref_b[:, :plan.L])
    # overwrite the traces only after prepare_sync
    plan.prepare(da, W, hop)
    plan.prepare_sync()
    da.copy_from(b)
    got = plan.pairs(da, pairs, W, hop, out=out, ld_out=ld).to_host()
    assert np.array_equal(got[:, :plan.L], ref_a[:, :plan.L])      # spectra of the OLD contents were prepared
    with pytest.raises(XcxError):
        CCPlan(ctx, n, stack=True).prepare(da, W, hop)
    with pytest.raises(ValueError):
        plan.prepare(a, W, hop)                    # host array
    # an UNPREPARED call (its forward transforms run on the main stream) followed at once by a preparation for the
    # next traces (forward transforms on the side stream): the two share the plan's transform workspace and must
    # not run at the same time -- sizes at which a forward pass takes about a millisecond
    S2, n2, hop2, W2 = 48, 20000, 10000, 24
    T2 = (W2 - 1) * hop2 + n2
    p2 = np.array([(0, 1), (5, 40), (47, 2), (13, 13), (30, 31), (8, 25)], np.int32)
    a2 = ctx.array(rng.standard_normal((S2, T2)).astype(np.float32))
    b2 = ctx.array(rng.standard_normal((S2, T2)).astype(np.float32))
    big = CCPlan(ctx, n2, normalize=True, demean=True)
    oa, ob = ctx.empty((len(p2) * W2, big.L)), ctx.empty((len(p2) * W2, big.L))
    ra = big.pairs(a2, p2, W2, hop2, out=oa).to_host().copy()
    rb = big.pairs(b2, p2, W2, hop2, out=ob).to_host().copy()
    for _ in range(3):
        big.pairs(a2, p2, W2, hop2, out=oa)        # unprepared, asynchronous
        big.prepare(b2, W2, hop2)                  # must wait for that forward pass, not for its pair phase
        big.pairs(b2, p2, W2, hop2, out=ob)        # consumes the preparation
        assert np.array_equal(oa.to_host(), ra) and np.array_equal(ob.to_host(), rb)
    big.close()


def test_column_views_select_window_blocks(ctx):
    """DeviceArray.columns_from: a rank that owns windows [w0, w1) of a day passes a view that
    starts at its first window (bench.py --shard windows)."""
    import numpy as np
    from xcorr_amd.engine import CCPlan
    rng = np.random.default_rng(5)
    S, n, hop, W = 4, 2000, 1000, 7
    T = (W - 1) * hop + n
    a = rng.standard_normal((S, T)).astype(np.float32)
    pairs = np.array([(0, 1), (2, 3), (1, 3)], np.int32)
    plan = CCPlan(ctx, n)
    d = ctx.array(a)
    ld = (plan.L + 31) // 32 * 32
    full = plan.pairs(d, pairs, W, hop, out=ctx.empty((3 * W, ld)), ld_out=ld).to_host().reshape(3, W, ld)
    w0, Wr = 2, 3
    v = d.columns_from(w0 * hop)
    assert v.shape == (S, T - w0 * hop)
    part = plan.pairs(v, pairs, Wr, hop, out=ctx.empty((3 * Wr, ld)), ld_out=ld, stride_s=T).to_host().reshape(3, Wr, ld)
    assert np.array_equal(part[..., :plan.L], full[:, w0:w0 + Wr, :plan.L])
    v.free()
    assert d.ptr                                   # the view frees nothing
    with pytest.raises(ValueError):
        d.columns_from(T)


def test_detrend_skipna_and_window_types(ctx):
    """signal.detrend(skipna=True) (signal/detrend.py:87-110: fit on the finite samples, NaN stays
    NaN) and taper with any scipy window type (signal/window.py:92)."""
    import scipy.signal.windows as W
    import scipy.stats
    rng = np.random.default_rng(41)
    x = (rng.standard_normal((6, 4001)) + 2.0 + 1e-3 * np.arange(4001)).astype(np.float32)
    x[1, 17] = np.nan
    x[2, 100:300] = np.nan
    x[5, :] = np.nan
    got = sig.detrend(x, type='constant')                        # skipna=True is the default
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                          # row 5 is all NaN
        ref = x.astype(np.float64) - np.nanmean(x.astype(np.float64), axis=-1, keepdims=True)
    ok = ~np.isnan(x)
    assert np.array_equal(np.isnan(got), np.isnan(x))
    assert np.abs(got[ok] - ref[ok]).max() < TOL * 5
    got = sig.detrend(x, type='linear')
    t = np.arange(4001, dtype=np.float64)
    for r in range(5):
        m = ok[r]
        sl = scipy.stats.linregress(t[m], x[r, m].astype(np.float64))
        assert np.abs(got[r, m] - (x[r, m] - (sl.slope * t[m] + sl.intercept))).max() < TOL * 5
        assert np.all(np.isnan(got[r, ~m]))
    assert np.all(np.isnan(got[5]))
    # skipna=False: a NaN poisons its row (x.mean(skipna=False))
    g2 = sig.detrend(x, type='constant', skipna=False)
    assert np.all(np.isnan(g2[1])) and not np.any(np.isnan(g2[0]))
    # window types
    y = rng.standard_normal((3, 3000)).astype(np.float32)
    for wt in ('hann', 'hamming', 'blackman', 'bartlett', 'cosine'):
        got = sig.taper(y, wtype=wt, max_length=5., sampling_rate=50.)
        w = np.ones(3000)
        win = getattr(W, wt)(500, False)
        w[:250], w[-250:] = win[:250], win[-250:]
        assert relerr(got, y * w) < TOL, wt
        assert np.allclose(sig.window(3000, 50., wtype=wt, max_length=5.), w)
    with pytest.raises(ValueError):
        sig.taper(y, wtype='no_such_window', max_length=5., sampling_rate=50.)


def test_dataarray_accessor_every_method(ctx):
    """B4 seam: every method of the re-registered ``.signal`` accessor on real xarray objects
    (xcorr/signal/accessor.py:60-321).  Runs wherever xarray is importable."""
    xr = pytest.importorskip('xarray')
    import xcorr_amd.signal.accessor  # noqa: F401  (registers .signal)
    rng = np.random.default_rng(43)
    n, fs = 1000, 50.
    lag = ucc.lag(n, 1 / fs)
    a = rng.standard_normal((2, 3, n)).astype(np.float32)
    t = np.arange(n) / fs
    da = xr.DataArray(a, dims=('pair', 'time', 't'), coords={'t': ('t', t, {'sampling_rate': fs})},
                      name='x', attrs={'long_name': 'x', 'units': '-'})
    assert relerr(da.signal.demean().values, R.demean(a)) < TOL
    assert relerr(da.signal.detrend(type='linear').values, R.detrend_linear(a)) < TOL
    assert relerr(da.signal.norm1d().values, R.norm1d(a)) < TOL
    assert relerr(da.signal.taper(max_length=2.).values, R.taper(a, fs, max_length=2.)) < TOL
    assert relerr(da.signal.filter(3., btype='highpass', order=2).values,
                  R.sosfiltfilt(R.butter_sos(2, 3., 'highpass', fs), a)) < TOL
    assert np.array_equal(da.signal.abs().values, np.abs(a)) and np.array_equal(da.signal.absolute().values, np.abs(a))
    assert relerr(da.signal.rms().values, np.sqrt((a.astype(np.float64) ** 2).mean(-1))) < TOL
    assert relerr(np.abs(da.signal.hilbert().values), R.envelope(a)) < 5 * TOL
    assert relerr(da.signal.envelope().values, R.envelope(a)) < 5 * TOL
    cc = da.signal.correlate1d(da)
    assert cc.shape[-1] == 2 * n - 1 and abs(float(cc.values[0, 0, n - 1]) - 1.0) < TOL
    ts = da.signal.timeshift(delay=0.013)
    assert relerr(ts.values, R.timeshift(a.astype(np.float64), 0.013, fs)) < 5 * TOL
    sm = (t > 2) & (t < 6)
    nm = (t > 10) & (t < 18)
    r = da.signal.snr(xr.DataArray(sm, dims='t'), xr.DataArray(nm, dims='t'))
    ro, _, _, _ = R.snr(a, sm, nm)
    assert r.dims == ('pair', 'time') and relerr(r.values, ro) < TOL
    ext = da.signal.snr(xr.DataArray(sm, dims='t'), xr.DataArray(nm, dims='t'), extend=True)
    assert set(ext.data_vars) == {'snr', 's', 'n', 't_s_max'}
    w = da.t.signal.window(max_length=2.)
    assert np.allclose(w.values, R.window(n, fs, max_length=2.))
    ccd = xr.DataArray(rng.standard_normal((1, 2, 2 * n - 1)).astype(np.float32), dims=('pair', 'time', 'lag'),
                       coords={'lag': ('lag', lag, {'sampling_rate': fs, 'delta': 1 / fs, 'npts': n, 'index_min': 0,
                                                    'index_max': 2 * n - 1})}, attrs={'unbiased': np.byte(0)})
    uw = ccd.signal.unbias_weights()
    assert np.allclose(uw.values, R.weight_full(n))
    ub = ccd.signal.unbias()
    assert ub.attrs['unbiased'] == 1 and relerr(ub.values, ccd.values * R.weight_full(n)) < TOL
    assert relerr(da.signal.fft().signal.ifft().values.real, a) < 5 * TOL
    assert relerr(da.signal.rfft().signal.irfft().values, a) < 5 * TOL
    sp = da.isel(pair=0, time=0).signal.spectrogram(duration=1., padding_factor=2)
    assert sp.shape[-1] == n
    b2 = xr.DataArray(np.abs(rng.standard_normal((8, 64))).astype(np.float32), dims=('f', 'l'))
    assert relerr(b2.signal.norm2d().values, R.norm2d(b2.values)) < TOL
    assert b2.signal.correlate2d(b2).shape == (15, 127)


def test_lazy_single_process_writes_resumes_and_skips(ctx, tmp_path):
    """core.lazy.single_process (core/lazy.py:26-82): process a (pair, day) file, write it, and on the next
    call skip it; a file with missing steps is resumed with retry_missing and only the new steps run."""
    from xcorr_amd import io
    from xcorr_amd.core import lazy
    fs, wl = 50., 60.
    t0 = 1580688000.0                                            # 2020-02-03T00:00:00Z
    rng = np.random.default_rng(5)
    T = int(600 * fs)
    tr = {'NL.A..HHZ': rng.standard_normal(T).astype(np.float32), 'NL.B..HHZ': rng.standard_normal(T).astype(np.float32)}
    pair = 'NL.A..HHZ-NL.B..HHZ'
    args = dict(preprocess={}, attrs={'institution': 'i', 'author': 'a', 'source': 's'}, sampling_rate=fs,
                window_length=wl, window_overlap=.5, clip_lag=(-5., 5.))

    class Counting(core.ArrayClient):
        calls = 0

        def get_pair_processed_waveforms(self, **kw):
            Counting.calls += 1
            return super().get_pair_processed_waveforms(**kw)
    # the record covers only the first 10 minutes of the "day" (step = 20 min): later windows are missing
    cl = Counting(tr, fs, t0=t0)
    assert lazy.single_process(pair, t0, args, cl, str(tmp_path), step=1200.) is True
    ds = io.read(io.ncfile(pair, t0, str(tmp_path)))
    n_ok, n_missing = int((ds.status == 1).sum()), int((ds.status == -1).sum())
    assert n_ok >= 17 and n_missing >= 18 and n_ok + n_missing == ds.status.size
    ref = R.cc(tr['NL.A..HHZ'][1500:4500], tr['NL.B..HHZ'][1500:4500], normalize=True, dtype=np.float64)
    k = int(np.flatnonzero(ds.status[0] == 1)[1])
    la = ds.lag_attrs
    assert relerr(ds.cc[0, k], ref[la['index_min']:la['index_max']]) < TOL
    # resume with the full record: only the missing steps are fetched again
    tr2 = {k2: np.concatenate([v, rng.standard_normal(T).astype(np.float32)]) for k2, v in tr.items()}
    Counting.calls = 0
    assert lazy.single_process(pair, t0, args, Counting(tr2, fs, t0=t0), str(tmp_path), step=1200.) is True
    assert Counting.calls == n_missing
    ds2 = io.read(io.ncfile(pair, t0, str(tmp_path)))
    assert (ds2.status == 1).sum() > n_ok and np.array_equal(ds2.cc[0, k], ds.cc[0, k])


def test_example_script_runs(ctx, tmp_path):
    """examples/process_day.py: init -> process -> write -> mfread -> postprocess -> snr; the stack
    of every pair peaks at the synthetic moveout (25 samples = 0.5 s per station step)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('example_process_day', os.path.join(root, 'examples', 'process_day.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pp = mod.main(str(tmp_path))
    assert len(pp.pair) == 3 and (pp.status == 1).sum() >= 27
    for ip, pair in enumerate(pp.pair):
        i, j = (int(s.split('.')[1][3]) for s in str(pair).split('-'))
        stack = pp.cc[ip][pp.status[ip] == 1].mean(axis=0)
        peak = pp.lag[int(np.argmax(np.abs(stack)))]
        assert abs(peak - 0.5 * (i - j)) <= 0.04, (pair, peak)


def test_plain_c_caller(ctx, tmp_path):
    """examples/abi_demo.c: gcc-compiled C program against include/xcx.h + libxcx.so reproduces the
    util/cc.py KAT of SURVEY 8c and finds the lag of a delayed copy."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'xcorr_amd', 'lib')
    exe = str(tmp_path / 'abi_demo')
    subprocess.run([gcc, '-std=c99', '-O2', '-I', os.path.join(root, 'include'), os.path.join(root, 'examples', 'abi_demo.c'),
                    '-L', lib, '-lxcx', f'-Wl,-rpath,{lib}', '-lm', '-o', exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.splitlines()
    kat = np.array([float(v) for v in out[0].split()[1:]])
    assert np.allclose(kat, [0, .5, 2, 3.5, 5, 4, 0], atol=1e-5)
    f = out[1].split()
    assert int(f[2]) == 359999 and int(f[6]) == -250 and 0.8 < float(f[8]) < 1.0


def test_bench_gpus_2_starts_its_own_ranks(ctx):
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (here both
    on the one visible GPU: --oversubscribe) and prints ONE line with n_gpus = 2; without --oversubscribe it refuses
    when fewer than two devices are visible, and --stack (an RCCL communicator of two ranks on ONE device is refused
    by RCCL) fails loudly instead of hanging: the launcher stops the surviving rank."""
    import ctypes
    import json
    import subprocess
    import sys
    from xcorr_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    small = ['--stations', '6', '--seconds', '14400', '--steps', '1', '--warmup', '1', '--no-cpu-baseline']
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--oversubscribe'] + small,
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['value'] > 0
    ndev = ctypes.c_int(0)
    _ffi.check(_ffi.lib().xcx_device_count(ctypes.byref(ndev)))
    if ndev.value < 2:
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'] + small,
                             capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert out.returncode != 0 and 'n_gpus' not in out.stdout and '--oversubscribe' in out.stderr
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--oversubscribe', '--stack',
                              '--days', '2'] + small, capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert out.returncode != 0 and 'n_gpus' not in out.stdout
        assert 'stopping the other ranks' in out.stderr or 'exited with status' in out.stderr


def test_context_trim_reclaim_and_explicit_context(ctx):
    """(1) freed device buffers wait in the binding's pool and go back to the driver when the LIBRARY asks
    (xcx_set_reclaim callback) or Context.trim() is called; xcx_trim drops the context's cached workspaces and plans,
    later calls rebuild them and give the same numbers; (2) an op that is given a context stages large host rows in
    THAT context's pinned pool and never opens the process-wide default context."""
    import subprocess
    import sys
    from xcorr_amd import ops
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 400001)).astype(np.float32)
    y1 = np.array(ops.rfft(ctx, x))
    ctx.trim()
    y2 = np.array(ops.rfft(ctx, x))
    assert np.array_equal(y1, y2)
    a = ctx.empty((3 << 18,))                     # 3 MB
    b = ctx.empty((3 << 18,))
    a.free(); b.free()
    assert ctx._dev_pool_bytes == 2 * (3 << 20)
    c = ctx.empty((3 << 18,))                     # same size: comes from the pool
    assert ctx._dev_pool_bytes == 3 << 20
    assert ctx._reclaim_cb(None) == 3 << 20 and ctx._dev_pool_bytes == 0        # what the library calls when it runs out
    c.free()
    ctx.trim()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import numpy as np, xcorr_amd as X\n'
            'from xcorr_amd import engine, ops\n'
            'c = X.Context(0)\n'
            'x = np.random.default_rng(0).standard_normal((4, 300000)).astype(np.float32)\n'      # 4.8 MB: pinned staging
            'y = ops.demean(c, x); ops.taper(c, x, 100); ops.norm1d(c, x)\n'
            'assert abs(float(np.asarray(y).mean())) < 1e-6\n'
            'assert engine._default == {}, "an op with an explicit context opened the default one"\n'
            'assert c._pin_live + c._pin_idle > 0\n'
            'c.close()\n')
    subprocess.run([sys.executable, '-c', code], check=True, cwd=root, timeout=300)
