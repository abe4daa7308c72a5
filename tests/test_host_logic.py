"""CPU: host-side mirrors of the reference interface (no GPU work)."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R
import xcorr_amd.util.cc as ucc
import xcorr_amd.core as core
from xcorr_amd import shard
import xcorr_amd.signal as sig


def test_lag_weight_match_reference(golden):
    for n in golden['sizes']:
        n = int(n)
        np.testing.assert_allclose(ucc.lag(n, 0.02), golden[f'lag_{n}'], atol=1e-12)
        if n >= 3:
            # reference weight() is defined on the first 2n-3 lags only
            np.testing.assert_allclose(ucc.weight(n)[:2 * n - 3], golden[f'weight_{n}'], rtol=1e-15)
            np.testing.assert_allclose(ucc.weight(n, clip=3.0)[:2 * n - 3],
                                       golden[f'weight_{n}_clip'], rtol=1e-15)
    np.testing.assert_allclose(ucc.lag(4, .02, pad=False), golden['kat_lag_nopad'])


def test_extract_shift(golden):
    r = np.zeros(11)
    r[2] = 1.0
    assert ucc.extract_shift(r) == -3
    assert ucc.extract_shift(r, 0.5) == -1.5
    assert ucc.extract_shift_and_max(r) == (-3, 1.0)


def test_cc_argument_errors_match_reference():
    with pytest.raises(AssertionError, match='same length'):
        ucc.cc(np.zeros(4, np.float32), np.zeros(5, np.float32))
    with pytest.raises(AssertionError, match='same dtype'):
        ucc.cc(np.zeros(4, np.float32), np.zeros(4, np.float64))


def test_init_layout_matches_reference_rules():
    ds = core.init('A-B', 0., 86400., 50., window_length=3600., window_overlap=.5)
    assert ds.cc.shape == (1, 48, 359999) and ds.cc.dtype == np.float32
    assert ds.lag_attrs['npts'] == 180000
    assert (ds.lag_attrs['index_min'], ds.lag_attrs['index_max']) == (0, 359999)
    np.testing.assert_allclose(ds.time[:3], [0., 1800., 3600.])
    assert np.all(ds.status == 0)
    ds = core.init('A-B', 0., 7200., 50., 3600., .5, clip_lag=(-10., 10.))
    assert (ds.lag_attrs['index_min'], ds.lag_attrs['index_max']) == R.clip_lag_indices(180000, .02, (-10., 10.))
    assert ds.lag.shape == (1000,) and ds.lag[0] == -10.0
    # update_lag_indices (util/time.py:104-108) is consistent with the slice
    assert R.lag_indices(180000, .02, ds.lag[0], ds.lag[-1]) == (179499, 180499)
    ds = core.init('A-B', 0., 7200., 50., 3600., .5, clip_lag=5.)
    assert ds.lag[0] == -5.0
    with pytest.raises(TypeError):
        core.init('A-B', 0., 7200., 50., 3600., .5, clip_lag=[1, 2])


def test_array_client_windows():
    fs = 50.
    tr = {'A': np.arange(1000, dtype=np.float32), 'B': np.arange(1000, dtype=np.float32) * 2}
    cl = core.ArrayClient(tr, fs, t0=100.)
    w = cl.get_pair_processed_waveforms('A-B', time=105., duration=4.)
    (xa, ta), (xb, tb) = w
    assert len(xa) == 200 and xa[0] == 150 and ta == 103. and xb[0] == 300
    assert cl.get_pair_processed_waveforms('A-B', time=100., duration=4.) is None
    assert cl.get_pair_processed_waveforms('A-C', time=105., duration=4.) is None


def test_shard_helpers():
    pairs = shard.all_pairs(64)
    assert pairs.shape == (2016, 2) and tuple(pairs[0]) == (0, 1) and tuple(pairs[-1]) == (62, 63)
    seen = np.zeros(2016, int)
    sizes = []
    for r in range(8):
        p, idx = shard.shard_pairs(pairs, r, 8)
        seen[idx] += 1
        sizes.append(len(p))
        np.testing.assert_array_equal(p, pairs[idx])
    assert np.all(seen == 1) and max(sizes) - min(sizes) <= 1
    days = [shard.shard_days(30, r, 8) for r in range(8)]
    assert [len(d) for d in days] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert sum(days, []) == list(range(30))
    assert len(shard.window_starts(4320000, 180000, 90000)) == 47


def test_signal_host_helpers():
    np.testing.assert_allclose(sig.window(1000, 50., max_length=5.), R.window(1000, 50., max_length=5.))
    np.testing.assert_allclose(sig.window(999, 50., max_percentage=.1), R.window(999, 50., max_percentage=.1))
    np.testing.assert_allclose(sig.unbias_weights(100, 10, 150), R.weight_full(100)[10:150])
    with pytest.raises(TypeError):
        sig.window(100, 50., max_length=5)
    with pytest.raises(ValueError):
        sig.filter(np.zeros((2, 100), np.float32), 3, sampling_rate=50.)
    with pytest.raises(KeyError):
        sig.detrend(np.zeros((2, 10), np.float32), type='cubic')


def test_synthetic_generator_matches_oracle_copy():
    from xcorr_amd.synth import synthetic_traces
    a = synthetic_traces(3, 500, day=2)
    b = R.synthetic_traces(3, 500, day=2)
    np.testing.assert_array_equal(a, b)
    assert a.dtype == np.float32 and a.shape == (3, 500)


def test_stream_filter_designs_and_lengths():
    """Host-side designs of the ObsPy-style ops (no GPU): Butterworth via iirfilter/zpk2sos,
    lowpass_cheby_2 order search, cosine taper flank length."""
    import scipy.signal
    from xcorr_amd import stream
    from oracle import ref_numpy as R
    sos = stream.butter_sos('bandpass', 50., 4, freqmin=0.1, freqmax=10.)
    assert sos.shape == (4, 6)
    assert np.allclose(sos, scipy.signal.butter(4, [0.1, 10.], 'bandpass', output='sos', fs=50.))
    assert np.allclose(stream.butter_sos('highpass', 250., 4, freq=.05),
                       scipy.signal.butter(4, .05, 'highpass', output='sos', fs=250.))
    with pytest.raises(NotImplementedError):
        stream.butter_sos('bandstop', 50., 4, freqmin=1., freqmax=2.)
    for factor in (2, 5, 10, 16):
        s = stream.cheby2_lowpass_sos(0.5 * 250. / factor, 250.)
        assert s.shape[0] <= 6 and np.allclose(s, R.cheby2_lowpass_sos(0.5 * 250. / factor, 250.))
        w, h = scipy.signal.sosfreqz(s, worN=4096, fs=250.)
        assert np.all(20 * np.log10(np.abs(h[w >= 0.5 * 250. / factor]) + 1e-300) <= -95.9)   # 96 dB stop band
        # independent of the (duplicated) design routine: a low-pass of order <= 12 that is flat to
        # 1 dB over the lower 60 % of the new band and monotone into the stop band
        db = 20 * np.log10(np.abs(h) + 1e-300)
        assert np.all(np.abs(db[w <= 0.3 * 250. / factor]) < 1.0) and abs(db[0]) < 1e-6
        assert 2 * s.shape[0] <= 12
    assert stream.cosine_taper_length(4320000, 50., 0.05, 30.) == 1500
    assert stream.cosine_taper_length(1000, 50., 0.05, 30.) == 50
    assert stream.cosine_taper_length(10, 50., None, None) == 5


def test_fft_axes_and_spectrogram_design():
    """Coordinate rules of signal/fft.py and the scipy-equivalent spectrogram design."""
    import scipy.signal
    from xcorr_amd.signal import fftops, spectral
    from oracle import ref_numpy as R
    t = np.arange(6001) * 0.02 - 60.
    assert np.array_equal(fftops.rfft_freq(t), R.rfft_freq(t)) and fftops.rfft_freq(t).size == 3001
    assert np.allclose(fftops.fft_freq(9, 0.02), np.fft.fftshift(np.fft.fftfreq(9, 0.02)))
    f = fftops.rfft_freq(t)
    assert np.allclose(fftops.irfft_time(f, -60.)[:3], [-60., -59.98, -59.96])
    win_len, nfft, win, scale, scaling, *_ = spectral._design(50., 2.5, 4, None, ('tukey', .25))
    assert (win_len, nfft, scaling) == (125, 500, 'density')
    assert np.allclose(win, scipy.signal.get_window(('tukey', .25), 125))
    assert scale == pytest.approx(1.0 / (50. * (win ** 2).sum()))
    with pytest.raises(ValueError):
        spectral._design(50., 0.2, 2, None, 'hann')
    with pytest.raises(ValueError):
        spectral._design(50., 2.5, 2, 'power', 'hann')


def test_output_buffer_validation_is_host_side():
    """`out` capacity / dtype / contiguity are checked before any library call."""
    from xcorr_amd.engine import _check_out
    L, ld = 1999, 2016
    _check_out(np.empty((5, ld), np.float32), 5, ld, L)
    _check_out(np.empty(4 * ld + L, np.float32), 5, ld, L)          # the last row needs only L
    with pytest.raises(ValueError, match='holds'):
        _check_out(np.empty(4 * ld + L - 1, np.float32), 5, ld, L)
    with pytest.raises(ValueError, match='ld_out'):
        _check_out(np.empty((5, L), np.float32), 5, L - 1, L)
    with pytest.raises(ValueError, match='float32'):
        _check_out(np.empty((5, ld), np.float64), 5, ld, L)
    with pytest.raises(ValueError, match='contiguous'):
        _check_out(np.empty((5, 2 * ld), np.float32)[:, ::2], 5, ld, L)
    ro = np.empty((5, ld), np.float32)
    ro.flags.writeable = False
    with pytest.raises(ValueError, match='read-only'):
        _check_out(ro, 5, ld, L)


def test_init_reference_signature():
    """core.init with the reference's argument order (core/init.py:25-40): timestamps, preprocess,
    attrs checks, and the production lag slice of scripts/cc.py:25-29."""
    pre = {'HHZ': [('detrend', {'type': 'demean'}), ('filter', {'type': 'highpass', 'freq': .05})]}
    att = {'institution': 'i', 'author': 'a', 'source': 's', 'title': 'T'}
    ds = core.init('NL.A..HHZ-NL.B..HHZ', '2020-01-01', '2020-01-02', pre, att, 50.,
                   window_length=86400., window_overlap=.875, clip_lag=(0., 9 * 3600.))
    assert len(ds.time) == 8 and ds.time_is_datetime and ds.time[1] - ds.time[0] == 10800.
    la = ds.lag_attrs
    assert (la['npts'], la['index_min'], la['index_max']) == (4320000, 4319999, 5939999)
    assert ds.lag[0] == 0.0 and abs(ds.lag[-1] - (9 * 3600. - 0.02)) < 1e-6 and ds.cc.shape == (1, 8, 1620000)
    assert ds.attrs['title'] == 'T Cross-correlations' and ds.pair_attrs['preprocess'] is pre
    with pytest.raises(TypeError):
        core.init('A-B', 0., 1., pre, 'not a dict', 50.)
    for missing in ('institution', 'author', 'source'):
        with pytest.raises(KeyError):
            core.init('A-B', 0., 1., pre, {k: v for k, v in att.items() if k != missing}, 50.)
    with pytest.raises(TypeError):
        core.init(17, 0., 1., pre, att, 50.)
    with pytest.raises(TypeError):
        core.init('A-B', 0., 1., pre, att, 50., clip_lag=(0, 10.))          # int start (core/init.py:183-187)
    # the build's earlier numeric form keeps working
    old = core.init('A-B', 0., 86400., 50., 3600., .5)
    assert len(old.time) == 48 and old.time_attrs['window_overlap'] == .5 and not old.time_is_datetime


def test_io_roundtrip_naming_and_skip_if_done(tmp_path):
    """f4 plumbing (io/write.py:31-184, io/filesystem.py:88-95, core/lazy.py:52-58): file scheme,
    write-through-temporary, empty datasets are not written, hash-checked read, and the resume logic
    returns early (no GPU) when every step of the file is processed."""
    import warnings
    from xcorr_amd import io
    from xcorr_amd.core import lazy
    pair = 'NL.A..HHZ-NL.B..HHZ'
    att = {'institution': 'i', 'author': 'a', 'source': 's'}
    args = dict(preprocess={'HHZ': [('detrend', {'type': 'demean'})]}, attrs=att, sampling_rate=50.,
                window_length=600., window_overlap=.5, clip_lag=(-10., 10.))
    ds = core.init(pair, '2020-02-03', '2020-02-04', **args)
    path = io.ncfile(pair, ds.time[0], str(tmp_path))
    assert path.endswith(os.path.join('2020', pair, pair + '.2020.034.nc'))
    assert io.ncfile(pair, np.datetime64('2021-12-31T10:00'), None) == os.path.join('2021', pair, pair + '.2021.365.nc')
    with pytest.raises(TypeError):
        io.ncfile(17, 0.)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert io.write(ds, path) is False and not os.path.exists(path) and 'no data' in str(w[0].message)
    assert io.read(path) is None
    ds.cc[...] = np.random.default_rng(0).standard_normal(ds.cc.shape).astype(np.float32)
    ds.status[...] = 1
    ds.status[0, 3] = -1
    ds.pair_offset[0, 2] = 0.004
    assert io.write(ds, path) is True and not [f for f in os.listdir(os.path.dirname(path)) if f.endswith('.tmp')]
    r = io.read(path)
    assert np.array_equal(r.cc, ds.cc) and np.array_equal(r.status, ds.status) and np.array_equal(r.pair_offset, ds.pair_offset)
    assert np.array_equal(r.time, ds.time) and np.allclose(r.lag, ds.lag) and list(r.pair) == [pair] and r.time_is_datetime
    assert r.lag_attrs['index_min'] == ds.lag_attrs['index_min'] and r.lag_attrs['npts'] == 30000
    assert r.cc_attrs['normalize'] == 1 and r.time_attrs['window_overlap'] == .5 and r.attrs['institution'] == 'i'
    assert io.validate(r)
    # corrupt one data byte: the stored sha256 no longer matches
    raw = bytearray(open(path, 'rb').read())
    raw[len(raw) // 2] ^= 0xff                          # the middle of the file lies inside cc
    open(path, 'wb').write(raw)
    with pytest.raises(ValueError, match='sha256'):
        io.read(path)
    assert io.read(path, quick_and_dirty=True) is not None
    # resume: a fully processed day is skipped without initialising anything
    ds.status[...] = 1
    io.write(ds, path)

    class NoClient:
        def get_pair_processed_waveforms(self, **kw):
            raise AssertionError('a finished file must not be reprocessed')
    assert lazy.single_process(pair, '2020-02-03', args, NoClient(), str(tmp_path)) is True


# ---- stream segment bookkeeping (stream/process.py:50-53,86-90; ObsPy rules restated) -------------
def test_merge_rules():
    from xcorr_amd.stream import merge
    fs = 10.
    a = (np.arange(10, dtype=np.float32), 0.0)                 # t = 0.0 .. 0.9
    b = (np.arange(100, 110, dtype=np.float32), 1.0)           # exact fit
    y, t0 = merge([b, a], fs)
    assert t0 == 0.0 and np.array_equal(y, np.r_[a[0], b[0]])
    # gap of 5 samples: NaN / number / latest / interpolate
    c = (np.full(4, 20, np.float32), 1.5)
    y, _ = merge([a, c], fs)
    assert y.size == 19 and np.isnan(y[10:15]).all() and y[15] == 20
    assert np.array_equal(merge([a, c], fs, fill_value=0)[0][10:15], np.zeros(5))
    assert np.array_equal(merge([a, c], fs, fill_value='latest')[0][10:15], np.full(5, 9))
    line = np.linspace(9, 20, 7)[1:-1].astype(np.float32)
    assert np.allclose(merge([a, c], fs, method=1, fill_value='interpolate', interpolation_samples=0)[0][10:15], line)
    # overlap of 3 samples, method 1: the later segment wins from the overlap on
    d = (np.arange(50, 56, dtype=np.float32), 0.7)
    y, _ = merge([a, d], fs, method=1)
    assert np.array_equal(y, np.r_[np.arange(7), np.arange(50, 56)])
    # ... interpolation_samples=-1: straight line across the whole overlap
    y, _ = merge([a, d], fs, method=1, interpolation_samples=-1)
    assert np.allclose(y[7:10], np.linspace(6, 53, 5)[1:-1]) and np.array_equal(y[10:], [53, 54, 55])
    # method 0: differing overlap samples are discarded, equal ones kept
    e = (np.r_[7, 8, 99, 10, 11].astype(np.float32), 0.7)
    y, _ = merge([a, e], fs, method=0)
    assert y[7] == 7 and y[8] == 8 and np.isnan(y[9]) and y[10] == 10
    # contained segment: dropped by method 1
    f = (np.full(3, -1, np.float32), 0.3)
    assert np.array_equal(merge([a, f], fs, method=1)[0], a[0])
    with pytest.raises(ValueError):
        merge([], fs)


def test_trim_rules():
    from xcorr_amd.stream import trim
    fs = 10.
    x = np.arange(20, dtype=np.float32)                        # t = 100.0 .. 101.9
    y, t0 = trim(x, fs, 100.0, 100.5, 101.2)
    assert np.array_equal(y, np.arange(5, 13)) and abs(t0 - 100.5) < 1e-12
    # nearest sample vs strictly inside
    y, t0 = trim(x, fs, 100.0, 100.46, 101.24)
    assert y[0] == 5 and y[-1] == 12
    y, t0 = trim(x, fs, 100.0, 100.46, 101.24, nearest_sample=False)
    assert y[0] == 5 and y[-1] == 12 and abs(t0 - 100.5) < 1e-12
    y, _ = trim(x, fs, 100.0, 100.44, 101.26, nearest_sample=False)
    assert y[0] == 5 and y[-1] == 12
    # bounds outside the data: unchanged without pad, extended with pad
    y, t0 = trim(x, fs, 100.0, 99.0, 103.0)
    assert np.array_equal(y, x) and t0 == 100.0
    y, t0 = trim(x, fs, 100.0, 99.8, 102.1, pad=True, fill_value=0)
    assert y.size == 24 and np.array_equal(y[:2], [0, 0]) and np.array_equal(y[-2:], [0, 0]) and abs(t0 - 99.8) < 1e-12
    assert np.isnan(trim(x, fs, 100.0, 99.8, None, pad=True)[0][:2]).all()
    # leading dimensions are kept; empty result when the window misses the data
    y, _ = trim(np.stack([x, x]), fs, 100.0, 100.5, 101.2)
    assert y.shape == (2, 8)
    assert trim(x, fs, 100.0, 105.0, 106.0)[0].size == 0
    with pytest.raises(ValueError):
        trim(x, fs, 100.0, 101.0, 100.0)


def test_rotate_and_sensitivity():
    from xcorr_amd.stream import rotate_ne_rt, remove_sensitivity
    n, e = np.array([1., 0.]), np.array([0., 1.])
    # the arithmetic of the rotation runs in the library (tests/test_gpu_stream.py); argument checks come first
    with pytest.raises(ValueError):
        rotate_ne_rt(n, e, 400.)
    with pytest.raises(TypeError):
        rotate_ne_rt(n, e[:1], 10.)
    x = np.array([2., 4.], np.float32)
    y = remove_sensitivity(x, 2.0)
    assert y.dtype == np.float32 and np.array_equal(y, [1, 2])
    from xcorr_amd.stream import normalize
    z = normalize(np.array([[1., -4., 2.], [0., 0., 0.]], np.float32))
    assert z.dtype == np.float32 and np.array_equal(z, [[.25, -1., .5], [0., 0., 0.]])
    assert np.array_equal(normalize(np.array([2., 4.]), norm=-2.), [1., 2.])


def test_rotate_and_select_act_on_component_sets():
    """stream/process.py:70-77 `rotate` / `select` through the registry: bookkeeping and refusals (no GPU)."""
    from xcorr_amd import stream
    from xcorr_amd.stream import ops as sops
    z, n, e = (np.zeros(4, np.float32) + i for i in range(3))
    comps = {'BHZ': z, 'BHN': n, 'BHE': e}
    assert sops.rotate(comps, '->ZNE') is comps                     # already ZNE: untouched
    with pytest.raises(NotImplementedError, match='inventory'):
        sops.rotate({'BHZ': z, 'BH1': n, 'BH2': e}, '->ZNE')
    with pytest.raises(ValueError, match='back_azimuth'):
        sops.rotate(comps, 'NE->RT')
    with pytest.raises(NotImplementedError):
        sops.rotate(comps, 'ZNE->LQT', back_azimuth=10., inclination=5.)
    with pytest.raises(ValueError, match='no pair'):
        sops.rotate({'BHZ': z, 'BHN': n}, 'NE->RT', back_azimuth=10.)
    assert list(sops.select(comps, channel='BH[NE]')) == ['BHN', 'BHE']
    assert list(sops.select(comps, component='Z')) == ['BHZ']
    with pytest.raises(ValueError, match='matches none'):
        sops.select(comps, channel='BHR')
    with pytest.raises(NotImplementedError):
        sops.select(comps, station='ANMO')
    # single traces: set operations and inventory operations are refused by name
    with pytest.raises(TypeError, match='component set'):
        stream.apply_operations(z, 50., [('select', {'channel': 'BHZ'})])
    for name in ('remove_response', 'attach_response', 'simulate'):
        with pytest.raises(NotImplementedError, match='inventory'):
            stream.apply_operations(z, 50., [(name, {})])
    # a set that only goes through set operations needs no GPU: select down to one component -> that array
    y, fs = stream.apply_operations(comps, 50., [('rotate', {'method': '->ZNE'}), ('select', {'channel': 'BHZ'})])
    assert y is z and fs == 50.


def test_hasher_restates_reference_serialisation():
    """util/hasher.py:38-48,106-139,174-181: json form and update order; sha256 recomputed by hand."""
    import hashlib, json
    from xcorr_amd.util import hasher
    obj = {'b': np.int32(2), 'a': np.float64(.5), 'sha256_hash': 'ignored', 'scale_factor': 1., 'c': [1, 2]}
    text = json.dumps({'a': .5, 'b': 2, 'c': [1, 2]}, separators=(',', ':'), sort_keys=True, indent=4)
    assert hasher.hash_obj(obj) == hashlib.sha256(text.encode()).hexdigest()
    assert hasher.utcdatetime_repr(1421280000.0) == 'UTCDateTime(2015, 1, 15, 0, 0)'
    assert hasher.utcdatetime_repr(1421280005.5) == 'UTCDateTime(2015, 1, 15, 0, 0, 5, 500000)'
    st = hasher.trace_stats('IM.H10N1..EDH', 1421280000.0, 50., 4)
    assert st['endtime'] == 1421280000.06 and st['location'] == '' and st['delta'] == .02
    data = np.arange(4, dtype=np.float32)
    head = dict(st, starttime='UTCDateTime(2015, 1, 15, 0, 0)', endtime='UTCDateTime(2015, 1, 15, 0, 0, 0, 60000)')
    h = hashlib.sha256()
    h.update(json.dumps(head, separators=(',', ':'), sort_keys=True, indent=4).encode())
    for d in data:
        h.update(d.tobytes())
    assert hasher.hash_trace(st, data) == h.hexdigest()
    # stream = one running hash, order matters
    two = hasher.hash_stream([(st, data), (st, data[::-1].copy())])
    assert two != hasher.hash_stream([(st, data[::-1].copy()), (st, data)])
    assert two != hasher.hash_trace(st, data)


def test_io_keeps_hash_variable(tmp_path):
    from xcorr_amd import io
    ds = core.init('A.B..C-D.E..F', 0., 3600., sampling_rate=10., window_length=600., window_overlap=.5,
                   hash_waveforms=True)
    ds.status[...] = 1
    ds.hash[0, 2] = 'ab' * 32
    path = str(tmp_path / 'h.nc')
    assert io.write(ds, path)
    r = io.read(path)
    assert r.hash.shape == ds.hash.shape and r.hash[0, 2] == 'ab' * 32 and r.hash[0, 0] == 'n/a'


def test_io_drops_digests_of_other_schemes(tmp_path):
    """A container that carries `sha256_hash` (earlier builds of this writer, or the reference's attribute) is
    written without it and read without it: the only digest in a file is the one computed over its own data."""
    from scipy.io import netcdf_file
    from xcorr_amd import io
    ds = core.init('A.B..C-D.E..F', 0., 3600., sampling_rate=10., window_length=600., window_overlap=.5)
    ds.status[...] = 1
    ds.attrs['sha256_hash'] = 'f' * 64
    ds.attrs['sha256_hash_metadata'] = 'e' * 64
    path = str(tmp_path / 's.nc')
    assert io.write(ds, path)
    with netcdf_file(path, 'r', mmap=False) as nc:
        assert 'sha256_hash' not in nc._attributes and 'sha256_hash_metadata' not in nc._attributes
        assert io.nc.HASH_ATTR in nc._attributes
    # a file of an earlier build: the old attribute is present in the file itself
    with netcdf_file(path, 'a', mmap=False) as nc:
        nc.sha256_hash = b'd' * 64
    r = io.read(path)
    assert 'sha256_hash' not in r.attrs and io.nc.HASH_ATTR not in r.attrs
    m = io.merge(r, r)
    assert 'sha256_hash' not in m.attrs


def test_init_hash_variable():
    from xcorr_amd.core import init
    ds = init('IM.H10N1..EDH-IU.RAR.10.BHZ', 0., 7200., sampling_rate=50., window_length=3600., window_overlap=.5,
              hash_waveforms=True)
    assert ds.hash.shape == ds.status.shape and (ds.hash == 'n/a').all()
    assert init('A-B', 0., 7200., sampling_rate=50., window_length=3600., window_overlap=.5).hash is None


def test_io_merge_and_mfread(tmp_path):
    """core/merge.py:53-79, io/read.py:95-200: union of pairs and times, processed steps win."""
    from xcorr_amd import io
    rng = np.random.default_rng(4)
    att = {'institution': 'i', 'author': 'a', 'source': 's', 'title': 'T'}
    kw = dict(attrs=att, sampling_rate=10., window_length=600., window_overlap=.5, clip_lag=(-5., 5.))
    days = []
    for day, pair in (('2020-02-03', 'A.B..Z-C.D..Z'), ('2020-02-04', 'A.B..Z-C.D..Z'), ('2020-02-03', 'A.B..Z-E.F..Z')):
        ds = core.init(pair, day, str(np.datetime64(day) + 1), **kw)
        ds.cc[...] = rng.standard_normal(ds.cc.shape).astype(np.float32)
        ds.status[...] = 1
        ds.status[0, 5] = 0
        ds.cc[0, 5] = 0
        path = io.ncfile(pair, ds.time[0], str(tmp_path))
        assert io.write(ds, path)
        days.append((ds, path))
    m = io.merge(days[0][0], days[1][0], days[2][0])
    n0 = len(days[0][0].time)
    assert list(m.pair) == ['A.B..Z-C.D..Z', 'A.B..Z-E.F..Z'] and len(m.time) == 2 * n0
    assert np.array_equal(m.cc[0, :n0], days[0][0].cc[0]) and np.array_equal(m.cc[0, n0:], days[1][0].cc[0])
    assert np.array_equal(m.cc[1, :n0], days[2][0].cc[0]) and (m.status[1, n0:] == 0).all() and (m.cc[1, n0:] == 0).all()
    assert m.attrs['title'].startswith('T Cross-correlations - 2020.034 to 2020.035') and m.attrs['history'].startswith('Merged @')
    # a processed step is not overwritten by a pending copy of the same step, in either order
    redo = core.init('A.B..Z-C.D..Z', '2020-02-03', '2020-02-04', **kw)          # all pending
    assert np.array_equal(io.merge(days[0][0], redo).cc, days[0][0].cc)
    assert np.array_equal(io.merge(redo, days[0][0]).status, days[0][0].status)
    # files: glob, list, extract
    g = io.mfread(os.path.join(str(tmp_path), '2020', '*', '*.nc'), extract=True)
    assert g.cc.shape == m.cc.shape and np.isnan(g.cc[0, 5]).all() and np.isnan(g.cc[1, n0:]).all()
    assert np.array_equal(np.nan_to_num(g.cc), m.cc)
    assert io.merge(os.path.join(str(tmp_path), 'nothing', '*.nc')) is None
    assert np.isnan(io.read(days[0][1], extract=True).cc[0, 5]).all()
    other = core.init('A-B', 0., 7200., sampling_rate=10., window_length=600., window_overlap=.5)
    with pytest.raises(ValueError):
        io.merge(days[0][0], other)


def test_top_level_names_of_the_reference():
    """xcorr/__init__.py:34-46: init / process / postprocess / read / write / merge / mfread, signal, stream, util, io."""
    import xcorr_amd as X
    for name in ('init', 'process', 'postprocess', 'read', 'write', 'merge', 'mfread'):
        assert callable(getattr(X, name)), name
    assert X.init is core.init and X.read is X.io.read
    assert X.signal.__name__ == 'xcorr_amd.signal' and X.stream.__name__ == 'xcorr_amd.stream'
    with pytest.raises(AttributeError):
        X.plot


def test_container_methods_mirror_the_dataset_accessor(tmp_path):
    """core/accessors.py:52-110: postprocess / process / write / merge as methods."""
    ds = core.init('A.B..Z-C.D..Z', 0., 7200., sampling_rate=10., window_length=600., window_overlap=.5)
    ds.status[...] = 1
    path = str(tmp_path / 'x.nc')
    assert ds.write(path) is True
    m = ds.merge(path)
    assert np.array_equal(m.status, ds.status) and m.attrs['history'].startswith('Merged @')
    assert callable(ds.process) and callable(ds.postprocess)


class _FakeVar:
    """Just enough of an xarray.DataArray for CCDataset.from_xarray / update_xarray (values + attrs)."""

    def __init__(self, values, attrs=None):
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})


class _FakeXrDataset(dict):
    attrs = {}


def test_ccdataset_xarray_conversion_on_a_duck_typed_dataset():
    """``from_xarray`` reads exactly the names xcorr/core/init.py:153-320 lays out and ``update_xarray`` writes
    the processed variables back in place.  xarray itself is not installable here: this exercises
    the conversion logic on an object with the same attribute surface, not xarray."""
    import importlib
    cinit = importlib.import_module('xcorr_amd.core.init')
    ds0 = cinit.init('AA.S1..HHZ-AA.S2..HHZ', 0., 7200., sampling_rate=50., window_length=3600., window_overlap=0.5,
                     clip_lag=(-10., 10.), hash_waveforms=True)
    x = _FakeXrDataset()
    x.attrs = dict(ds0.attrs)
    x['pair'] = _FakeVar(ds0.pair, {'units': '-'})
    x['time'] = _FakeVar((ds0.time * 1e9).astype('datetime64[ns]'), ds0.time_attrs)
    x['lag'] = _FakeVar(ds0.lag, {**ds0.lag_attrs, 'npts': np.int64(ds0.lag_attrs['npts'])})
    x['cc'] = _FakeVar(ds0.cc.copy(), ds0.cc_attrs)
    for k in ('status', 'pair_offset', 'time_offset', 'hash'):
        x[k] = _FakeVar(getattr(ds0, k).copy())
    back = cinit.CCDataset.from_xarray(x)
    assert back.time_is_datetime and np.allclose(back.time, ds0.time)
    assert list(back.pair) == list(ds0.pair) and back.lag_attrs['npts'] == ds0.lag_attrs['npts']
    assert isinstance(back.lag_attrs['npts'], int) and back.cc.shape == ds0.cc.shape and back.hash is not None
    back.cc[...] = 1.5
    back.status[...] = 1
    back.hash[...] = 'h'
    back.update_xarray(x)
    assert (x['cc'].values == 1.5).all() and (x['status'].values == 1).all() and (x['hash'].values == 'h').all()
    assert not cinit.is_xarray_dataset(x)            # the real type check needs the real xarray


def _pair_groups(pairs, S, batch):
    import ctypes as C
    from xcorr_amd import _ffi
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    P = len(pairs)
    perm = np.empty(P, np.int32)
    bounds = np.empty(P + 1, np.int32)
    ng = C.c_int(0)
    _ffi.check(_ffi.lib().xcx_pair_groups(_ffi.ptr(pairs), S, P, batch, _ffi.ptr(perm), _ffi.ptr(bounds), C.byref(ng)))
    return perm, bounds[:ng.value + 1]


def test_pair_groups_processing_order():
    """csrc/pair_order.h (host logic, no GPU): every pair exactly once, groups within the batch size, and for the
    all-pairs list of 64 stations the 72 lines of AG(2, 8): 28 pairs on 8 stations each."""
    for S, batch in ((64, 28), (64, 25), (16, 28), (30, 25), (100, 28), (3, 28), (2, 4)):
        pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
        perm, bounds = _pair_groups(pairs, S, batch)
        assert sorted(perm.tolist()) == list(range(len(pairs)))
        sizes = np.diff(bounds)
        assert bounds[0] == 0 and bounds[-1] == len(pairs) and sizes.min() >= 1 and sizes.max() <= batch
        stations = [len(set(pairs[perm[a:b]].ravel().tolist())) for a, b in zip(bounds[:-1], bounds[1:])]
        if (S, batch) == (64, 28):
            assert len(sizes) == 72 and set(sizes.tolist()) == {28} and set(stations) == {8}
        # never worse than one new station pair per pair
        assert sum(stations) <= 2 * len(pairs)
        if S >= 16:
            assert sum(stations) / len(pairs) < 0.45          # 8 x 8 tiles in 25-pair cuts: 0.48
    # arbitrary lists: reversed pairs, auto-pairs, duplicates, a pair listed more often than the batch size
    rng = np.random.default_rng(3)
    pairs = np.concatenate([rng.integers(0, 11, size=(60, 2)), [[3, 3], [10, 0], [10, 0]], [[5, 6]] * 9]).astype(np.int32)
    perm, bounds = _pair_groups(pairs, 11, 4)
    assert sorted(perm.tolist()) == list(range(len(pairs))) and np.diff(bounds).max() <= 4 and np.diff(bounds).min() >= 1


def test_fftlib_host_rules(monkeypatch):
    """xcorr_amd.fftlib (B3 seam): numpy's n= / s= / axis= / norm= rules are host logic around the device
    transforms.  With the device calls replaced by numpy's own (float32 in, same signatures), every wrapper
    must reproduce numpy.fft for cropped, padded, odd and multi-axis calls."""
    from xcorr_amd import fftlib, ops
    monkeypatch.setattr(ops, 'fft', lambda ctx, x: np.fft.fft(x, axis=-1).astype(np.complex64))
    monkeypatch.setattr(ops, 'ifft', lambda ctx, x: np.fft.ifft(x, axis=-1).astype(np.complex64))
    monkeypatch.setattr(ops, 'rfft', lambda ctx, x: np.fft.rfft(x, axis=-1).astype(np.complex64))
    monkeypatch.setattr(ops, 'irfft', lambda ctx, x: np.fft.irfft(x, axis=-1).astype(np.float32))
    rng = np.random.default_rng(8)
    x = rng.standard_normal((5, 12, 9)).astype(np.float32)
    z = (x + 1j * rng.standard_normal(x.shape)).astype(np.complex64)
    close = lambda a, b: np.allclose(a, b, rtol=2e-5, atol=2e-5 * np.abs(b).max())
    for axis in (-1, 0, 1):
        for n in (None, 4, 9, 12, 20):
            for norm in (None, 'ortho', 'forward'):
                assert close(fftlib.fft(z, n=n, axis=axis, norm=norm), np.fft.fft(z, n=n, axis=axis, norm=norm))
                assert close(fftlib.ifft(z, n=n, axis=axis, norm=norm), np.fft.ifft(z, n=n, axis=axis, norm=norm))
                assert close(fftlib.rfft(x, n=n, axis=axis, norm=norm), np.fft.rfft(x, n=n, axis=axis, norm=norm))
                h = np.fft.rfft(x, axis=axis)
                assert close(fftlib.irfft(h, n=n, axis=axis, norm=norm), np.fft.irfft(h, n=n, axis=axis, norm=norm))
    for s in (None, (8, 16), (12, 9), (5, 5)):
        for axes in ((-2, -1), (0, 2), (2, 0)):
            assert close(fftlib.fft2(z, s=s, axes=axes), np.fft.fft2(z, s=s, axes=axes))
            assert close(fftlib.ifft2(z, s=s, axes=axes), np.fft.ifft2(z, s=s, axes=axes))
    with pytest.raises(ValueError):
        fftlib.fft(z, n=0)
    with pytest.raises(ValueError):
        fftlib.fft2(z, axes=(0, 1, 2))
    with pytest.raises(ValueError):
        fftlib.fft(z, norm='bogus')
    assert fftlib.fftshift is np.fft.fftshift and fftlib.rfftfreq is np.fft.rfftfreq


def test_filter_designs_are_cached_copies():
    """stream.butter_sos / cheby2_lowpass_sos and signal's butter_design keep their scipy designs per argument set (a
    Chebyshev anti-alias design costs more host time than decimating a station-day on the device): the cached design
    equals a fresh one bit for bit, and what a caller does to the array it got does not reach the next caller."""
    import scipy.signal
    from xcorr_amd import stream
    from xcorr_amd.signal.ops1d import butter_design
    a = stream.cheby2_lowpass_sos(25., 250.)
    assert np.array_equal(a, R.cheby2_lowpass_sos(25., 250.))
    a[:] = 0.0
    assert np.array_equal(stream.cheby2_lowpass_sos(25., 250.), R.cheby2_lowpass_sos(25., 250.))
    for ftype, kw in (('highpass', dict(freq=0.05)), ('lowpass', dict(freq=20.)), ('bandpass', dict(freqmin=0.1, freqmax=10.))):
        b = stream.butter_sos(ftype, 50., 4, **kw)
        fe = 25.
        wn = [kw['freqmin'] / fe, kw['freqmax'] / fe] if ftype == 'bandpass' else kw['freq'] / fe
        z, p, k = scipy.signal.iirfilter(4, wn, btype='band' if ftype == 'bandpass' else ftype[:-4], ftype='butter', output='zpk')
        assert np.array_equal(b, scipy.signal.zpk2sos(z, p, k))
        b[:] = 1.0
        assert not np.array_equal(stream.butter_sos(ftype, 50., 4, **kw), b)
    with pytest.raises(NotImplementedError):
        stream.butter_sos('bandstop', 50., 4, freqmin=1., freqmax=2.)
    for order, wn, bt in ((2, 3.0, 'highpass'), (4, (0.1, 10.), 'bandpass'), (4, [1.5, 10.], 'bandpass')):
        c = butter_design(order, wn, bt, 50.)
        ref = scipy.signal.butter(N=order, Wn=list(wn) if not isinstance(wn, float) else wn, btype=bt, output='sos', fs=50.)
        assert np.array_equal(c, ref)
        c[:] = 0.0
        assert np.array_equal(butter_design(order, wn, bt, 50.), ref)
