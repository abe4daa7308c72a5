"""GPU: hop-1 spectrogram (xcorr/signal/spectrogram.py:124-141) on fp32 MFMA against
scipy.signal.spectrogram in float64; tolerance 1e-5 of the largest PSD value."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
from xcorr_amd import ops, signal          # noqa: E402
from xcorr_amd._ffi import XcxError        # noqa: E402

TOL = 1e-5
FS = 50.


def ccf_like(rows, n, seed):
    rng = np.random.default_rng(seed)
    t = (np.arange(n) - n // 2) / FS
    x = 0.05 * rng.standard_normal((rows, n))
    x += np.exp(-0.5 * (t / 8.0) ** 2) * np.cos(2 * np.pi * 4.0 * t)          # wave packet around lag 0
    x += 0.3                                                                  # offset: exercises detrend
    return x.astype(np.float32)


def check(S, ref):
    assert S.shape == ref.shape
    assert np.abs(S - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize('duration,pf', [(2.5, 4), (2.5, 2), (0.5, 1), (1.0, 8), (3.0, 2), (4.5, 1), (5.12, 2)])
def test_matches_scipy(ctx, duration, pf):
    x = ccf_like(3, 1501, seed=int(duration * 10) + pf)
    freq, S = signal.spectrogram(x, duration=duration, padding_factor=pf, sampling_rate=FS)
    rfreq, ref = R.spectrogram(x, FS, duration, pf)
    assert np.allclose(freq, rfreq)
    check(S, ref)
    edge = int(np.rint(int(duration * FS) / 2))
    assert np.all(S[..., :edge] == 0) and np.all(S[..., edge + 1501 - int(duration * FS) + 1:] == 0)


def test_timelapse_settings_long_rows(ctx):
    """scripts/timelapse.py:207: duration 2.5 s, padding 4 -> 251 bins; a +-80 s lag axis."""
    x = ccf_like(4, 8001, seed=1)
    freq, S = signal.spectrogram(x, duration=2.5, padding_factor=4, sampling_rate=FS)
    assert S.shape == (4, 251, 8001) and S.dtype == np.float32
    check(S, R.spectrogram(x, FS, 2.5, 4)[1])
    assert freq[np.argmax(S[0].max(axis=1))] == pytest.approx(4.0, abs=0.11)


def test_scaling_window_detrend_options(ctx):
    x = ccf_like(2, 700, seed=2)
    for kw in ({'scaling': 'spectrum'}, {'window': 'hann'}, {'detrend': False}, {'window': ('tukey', 0.5)}):
        _, S = signal.spectrogram(x, duration=1.28, padding_factor=2, sampling_rate=FS, **kw)
        check(S, R.spectrogram(x, FS, 1.28, 2, **kw)[1])
    # a large offset must not leak into the detrended PSD
    _, S0 = signal.spectrogram(x, duration=1.28, sampling_rate=FS)
    _, S1 = signal.spectrogram(x + np.float32(1000.), duration=1.28, sampling_rate=FS)
    assert np.abs(S1 - S0).max() <= 1e-3 * np.abs(S0).max()


def test_shapes_device_and_errors(ctx):
    x = ccf_like(6, 400, seed=3).reshape(2, 3, 400)
    _, S = signal.spectrogram(x, duration=0.64, padding_factor=2, sampling_rate=FS)
    assert S.shape == (2, 3, 33, 400)
    check(S, R.spectrogram(x, FS, 0.64, 2)[1])
    d = ctx.array(x.reshape(6, 400))
    import scipy.signal as ss
    w = ss.get_window(('tukey', .25), 32)
    Sd = ops.spectrogram(ctx, d, 32, 64, w, 1.0 / (FS * (w * w).sum()))
    check(Sd.to_host().reshape(2, 3, 33, 400), R.spectrogram(x, FS, 0.64, 2)[1])
    with pytest.raises(ValueError):
        signal.spectrogram(x, duration=0.1, sampling_rate=FS)              # < 16 samples
    with pytest.raises(XcxError):
        ops.spectrogram(ctx, x, 300, 600, np.ones(300), 1.0)               # 300 > 256 samples: the MFMA kernel's tile
    with pytest.raises(XcxError):
        ops.spectrogram(ctx, x[..., :20], 32, 64, w, 1.0)                  # row shorter than a segment


def test_feeds_correlate2d(ctx):
    """spectrogram tiles -> 2-D correlation peak: a delayed copy peaks at that lag."""
    x = ccf_like(1, 2001, seed=4)
    y = np.roll(x, 7, axis=-1)
    _, Sx = signal.spectrogram(x, duration=2.5, padding_factor=4, sampling_rate=FS)
    _, Sy = signal.spectrogram(y, duration=2.5, padding_factor=4, sampling_rate=FS)
    band = slice(10, 74)                                                   # 64 bins x 2001 lags
    pk, ix = signal.correlate2d_peak(Sy[0, band], Sx[0, band])
    assert tuple(ix) == (63, 2000 + 7) and pk > 0.9


@pytest.mark.parametrize('duration,pf,overlap', [(2.5, None, None), (2.5, 2, .5), (1.0, 4, .9), (5.12, 1, .75),
                                                 (0.5, 3, .2), (20., 2, .5), (2.5, 1, .99)])
def test_spectrogram_mtc_matches_scipy(ctx, duration, pf, overlap):
    """signal/spectrogram.py:199-335: manual segment control (any hop; odd and even nfft)."""
    x = ccf_like(3, 6001, seed=int(duration * 10))
    f, t, S = signal.spectrogram_mtc(x, duration=duration, padding_factor=pf, overlap=overlap, sampling_rate=FS)
    rf, rt, ref = R.spectrogram_mtc(x, FS, duration, pf, overlap)
    assert np.allclose(f, rf) and np.allclose(t, rt)
    check(S, ref)


def test_spectrogram_mtc_argument_rules(ctx):
    x = ccf_like(1, 2001, seed=5)
    with pytest.raises(ValueError):
        signal.spectrogram_mtc(x, duration=0.1, sampling_rate=FS)          # 5 samples < 16
    with pytest.raises(ValueError):
        signal.spectrogram_mtc(x, duration=2.5)                            # array input needs sampling_rate
    with pytest.raises(XcxError):
        ops.spectrogram_seg(ctx, x, 125, 200, 125, np.ones(125), 1.0)      # hop > win_len
    # leading dimensions are kept
    f, t, S = signal.spectrogram_mtc(x.reshape(1, 1, -1), duration=2.5, sampling_rate=FS)
    assert S.shape == (1, 1, f.size, t.size)


def test_long_segments_fall_back_to_framed_transforms(ctx):
    """duration 8 s -> 400-sample segments (beyond the 256-sample MFMA tile)."""
    x = ccf_like(2, 3001, seed=9)
    freq, S = signal.spectrogram(x, duration=8., padding_factor=2, sampling_rate=FS)
    rfreq, ref = R.spectrogram(x, FS, 8., 2)
    assert np.allclose(freq, rfreq)
    check(S, ref)
