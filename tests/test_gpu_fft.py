"""GPU: labelled-FFT kernels (xcorr/signal/fft.py, a6) for arbitrary lengths against
numpy float64; tolerance 1e-5 of max|result| (float32 path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
from xcorr_amd import ops, signal          # noqa: E402
from xcorr_amd._ffi import XcxError        # noqa: E402

TOL = 1e-5
# primes, prime powers, 2 x prime (the 359 998-type lag axis), smooth and odd sizes
SIZES = [2, 3, 4, 5, 7, 16, 17, 100, 127, 128, 1000, 1021, 2 * 4999, 6001, 12001, 16384, 65537]


def relerr(got, ref):
    ref = np.asarray(ref)
    return float(np.abs(np.asarray(got) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize('n', SIZES)
def test_fft_ifft_any_length(ctx, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    x[0] += 2.0                                      # DC offset
    F = ops.fft(ctx, x)
    assert F.dtype == np.complex64 and F.shape == x.shape
    assert relerr(F, np.fft.fft(x.astype(np.complex128), axis=-1)) < TOL
    f = ops.ifft(ctx, x)
    assert relerr(f, np.fft.ifft(x.astype(np.complex128), axis=-1)) < TOL
    # reference kernels (with the fftshift) and round trip
    assert relerr(signal.fft(x), R.fft_kernel(x)) < TOL
    assert relerr(signal.ifft(x), R.ifft_kernel(x)) < TOL
    # the reference's ifft applies fftshift (not ifftshift, fft.py:240): the pair only
    # round-trips for even lengths -- mirrored, not repaired
    if n % 2 == 0:
        assert relerr(signal.ifft(signal.fft(x)), x) < 2 * TOL


@pytest.mark.parametrize('size', SIZES + [9, 1001, 8191])
def test_rfft_irfft_any_length(ctx, size):
    rng = np.random.default_rng(size)
    x = rng.standard_normal((4, size)).astype(np.float32) + 0.5
    X = ops.rfft(ctx, x)
    ref = R.rfft_kernel(x)
    assert X.shape == ref.shape and X.dtype == np.complex64
    assert relerr(X, ref) < TOL
    if size >= 4:
        y = ops.irfft(ctx, ref.astype(np.complex64))
        yref = R.irfft_kernel(ref.astype(np.complex64))
        assert y.shape == yref.shape and y.dtype == np.float32
        assert relerr(y, yref) < TOL
        # imaginary parts of DC / Nyquist are ignored, like numpy
        Z = ref.astype(np.complex64).copy()
        Z[:, 0] += 3j
        Z[:, -1] -= 2j
        assert relerr(ops.irfft(ctx, Z), yref) < TOL


def test_lag_axis_sized_rows(ctx):
    """rows as long as a +-600 s lag axis at 50 Hz (60 001 samples, odd -> 60 000 used)
    and the half-hour window (90 000)."""
    rng = np.random.default_rng(5)
    for size in (60001, 90000):
        x = rng.standard_normal((2, size)).astype(np.float32)
        assert relerr(ops.rfft(ctx, x), R.rfft_kernel(x)) < TOL
    x = rng.standard_normal((1, 2 * 89989)).astype(np.float32)     # 2 x prime
    X = ops.rfft(ctx, x)
    assert relerr(X, R.rfft_kernel(x)) < TOL
    assert relerr(ops.irfft(ctx, X), x) < 2 * TOL


def test_batches_leading_dims_and_coords(ctx):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 3, 5, 300)).astype(np.float32)
    X = signal.rfft(x)
    assert X.shape == (2, 3, 5, 151) and X.dtype == np.complex64
    assert relerr(X, R.rfft_kernel(x)) < TOL
    big = rng.standard_normal((700, 64)).astype(np.float32)          # more rows than one pass
    assert relerr(ops.rfft(ctx, big), R.rfft_kernel(big)) < TOL
    t = np.arange(6001) * 0.02 - 60.0
    assert np.array_equal(signal.fftops.rfft_freq(t), R.rfft_freq(t))
    assert signal.fftops.rfft_freq(t)[-1] == 25.0
    assert np.allclose(signal.fftops.fft_freq(8, 0.02), R.fft_freq(8, 0.02))
    f = R.rfft_freq(t)
    assert np.allclose(signal.fftops.irfft_time(f, -60.0), R.irfft_time(f, -60.0))


@pytest.mark.parametrize('n', [2, 3, 8, 9, 127, 1000, 6001, 12001, 60000])
def test_hilbert_envelope(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n)).astype(np.float32)
    if n >= 100:
        t = np.arange(n)
        x[0] = np.exp(-0.5 * ((t - n / 2) / (n / 20)) ** 2) * np.cos(2 * np.pi * t / 13.0)   # wave packet
    z = ops.hilbert(ctx, x)
    ref = R.hilbert(x)
    assert z.dtype == np.complex64 and z.shape == x.shape
    assert relerr(z, ref) < TOL
    e = ops.hilbert(ctx, x, envelope=True)
    assert e.dtype == np.float32
    assert relerr(e, np.abs(ref)) < TOL
    assert relerr(signal.envelope(x), R.envelope(x)) < TOL


def test_snr_with_envelope(ctx):
    rng = np.random.default_rng(3)
    n = 6001
    t = np.arange(n)
    x = (0.1 * rng.standard_normal((5, n))).astype(np.float32)
    x += (np.exp(-0.5 * ((t - 3300) / 60.0) ** 2) * np.cos(2 * np.pi * t / 17.0)).astype(np.float32)
    sig = (t > 2800) & (t < 3800)
    noi = t < 2000
    r, s, nn, a = signal.snr(x, sig, noi, extend=True, envelope=True)
    rr, rs, rn, ra = R.snr_envelope(x, sig, noi)
    assert np.allclose(s, rs, rtol=1e-5) and np.allclose(nn, rn, rtol=1e-5) and np.allclose(r, rr, rtol=2e-5)
    assert np.all(np.abs(a - ra) <= 1)          # envelope maxima are flat: allow a one-sample tie
    assert np.all(s >= signal.snr(x, sig, noi, extend=True)[1] * (1 - 1e-5))   # envelope >= |x|


def test_errors(ctx):
    with pytest.raises(XcxError):
        ops.fft(ctx, np.zeros((1, 1), np.complex64))
    with pytest.raises(TypeError):
        signal.fft(np.zeros(8, np.float32), dtype=np.dtype('float32'))
    with pytest.raises(TypeError):
        signal.rfft(np.zeros(8, np.float32), dtype=np.dtype('complex64'))


def test_full_lag_axis_transforms(ctx):
    """The lag axis of the headline workload: 359 999 lags -> rfft of 359 998 = 2 x 179 999 (prime)
    samples, hilbert / envelope of all 359 999 = 599 x 601.  Longer than one chirp convolution of
    the engine: the transform is split into partial transforms over input / output blocks."""
    rng = np.random.default_rng(8)
    n = 359999
    t = np.arange(n)
    x = (0.05 * rng.standard_normal((2, n))).astype(np.float32)
    x += (np.exp(-0.5 * ((t - 179800) / 900.0) ** 2) * np.cos(2 * np.pi * t / 23.0)).astype(np.float32)
    X = ops.rfft(ctx, x)
    ref = np.fft.rfft(x[:, :n - 1].astype(np.float64), axis=-1)
    assert X.shape == ref.shape == (2, 180000) and relerr(X, ref) < TOL
    back = ops.irfft(ctx, X)
    assert back.shape == (2, n - 1) and relerr(back, x[:, :n - 1]) < 2 * TOL
    z = ops.hilbert(ctx, x)
    zr = R.hilbert(x)
    assert z.shape == (2, n) and relerr(z, zr) < TOL
    e = ops.hilbert(ctx, x, envelope=True)
    assert relerr(e, np.abs(zr)) < TOL
    # complex transform of the same length (2 x 2 partial transforms), device-resident rows
    c = (x[:1] + 1j * np.roll(x[:1], 1000, axis=-1)).astype(np.complex64)
    assert relerr(ops.fft(ctx, c), np.fft.fft(c.astype(np.complex128), axis=-1)) < TOL
    assert relerr(ops.ifft(ctx, c), np.fft.ifft(c.astype(np.complex128), axis=-1)) < TOL


def test_split_transform_boundaries(ctx):
    """Lengths right at and beyond the single-convolution limit (np = nin + nout - 1 <= 368640)."""
    rng = np.random.default_rng(9)
    for n in (184320, 184321, 245760, 245762, 400001):
        x = rng.standard_normal((1, n)).astype(np.float32)
        X = ops.rfft(ctx, x)
        m = n - (n & 1)
        assert relerr(X, np.fft.rfft(x[:, :m].astype(np.float64), axis=-1)) < TOL
    c = (rng.standard_normal((1, 184321)) + 1j * rng.standard_normal((1, 184321))).astype(np.complex64)
    assert relerr(ops.fft(ctx, c), np.fft.fft(c.astype(np.complex128), axis=-1)) < TOL


def test_transforms_beyond_every_single_transform(ctx):
    """Axes whose chirp convolution exceeds the longest transform (2 max(nin, nout) - 1 > 786 432) run as block-decomposed
    convolutions (fftn.hip: bluestein_seg over seg_run's convolution mode): day-long rows (rfft of 4 320 000 samples = a
    complex transform of 2 160 000 points), odd and prime-ish lengths, complex transforms, hilbert / envelope; the
    stretch in between (np > 737 280 but one transform still fits 2n - 1) keeps the partial-transform route."""
    rng = np.random.default_rng(19)
    for n in (1000002, 1000003, 4320000, 760000):
        x = rng.standard_normal((2, n)).astype(np.float32)
        x[:, n // 3: n // 3 + 500] += 5.0
        m = n - (n & 1)
        X = ops.rfft(ctx, x)
        ref = np.fft.rfft(x[:, :m].astype(np.float64), axis=-1)
        assert X.shape == ref.shape and relerr(X, ref) < TOL, n
        back = ops.irfft(ctx, X)
        assert back.shape == (2, m) and relerr(back, x[:, :m]) < 2 * TOL, n
    n = 400001                                             # complex, odd: nin = nout = n, 2n - 1 = 800 001
    c = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    assert relerr(ops.fft(ctx, c), np.fft.fft(c.astype(np.complex128), axis=-1)) < TOL
    assert relerr(ops.ifft(ctx, c), np.fft.ifft(c.astype(np.complex128), axis=-1)) < TOL
    n = 800001
    x = (rng.standard_normal((1, n)) * np.hanning(n)).astype(np.float32)
    zr = R.hilbert(x)
    assert relerr(ops.hilbert(ctx, x), zr) < TOL
    assert relerr(ops.hilbert(ctx, x, envelope=True), np.abs(zr)) < TOL


def test_device_resident_transforms(ctx):
    """DeviceArray in -> DeviceArray out (complex rows as interleaved float32 pairs): the same values as the host
    path, nothing staged through the host."""
    rng = np.random.default_rng(12)
    for n in (4096, 4999, 10006):
        x = rng.standard_normal((3, n)).astype(np.float32)
        m = (n - (n & 1)) // 2 + 1
        d = ctx.array(x)
        X = ops.rfft(ctx, d)
        assert X.shape == (3, 2 * m)
        Xh = X.to_host().view(np.complex64)
        assert np.array_equal(Xh, ops.rfft(ctx, x))
        xr = ops.irfft(ctx, X)
        assert xr.shape == (3, 2 * m - 2)
        assert np.array_equal(xr.to_host(), ops.irfft(ctx, Xh))
        z = (x + 1j * rng.standard_normal((3, n))).astype(np.complex64)
        dz = ctx.array(z.view(np.float32))
        Z = ops.fft(ctx, dz)
        assert Z.shape == (3, 2 * n)
        assert np.array_equal(Z.to_host().view(np.complex64), ops.fft(ctx, z))
        assert np.array_equal(ops.ifft(ctx, Z).to_host().view(np.complex64), ops.ifft(ctx, Z.to_host().view(np.complex64)))
        h = ops.hilbert(ctx, d)
        assert h.shape == (3, 2 * n)
        assert np.array_equal(h.to_host().view(np.complex64), ops.hilbert(ctx, x))
        e = ops.hilbert(ctx, d, envelope=True)
        assert e.shape == (3, n)
        assert np.array_equal(e.to_host(), ops.hilbert(ctx, x, envelope=True))
        for b in (d, X, xr, dz, Z, h, e):
            b.free()
    with pytest.raises(ValueError):
        ops.fft(ctx, ctx.array(np.zeros((2, 5), np.float32)))
