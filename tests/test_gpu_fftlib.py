"""B3 seam (`xcorr_amd.fftlib`): numpy.fft signatures over libxcx, against numpy.fft in float64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import xcorr_amd.fftlib as F            # noqa: E402
from oracle import ref_numpy as R       # noqa: E402  (checker only)

TOL = 1e-5


def relerr(a, b):
    return float(np.abs(np.asarray(a) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('shape,axis,n', [((5, 360), -1, None), ((3, 4, 101), -1, None), ((7, 64), 0, None),
                                          ((2, 250), -1, 257), ((2, 250), -1, 100), ((6, 9, 16), 1, 12)])
def test_fft_ifft_match_numpy(shape, axis, n):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    got = F.fft(x, n=n, axis=axis)
    ref = np.fft.fft(x.astype(np.complex128), n=n, axis=axis)
    assert got.dtype == np.complex64 and got.shape == ref.shape
    assert relerr(got, ref) < TOL
    assert relerr(F.ifft(x, n=n, axis=axis), np.fft.ifft(x.astype(np.complex128), n=n, axis=axis)) < TOL
    assert relerr(F.fft(x, n=n, axis=axis, norm='ortho'), np.fft.fft(x.astype(np.complex128), n=n, axis=axis, norm='ortho')) < TOL


@pytest.mark.parametrize('size,n', [(360, None), (359, None), (1000, 1024), (1001, 777), (6001, None)])
def test_rfft_irfft_match_numpy(size, n):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, size)).astype(np.float32)
    got = F.rfft(x, n=n)
    ref = np.fft.rfft(x.astype(np.float64), n=n)
    assert got.shape == ref.shape and relerr(got, ref) < TOL
    m = ref.shape[-1]
    for nout in (None, 2 * (m - 1) + 1, 2 * (m - 1) - 2):
        back = F.irfft(ref.astype(np.complex64), n=nout)
        refb = np.fft.irfft(ref, n=nout)
        assert back.dtype == np.float32 and back.shape == refb.shape
        assert relerr(back, refb) < TOL


def test_fft2_ifft2_and_helpers():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 12, 40)).astype(np.float32)
    s = (23, 79)                                   # correlate.py:240: zero-padded to (2F-1, 2L-1)
    got = F.fft2(x, s=s, axes=(-2, -1))
    ref = np.fft.fft2(x.astype(np.float64), s=s, axes=(-2, -1))
    assert relerr(got, ref) < TOL
    assert relerr(F.ifft2(ref.astype(np.complex64)), np.fft.ifft2(ref)) < TOL
    assert F.fftshift is np.fft.fftshift and F.fftfreq is np.fft.fftfreq
    np.testing.assert_array_equal(F.fftshift(F.fftfreq(7, 0.5)), np.fft.fftshift(np.fft.fftfreq(7, 0.5)))


def test_reference_cc_formula_through_the_seam():
    """xcorr/util/cc.py:60-79 written against the backend module reproduces the oracle's cc."""
    n = 1500
    tr = R.synthetic_traces(2, n)
    x, y = tr[0], tr[1]
    xx = np.concatenate([np.zeros(n - 1, np.float32), x])
    yy = np.concatenate([np.zeros(n - 1, np.float32), y])
    fg = F.fft(xx) * np.conjugate(F.fft(yy))
    fg = fg / (np.linalg.norm(xx) * np.linalg.norm(yy))
    got = F.fftshift(np.real(F.ifft(fg)))
    ref = R.cc(x, y, normalize=True, dtype=np.float64)
    assert relerr(got, ref) < TOL
