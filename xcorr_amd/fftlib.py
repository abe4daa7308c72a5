"""
B3 seam: an FFT backend module with numpy.fft signatures over libxcx.

The reference selects its transform library at import time
(``xcorr/util/cc.py:12-15``, ``xcorr/signal/correlate.py:14-17``,
``xcorr/signal/fft.py:15-18``)::

    try:
        from pyfftw.interfaces import numpy_fft as fftlib
    except ModuleNotFoundError:
        from numpy import fft as fftlib

A maintainer can add ``import xcorr_amd.fftlib as fftlib`` as a third choice:
``fft, ifft, rfft, irfft, fft2, ifft2, fftshift, fftfreq`` below take the
arguments the reference passes (``n=`` / ``s=`` zero-padding or cropping,
``axis=`` / ``axes=``) and return numpy arrays (complex64 / float32: the
transforms run in float32 on the GPU, any length, chirp-z for non-smooth sizes).

This is the SLOW seam, provided for completeness: every call pays a host ->
device -> host round trip and an any-length transform, and none of the fusion
that makes the hot path fast (pad -> FFT -> multiply -> normalise -> IFFT ->
shift -> clip in one device-resident call, with a chosen transform length)
survives behind an ``fft(x)`` signature.  Use ``xcorr_amd.util.cc.cc`` /
``signal.correlate1d`` / ``core.process`` for speed.

There is no CPU fallback: without libxcx.so or a GPU every transform raises
``XcxError``.  ``fftshift`` / ``ifftshift`` / ``fftfreq`` / ``rfftfreq`` are index
and label arithmetic on the host (numpy's own).
"""
import numpy as np

from . import ops

__all__ = ['fft', 'ifft', 'rfft', 'irfft', 'fft2', 'ifft2', 'fftshift', 'ifftshift', 'fftfreq', 'rfftfreq']

fftshift = np.fft.fftshift
ifftshift = np.fft.ifftshift
fftfreq = np.fft.fftfreq
rfftfreq = np.fft.rfftfreq


def _fit(a, n, axis):
    """numpy's ``n=`` rule: crop or zero-pad `a` to length n along `axis`."""
    a = np.asarray(a)
    if n is None or n == a.shape[axis]:
        return a
    if n < 1:
        raise ValueError(f'Invalid number of FFT data points ({n}) specified.')
    if n < a.shape[axis]:
        idx = [slice(None)] * a.ndim
        idx[axis] = slice(0, n)
        return a[tuple(idx)]
    pad = [(0, 0)] * a.ndim
    pad[axis] = (0, n - a.shape[axis])
    return np.pad(a, pad)


def _along(fn, a, axis):
    a = np.moveaxis(a, axis, -1)
    return np.moveaxis(fn(None, a), -1, axis)


def _norm(y, norm, n, inverse):
    if norm in (None, 'backward'):
        return y
    if norm == 'ortho':
        return y * (np.sqrt(n) if inverse else 1.0 / np.sqrt(n)).astype(np.float32)
    if norm == 'forward':
        return y * np.float32(n if inverse else 1.0 / n)
    raise ValueError(f'Invalid norm value {norm}; should be "backward", "ortho" or "forward".')


def fft(a, n=None, axis=-1, norm=None):
    """``numpy.fft.fft`` (complex64 result)."""
    a = _fit(a, n, axis)
    return _norm(_along(ops.fft, a, axis), norm, a.shape[axis], False)


def ifft(a, n=None, axis=-1, norm=None):
    """``numpy.fft.ifft`` (complex64 result)."""
    a = _fit(a, n, axis)
    return _norm(_along(ops.ifft, a, axis), norm, a.shape[axis], True)


def rfft(a, n=None, axis=-1, norm=None):
    """``numpy.fft.rfft`` of real input (complex64, n // 2 + 1 bins).  libxcx transforms even lengths
    (the reference's own rfft drops the last sample of an odd axis, signal/fft.py:357); an odd
    length goes through the complex transform."""
    a = _fit(np.real(a), n, axis)
    m = a.shape[axis]
    if m % 2 == 0:
        y = _along(ops.rfft, a, axis)
    else:
        y = np.moveaxis(np.moveaxis(_along(ops.fft, a, axis), axis, -1)[..., :m // 2 + 1], -1, axis)
    return _norm(y, norm, m, False)


def irfft(a, n=None, axis=-1, norm=None):
    """``numpy.fft.irfft``: Hermitian half spectrum -> n real samples (default 2 (m - 1))."""
    a = np.asarray(a)
    m = a.shape[axis]
    if n is None:
        n = 2 * (m - 1)
    if n < 1:
        raise ValueError(f'Invalid number of data points ({n}) specified.')
    a = _fit(a, n // 2 + 1, axis)
    if n % 2 == 0:
        y = _along(ops.irfft, a, axis)
    else:                                   # odd output length: rebuild the full Hermitian spectrum
        h = np.moveaxis(a, axis, -1)
        full = np.concatenate([h, np.conj(h[..., :0:-1])], axis=-1)
        y = np.moveaxis(np.real(ops.ifft(None, full)), -1, axis)
    return _norm(y.astype(np.float32, copy=False), norm, n, True)


def _axes2(a, s, axes):
    a = np.asarray(a)
    axes = tuple(axes)
    if len(axes) != 2:
        raise ValueError('fft2 / ifft2 take exactly two axes')
    if s is not None:
        if len(s) != 2:
            raise ValueError('Shape and axes have different lengths.')
        for n, ax in zip(s, axes):
            a = _fit(a, n, ax)
    return a, axes


def fft2(a, s=None, axes=(-2, -1), norm=None):
    """``numpy.fft.fft2``: the 1-D transform along each of the two axes (``correlate.py:240-241``)."""
    a, axes = _axes2(a, s, axes)
    return fft(fft(a, axis=axes[1], norm=norm), axis=axes[0], norm=norm)


def ifft2(a, s=None, axes=(-2, -1), norm=None):
    """``numpy.fft.ifft2`` (``correlate.py:243``)."""
    a, axes = _axes2(a, s, axes)
    return ifft(ifft(a, axis=axes[1], norm=norm), axis=axes[0], norm=norm)
