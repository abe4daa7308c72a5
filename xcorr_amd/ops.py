"""
Row-wise signal operations on plain arrays (core dimension last), executed by
libxcx.so.  Inputs are float32 numpy arrays ``[..., n]`` (a processed copy is
returned) or DeviceArrays ``[batch, n]`` (processed in place in HBM).

These are the numpy-level kernels underneath the reference's
``xr.apply_ufunc`` closures; ``xcorr_amd.signal`` wraps them with the
reference's labelled-array interface.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .engine import DeviceArray, default_context

__all__ = ['copy', 'norm1d', 'demean', 'detrend_linear', 'taper', 'taper_table', 'detrend_nan', 'unbias', 'post_prep',
           'onebit', 'sosfiltfilt', 'sosfilt', 'sosfilt_zerophase', 'timeshift', 'snr', 'correlate2d',
           'fft', 'ifft', 'rfft', 'irfft', 'hilbert', 'decimate', 'resample', 'lanczos', 'block_rms', 'spectrogram', 'spectrogram_seg',
           'ram_normalize', 'scale_rows', 'absmax_rows', 'rotate2']


def _prep(x, ctx=None):
    """-> (buffer, batch, n, on_device, restore).  Host rows are copied (the ops work in place on their argument, the
    mirrors of the reference return a new array): for a megabyte and more the copy lives in page-locked memory from
    the CALLER's context pool (`ctx`; the default context only when the op was called without one), so that the
    library's H2D / D2H of it run at link speed -- through pageable memory the two transfers of a 68 MB block of CCF
    rows took 17 of a call's 21 ms.  An op given an explicit context never opens another one: a rank on GPU k must
    not create streams, events and pinned pools on GPU 0."""
    if isinstance(x, DeviceArray):
        n = x.shape[-1]
        batch = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        return x, batch, n, 1, None
    src = np.asarray(x)
    if src.ndim == 0:
        raise ValueError('array with at least one dimension required')
    if src.size * 4 >= (1 << 20):
        a = (ctx if ctx is not None else default_context()).pinned_empty(src.shape, np.float32)
        np.copyto(a, src, casting='unsafe')
    else:
        a = np.array(src, dtype=np.float32, order='C', copy=True)
    if a.ndim == 0:
        raise ValueError('array with at least one dimension required')
    n = a.shape[-1]
    batch = int(np.prod(a.shape[:-1])) if a.ndim > 1 else 1
    return a, batch, n, 0, a


def _host_out(ctx, shape, dtype):
    """Host result array: page-locked from the context's pool for a megabyte and more (the library's D2H into it runs
    at link speed; pageable memory takes 8 ms per 68 MB block of CCF rows), ordinary numpy memory below."""
    dtype = np.dtype(dtype)
    if int(np.prod(shape, dtype=np.int64)) * dtype.itemsize >= (1 << 20):
        return ctx.pinned_empty(tuple(shape), dtype)
    return np.empty(shape, dtype)


def copy(ctx, dst, src):
    check(_ffi.lib().xcx_memcpy_d2d(ctx.handle, ptr(dst), ptr(src), src.nbytes))
    return dst


def _ctx(ctx, x):
    if ctx is not None:
        return ctx
    return x.ctx if isinstance(x, DeviceArray) else default_context()


def norm1d(ctx, x):
    """x / ||x||_2 (xcorr/signal/normalize.py:73-80)."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_norm1d(ctx.handle, ptr(buf), b, n, dev))
    return buf


def _detrend_to(ctx, x, out, linear):
    """device rows x -> device rows out (x untouched)."""
    if not isinstance(x, DeviceArray) or not isinstance(out, DeviceArray) or out.nbytes != x.nbytes:
        raise ValueError('out= needs DeviceArray source and destination of the same size')
    n = x.shape[-1]
    b = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
    check(_ffi.lib().xcx_detrend_to(ctx.handle, ptr(x), ptr(out), b, n, int(linear)))
    return out


def demean(ctx, x, out=None, cols=None):
    """x - mean(x) (xcorr/signal/detrend.py:110).  In place; with ``out`` (device arrays) the
    source stays untouched and no copy of it is needed first.  ``cols=(c0, c1)`` (with ``out``):
    write only those columns -- the mean is still that of the whole row."""
    ctx = _ctx(ctx, x)
    if cols is not None:
        if out is None or not isinstance(x, DeviceArray) or not isinstance(out, DeviceArray) or out.nbytes != x.nbytes:
            raise ValueError('cols= needs DeviceArray source and out= of the same size')
        n = x.shape[-1]
        b = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        check(_ffi.lib().xcx_detrend_to_cols(ctx.handle, ptr(x), ptr(out), b, n, 0, int(cols[0]), int(cols[1])))
        return out
    if out is not None:
        return _detrend_to(ctx, x, out, 0)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_demean(ctx.handle, ptr(buf), b, n, dev))
    return buf


def detrend_linear(ctx, x):
    """Remove the least-squares line (xcorr/signal/detrend.py:87-99)."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_detrend_linear(ctx.handle, ptr(buf), b, n, dev))
    return buf


def taper(ctx, x, nmax, wtype='hann'):
    """Taper both flanks over nmax samples (xcorr/signal/window.py:92-106;
    wtype 'cosine' = ObsPy Trace.taper flank)."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    wt = {'hann': 0, 'cosine': 1}[wtype]
    check(_ffi.lib().xcx_taper(ctx.handle, ptr(buf), b, n, int(nmax), wt, dev))
    return buf


def unbias(ctx, x, npts, index_min=0, index_max=0):
    """x * npts/(npts-|lag|) (xcorr/signal/unbias.py:73-85)."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    index_max = index_max or 2 * npts - 1
    if n != index_max - index_min:
        raise ValueError('last dimension does not match the lag slice')
    check(_ffi.lib().xcx_unbias(ctx.handle, ptr(buf), b, int(npts),
                                int(index_min), int(index_max), dev))
    return buf


def detrend_nan(ctx, x, linear=False):
    """detrend with skipna=True (signal/detrend.py:87-110): fit on the finite samples, NaN stays NaN."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_detrend_nan(ctx.handle, ptr(buf), b, n, int(bool(linear)), dev))
    return buf


def taper_table(ctx, x, flanks):
    """Taper with explicit flanks: flanks = window(2*nmax) (float32, host): first half on the left,
    second half on the right (signal/window.py:92-106 for any scipy window)."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    fl = np.ascontiguousarray(flanks, dtype=np.float32)
    if fl.ndim != 1 or fl.size % 2:
        raise ValueError('flanks must hold 2*nmax samples')
    check(_ffi.lib().xcx_taper_table(ctx.handle, ptr(buf), b, n, ptr(fl), fl.size // 2, dev))
    return buf


def post_prep(ctx, x, npts, index_min=0, index_max=0, unbiased=True, taper_nmax=0, wtype='hann'):
    """unbias -> demean -> taper in one op (core/postprocess.py:133-136), in place; bit-identical
    to ``taper(demean(unbias(x)))`` with two passes over the rows instead of five."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    index_max = index_max or 2 * npts - 1
    if n != index_max - index_min:
        raise ValueError('last dimension does not match the lag slice')
    wt = {'hann': 0, 'cosine': 1}[wtype]
    check(_ffi.lib().xcx_post_prep(ctx.handle, ptr(buf), b, int(npts), int(index_min), int(index_max),
                                   int(bool(unbiased)), int(taper_nmax), wt, dev))
    return buf


def scale_rows(ctx, x, factor):
    """x[r] *= factor[r] in place (one float64 factor per row, or a scalar): the division by a sensitivity or a
    norm of the ObsPy-named preprocess ops on device-resident (or host) rows."""
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    f = np.ascontiguousarray(np.broadcast_to(np.asarray(factor, dtype=np.float64).reshape(-1), (b,)))
    check(_ffi.lib().xcx_scale_rows(ctx.handle, ptr(buf), b, n, ptr(f), dev))
    return buf


def rotate2(ctx, a, b, c00, c01, c10, c11):
    """(a, b) <- (c00 a + c01 b, c10 a + c11 b): two components of equal shape, both device-resident (in place) or
    both host arrays (copies are returned).  The horizontal rotations of ``Stream.rotate`` (stream/process.py:70-73)."""
    if isinstance(a, DeviceArray) != isinstance(b, DeviceArray):
        raise ValueError('rotate2: both components on the device or both on the host')
    if tuple(a.shape) != tuple(b.shape):
        raise TypeError('North and East component have different length.')
    ctx = _ctx(ctx, a)
    ba, nb, n, dev, _ = _prep(a, ctx)
    bb, _, _, _, _ = _prep(b, ctx)
    c4 = np.array([c00, c01, c10, c11], np.float64)
    check(_ffi.lib().xcx_rotate2(ctx.handle, ptr(ba), ptr(bb), nb, n, ptr(c4), dev))
    return ba, bb


def absmax_rows(ctx, x):
    """max |x[r]| per row -> host float32 array [rows]."""
    ctx = _ctx(ctx, x)
    if isinstance(x, DeviceArray):
        src, dev = x, 1
        n = x.shape[-1]
        b = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
    else:
        src = np.ascontiguousarray(x, dtype=np.float32)
        dev, n = 0, src.shape[-1]
        b = int(np.prod(src.shape[:-1])) if src.ndim > 1 else 1
    out = np.empty(b, np.float32)
    check(_ffi.lib().xcx_absmax_rows(ctx.handle, ptr(src), b, n, ptr(out), dev))
    return out


def onebit(ctx, x):
    ctx = _ctx(ctx, x)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_onebit(ctx.handle, ptr(buf), b, n, dev))
    return buf


def _sos(sos):
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError('sos array must be shape (n_sections, 6)')
    return sos


def sosfiltfilt(ctx, x, sos):
    """scipy.signal.sosfiltfilt(sos, x, axis=-1) (xcorr/signal/filter.py:95)."""
    ctx = _ctx(ctx, x)
    sos = _sos(sos)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_sosfiltfilt(ctx.handle, ptr(buf), b, n, ptr(sos),
                                     sos.shape[0], dev))
    return buf


def sosfilt_warmup(sos):
    """Samples after which the cascade has forgotten its start state to below float64 rounding
    (xcx_sosfilt_warmup: ||A^warm||_inf <= 2^-64); -1 when it never does."""
    sos = _sos(sos)
    w = C.c_long(0)
    check(_ffi.lib().xcx_sosfilt_warmup(ptr(sos), sos.shape[0], C.byref(w)))
    return w.value


def sosfilt(ctx, x, sos, cols=None):
    """scipy.signal.sosfilt(sos, x, axis=-1) from rest.  ``cols=(c0, c1)`` (device rows): only those
    columns are produced, from a warm-up that starts ``sosfilt_warmup(sos)`` samples before c0 (the
    returned first column; what lies between it and c0 is overwritten with intermediate values)."""
    ctx = _ctx(ctx, x)
    sos = _sos(sos)
    if cols is not None:
        if not isinstance(x, DeviceArray) or len(x.shape) != 2:
            raise ValueError('cols= needs a 2-D DeviceArray')
        first = C.c_long(0)
        check(_ffi.lib().xcx_sosfilt_cols(ctx.handle, ptr(x), x.shape[0], x.shape[1], ptr(sos), sos.shape[0],
                                          int(cols[0]), int(cols[1]), C.byref(first)))
        return first.value
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_sosfilt(ctx.handle, ptr(buf), b, n, ptr(sos),
                                 sos.shape[0], dev))
    return buf


def sosfilt_zerophase(ctx, x, sos):
    """ObsPy ``zerophase=True``: sosfilt from rest, then sosfilt of the reversed
    signal from rest (no padding; not scipy's sosfiltfilt)."""
    ctx = _ctx(ctx, x)
    sos = _sos(sos)
    buf, b, n, dev, _ = _prep(x, ctx)
    check(_ffi.lib().xcx_sosfilt_zerophase(ctx.handle, ptr(buf), b, n, ptr(sos),
                                           sos.shape[0], dev))
    return buf


def timeshift(ctx, x, delay, sampling_rate, out=None, fast=True):
    """Fractional time shift (xcorr/signal/timeshift.py:90-96).  fast=True
    (rfft/irfft): rows of length `size` -> rows of length size - (size & 1).
    fast=False (fft/ifft, signal/fft.py:104-113,239-241): every sample kept;
    for an odd length the reference's result carries a cos(2 pi m / size)
    factor (its second fftshift is not the inverse of the first), reproduced.
    `delay` is a scalar or one value per row (seconds)."""
    ctx = _ctx(ctx, x)
    if isinstance(x, DeviceArray):
        size = x.shape[-1]
        batch = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        n = size - (size & 1) if fast else size
        if out is None:
            out = DeviceArray(ctx, (batch, n))
        src, dev, lead = x, 1, None
    else:
        src = np.ascontiguousarray(x, dtype=np.float32)
        size = src.shape[-1]
        lead = src.shape[:-1]
        batch = int(np.prod(lead)) if src.ndim > 1 else 1
        n = size - (size & 1) if fast else size
        out = _host_out(ctx, lead + (n,), np.float32)
        dev = 0
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(delay, dtype=np.float64),
                                             lead if lead is not None else (batch,)).reshape(-1))
    if d.size != batch:
        raise ValueError('delay must be a scalar or one value per row')
    if fast:
        check(_ffi.lib().xcx_timeshift(ctx.handle, ptr(src), ptr(out), batch, size, ptr(d),
                                       float(sampling_rate), dev))
    else:
        check(_ffi.lib().xcx_timeshift_fft(ctx.handle, ptr(src), ptr(out), batch, size, ptr(d),
                                           1.0 / float(sampling_rate), dev))
    return out


def snr(ctx, x, signal, noise):
    """Per row: (s, n, argmax) with s = max|x| over the boolean `signal` mask,
    n = rms over the `noise` mask (xcorr/signal/snr.py:107-116, envelope=False)."""
    ctx = _ctx(ctx, x)
    if isinstance(x, DeviceArray):
        src, dev = x, 1
        nn = x.shape[-1]
        lead = x.shape[:-1]
    else:
        src = np.ascontiguousarray(x, dtype=np.float32)
        dev = 0
        nn = src.shape[-1]
        lead = src.shape[:-1]
    batch = int(np.prod(lead)) if len(lead) else 1
    sm = np.ascontiguousarray(signal, dtype=np.uint8)
    nm = np.ascontiguousarray(noise, dtype=np.uint8)
    if sm.shape != (nn,) or nm.shape != (nn,):
        raise ValueError('masks must be 1-D with the length of the last axis')
    if not sm.any() or not nm.any():
        raise ValueError('empty signal or noise window')
    s = np.empty(batch, np.float32)
    n = np.empty(batch, np.float32)
    a = np.empty(batch, np.int32)
    check(_ffi.lib().xcx_snr(ctx.handle, ptr(src), batch, nn, ptr(sm), ptr(nm), ptr(s), ptr(n), ptr(a), dev))
    return s.reshape(lead), n.reshape(lead), a.reshape(lead)


def correlate2d(ctx, tiles, pairs, normalize=True, full=False):
    """2-D cross-correlation of tile pairs (xcorr/signal/correlate.py:162-291).
    tiles: [T, F, L] float32 (numpy or DeviceArray); pairs: [P, 2] tile indices.
    Returns (peak, index[P, 2]) and, with full=True, also cc2[P, 2F-1, 2L-1]."""
    ctx = _ctx(ctx, tiles)
    if isinstance(tiles, DeviceArray):
        src, dev = tiles, 1
    else:
        src, dev = np.ascontiguousarray(tiles, dtype=np.float32), 0
    if len(src.shape) != 3:
        raise ValueError('tiles should be [T, F, L]')
    T, F, L = (int(v) for v in src.shape)
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    P = pr.shape[0]
    peak = np.empty(P, np.float32)
    idx = np.empty((P, 2), np.int32)
    out = None
    if full:
        shp = (P, 2 * F - 1, 2 * L - 1)
        out = ctx.empty(shp) if dev else _host_out(ctx, shp, np.float32)
    check(_ffi.lib().xcx_cc2d(ctx.handle, ptr(src), T, F, L, ptr(pr), P, int(bool(normalize)), ptr(peak), ptr(idx),
                              ptr(out) if out is not None else None, dev))
    return (peak, idx, out) if full else (peak, idx)


def _fft_call(name, ctx, x, in_dtype, out_dtype, out_len):
    if isinstance(x, DeviceArray):
        # device-resident rows: float32 DeviceArrays, complex rows as interleaved (re, im) pairs -- a complex row of
        # m values is 2 m floats; the result is a new DeviceArray of the same convention and nothing crosses PCIe
        ctx = _ctx(ctx, x)
        cin, cout = in_dtype == np.complex64, out_dtype == np.complex64
        width = x.shape[-1]
        if cin and width % 2:
            raise ValueError('complex device rows are interleaved (re, im) pairs: an even number of floats per row')
        size = width // 2 if cin else width
        batch = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        m = out_len(size)
        out = DeviceArray(ctx, (batch, 2 * m if cout else m))
        if batch and m:
            check(getattr(_ffi.lib(), name)(ctx.handle, ptr(x), ptr(out), batch, size, 1))
        return out
    ctx = ctx if ctx is not None else default_context()
    a = np.ascontiguousarray(x, dtype=in_dtype)
    if a.ndim == 0:
        raise ValueError('array with at least one dimension required')
    size = a.shape[-1]
    lead = a.shape[:-1]
    batch = int(np.prod(lead)) if lead else 1
    out = _host_out(ctx, lead + (out_len(size),), out_dtype)
    if batch and out.size:
        check(getattr(_ffi.lib(), name)(ctx.handle, ptr(a), ptr(out), batch, size, 0))
    return out


def fft(ctx, x):
    """numpy.fft.fft along the last axis, any length, complex64 (xcorr/signal/fft.py:111-113
    without the fftshift)."""
    return _fft_call('xcx_fft', ctx, x, np.complex64, np.complex64, lambda n: n)


def ifft(ctx, x):
    """numpy.fft.ifft along the last axis (signal/fft.py:239-241 without the fftshift)."""
    return _fft_call('xcx_ifft', ctx, x, np.complex64, np.complex64, lambda n: n)


def rfft(ctx, x):
    """numpy.fft.rfft(x, n) with n = size - (size & 1) (signal/fft.py:357, 367-374)."""
    return _fft_call('xcx_rfft', ctx, x, np.float32, np.complex64, lambda n: (n - (n & 1)) // 2 + 1)


def irfft(ctx, x):
    """numpy.fft.irfft(x) -> 2 (m - 1) real samples (signal/fft.py:475, 487-494)."""
    return _fft_call('xcx_irfft', ctx, x, np.complex64, np.float32, lambda m: 2 * m - 2)


def hilbert(ctx, x, envelope=False):
    """Analytic signal (complex64) of every row, scipy.signal.hilbert semantics
    (xcorr/signal/hilbert.py:69); ``envelope=True`` returns its modulus (float32, :120)."""
    if isinstance(x, DeviceArray):                # device rows in, device rows out (complex: interleaved pairs)
        ctx = _ctx(ctx, x)
        n = x.shape[-1]
        batch = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        out = DeviceArray(ctx, (batch, n if envelope else 2 * n))
        if batch:
            check(_ffi.lib().xcx_hilbert(ctx.handle, ptr(x), ptr(out), batch, n, int(bool(envelope)), 1))
        return out
    ctx = ctx if ctx is not None else default_context()
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim == 0:
        raise ValueError('array with at least one dimension required')
    n = a.shape[-1]
    batch = int(np.prod(a.shape[:-1])) if a.ndim > 1 else 1
    out = _host_out(ctx, a.shape, np.float32 if envelope else np.complex64)
    if batch:
        check(_ffi.lib().xcx_hilbert(ctx.handle, ptr(a), ptr(out), batch, n, int(bool(envelope)), 0))
    return out


def _rows(x):
    if isinstance(x, DeviceArray):
        return x, 1, x.shape[:-1], int(x.shape[-1])
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim == 0:
        raise ValueError('array with at least one dimension required')
    return a, 0, a.shape[:-1], int(a.shape[-1])


def _out(ctx, src, dev, shape):
    return ctx.empty(shape) if dev else _host_out(ctx, shape, np.float32)


def decimate(ctx, x, factor, sos=None):
    """y = lowpass(x)[..., ::factor]; `sos` (scipy layout) is the causal anti-alias
    filter (ObsPy decimate: lowpass_cheby_2), None = no filter."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    batch = int(np.prod(lead)) if lead else 1
    y = _out(ctx, src, dev, tuple(lead) + ((n + factor - 1) // factor,))
    if sos is None:
        s2, K = None, 0
    else:
        s2 = _sos(sos)
        K = s2.shape[0]
    check(_ffi.lib().xcx_decimate(ctx.handle, ptr(src), ptr(y), batch, n, int(factor), ptr(s2), K, dev))
    return y


def resample(ctx, x, num, ratio, window_half=None):
    """Fourier resampling of every row to `num` samples (ObsPy Trace.resample core, xcx_resample):
    irfft(interp(rfft(x) * window_half, bins j*ratio), num) * num / npts."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    batch = int(np.prod(lead)) if lead else 1
    y = _out(ctx, src, dev, tuple(lead) + (int(num),))
    wh = None
    if window_half is not None:
        wh = np.ascontiguousarray(window_half, dtype=np.float32)
        if wh.shape != (n // 2 + 1,):
            raise ValueError('window_half must hold npts//2 + 1 samples')
    check(_ffi.lib().xcx_resample(ctx.handle, ptr(src), ptr(y), batch, n, int(num), float(ratio), ptr(wh), dev))
    return y


def lanczos(ctx, x, n_out, offset, step, a=20):
    """Lanczos (windowed sinc, half-width `a`) interpolation at positions
    offset + i * step (input samples), i < n_out."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    batch = int(np.prod(lead)) if lead else 1
    y = _out(ctx, src, dev, tuple(lead) + (int(n_out),))
    check(_ffi.lib().xcx_lanczos(ctx.handle, ptr(src), ptr(y), batch, n, int(n_out), float(offset), float(step),
                                 int(a), dev))
    return y


def block_rms(ctx, x, win, step=None, nwin=None):
    """rms of blocks x[..., i*step : i*step + win] (xcorr/stream/running_rms.py:104-111)."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    step = int(step or win)
    if nwin is None:
        nwin = (n - int(win)) // step + 1
    batch = int(np.prod(lead)) if lead else 1
    y = _out(ctx, src, dev, tuple(lead) + (int(nwin),))
    check(_ffi.lib().xcx_block_rms(ctx.handle, ptr(src), ptr(y), batch, n, int(win), step, int(nwin), dev))
    return y


def spectrogram(ctx, x, win_len, nfft, window, scale, detrend=True):
    """Hop-1 one-sided PSD of every row (xcorr/signal/spectrogram.py:124-141):
    [..., n] -> [..., nfft//2+1, n] with zero columns at the outer edges.  `window`:
    win_len samples; `scale`: 1/(fs*sum(w**2)) or 1/sum(w)**2."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    batch = int(np.prod(lead)) if lead else 1
    w = np.ascontiguousarray(window, dtype=np.float64)
    if w.shape != (int(win_len),):
        raise ValueError('window must have win_len samples')
    y = _out(ctx, src, dev, tuple(lead) + (int(nfft) // 2 + 1, n))
    check(_ffi.lib().xcx_spectrogram(ctx.handle, ptr(src), ptr(y), batch, n, int(win_len), int(nfft), ptr(w),
                                     float(scale), int(bool(detrend)), dev))
    return y



def spectrogram_seg(ctx, x, win_len, hop, nfft, window, scale, detrend=True):
    """One-sided PSD of segments starting every `hop` samples (scipy.signal.spectrogram as
    called by spectrogram_mtc, xcorr/signal/spectrogram.py:266-276):
    [..., n] -> [..., nfft//2+1, (n-win_len)//hop+1]."""
    ctx = _ctx(ctx, x)
    src, dev, lead, n = _rows(x)
    batch = int(np.prod(lead)) if lead else 1
    w = np.ascontiguousarray(window, dtype=np.float64)
    if w.shape != (int(win_len),):
        raise ValueError('window must have win_len samples')
    nseg = (n - int(win_len)) // int(hop) + 1
    y = _out(ctx, src, dev, tuple(lead) + (int(nfft) // 2 + 1, nseg))
    check(_ffi.lib().xcx_spectrogram_seg(ctx.handle, ptr(src), ptr(y), batch, n, int(win_len), int(hop), int(nfft),
                                         ptr(w), float(scale), int(bool(detrend)), dev))
    return y


def ram_normalize(ctx, x, half_width):
    """Running-absolute-mean normalisation: x / mean(|x|) over a centred window of
    2*half_width+1 samples (truncated at the ends).  Self-defined (Bensen et al. 2007)."""
    ctx = _ctx(ctx, x)
    buf, batch, n, dev, ret = _prep(x, ctx)
    check(_ffi.lib().xcx_ram_normalize(ctx.handle, ptr(buf), batch, n, int(half_width), dev))
    return buf
