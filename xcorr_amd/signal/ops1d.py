"""numpy-level + DataArray-level implementations (see package docstring)."""
import functools

import numpy as np

from .. import ops as _ops
from ..engine import CCPlan, default_context

try:
    import xarray as xr
except ModuleNotFoundError:
    xr = None

_types = ('constant', 'linear')
_plans = {}


def _is_da(x):
    return xr is not None and isinstance(x, xr.DataArray)


def _check_dim(x, dim):
    dim = dim or x.dims[-1]
    if not isinstance(dim, str):
        raise TypeError('dim should be a string')
    if dim not in x.dims:
        raise ValueError(f'x has no dimensions "{dim}"')
    return dim


def _apply_da(x, dim, fn):
    """Move `dim` last, run fn on the float32 values, rebuild the DataArray."""
    xt = x.transpose(..., dim)
    vals = np.ascontiguousarray(xt.values, dtype=np.float32)
    out = fn(vals).astype(x.dtype)
    y = xt.copy(data=out).transpose(*x.dims)
    y.attrs = x.attrs
    y.name = x.name
    return y


def _cc_plan(n, normalize):
    key = (n, bool(normalize))
    if key not in _plans:
        if len(_plans) > 16:
            _plans.pop(next(iter(_plans))).close()
        _plans[key] = CCPlan(default_context(), n, normalize=normalize)
    return _plans[key]


# -- a4: signal/correlate.py:32-159 -----------------------------------------
def correlate1d(in1, in2, normalize=True, dtype=None, dim=None, **kwargs):
    """One-dimensional cross-correlation along the last axis (or `dim`).
    Output length 2N-1, lags -(N-1)..(N-1) (signal/correlate.py:97-113)."""
    if _is_da(in1):
        dim = _check_dim(in1, dim)
        if dim not in in2.dims:
            raise ValueError(f'in2 has no dimensions "{dim}"')
        if in1.shape != in2.shape:
            raise ValueError('in1 and in2 should have the same shape!')
        dtype = dtype or in1.dtype
        if not isinstance(dtype, np.dtype):
            raise TypeError('dtype should be a numpy.dtype')
        a = in1.transpose(..., dim)
        b = in2.transpose(..., dim)
        out = correlate1d(a.values, b.values, normalize=normalize,
                          dtype=dtype)
        coord = in1[dim]
        N = coord.size
        delta = float((coord[1] - coord[0]).item()) if N > 1 else 1.0
        new_dim = f'delta_{dim}'
        cc = xr.DataArray(
            out, dims=a.dims[:-1] + (new_dim,),
            coords={**{d: a[d] for d in a.dims[:-1] if d in a.coords},
                    new_dim: np.arange(-(N - 1), N) * delta},
            name='cc')
        cc.attrs = {
            'long_name': 'Cross-correlation Estimate',
            'standard_name': 'cross_correlation_estimate',
            'units': '-',
            'normalize': np.byte(normalize),
            'bias_correct': np.byte(0),
            'unbiased': np.byte(0),
        }
        return cc
    in1 = np.asarray(in1)
    in2 = np.asarray(in2)
    if in1.shape != in2.shape:
        raise ValueError('in1 and in2 should have the same shape!')
    dtype = np.dtype(dtype or in1.dtype)
    n = in1.shape[-1]
    lead = in1.shape[:-1]
    plan = _cc_plan(n, normalize)
    out = plan.cc(np.ascontiguousarray(in1, np.float32).reshape(-1, n),
                  np.ascontiguousarray(in2, np.float32).reshape(-1, n))
    return out.reshape(lead + (2 * n - 1,)).astype(dtype)


# -- a5: signal/normalize.py:33-92 -------------------------------------------
def norm1d(x, dim=None, **kwargs):
    """x / ||x||_2 along the last axis (signal/normalize.py:73-80)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        return _apply_da(x, dim, lambda v: _ops.norm1d(None, v))
    return _ops.norm1d(None, x)


norm = norm1d


# -- a7: signal/detrend.py:31-145 ----------------------------------------------
def detrend(x, dim=None, type=None, skipna=True):
    """Remove the mean ('constant') or the least-squares line ('linear')
    (signal/detrend.py:87-110).  ``skipna=True`` (the reference's default) fits on the finite
    samples only and leaves NaN samples NaN; ``skipna=False`` lets a NaN poison its row, like
    ``x.mean(skipna=False)`` / ``linregress`` on data with NaN."""
    type = type or _types[0]
    if not isinstance(type, str):
        raise TypeError('type should be a string')
    if type not in _types:
        raise KeyError(f'type should be any of "{"|".join(_types)}"')
    if skipna:
        def fn(c, v):
            return _ops.detrend_nan(c, v, linear=(type == 'linear'))
    else:
        fn = _ops.detrend_linear if type == 'linear' else _ops.demean
    if _is_da(x):
        dim = _check_dim(x, dim)
        return _apply_da(x, dim, lambda v: fn(None, v))
    return fn(None, x)


def demean(x, dim=None, **kwargs):
    """signal/detrend.py:127-145."""
    return detrend(x, dim=dim, type='constant', **kwargs)


# -- a8: signal/window.py:27-116, signal/taper.py:23-85 -------------------------
def _nmax(npts, sampling_rate, max_percentage, max_length, xmax=None):
    nmax = 0.5 * npts
    if max_percentage:
        if not isinstance(max_percentage, float):
            raise TypeError('max_percentage should be of type `float`.')
        if max_percentage < 0. or max_percentage > .5:
            raise ValueError('max_percentage should be within [0, .5].')
        nmax = max_percentage * npts
    if max_length:
        if not isinstance(max_length, float):
            raise TypeError('max_length should be of type `float`.')
        if max_length < 0. or (xmax is not None and max_length > xmax):
            raise ValueError('max_length should be within [0, x].')
        nmax = max_length * sampling_rate
    return int(round(nmax))


def window(npts, sampling_rate, wtype='hann', max_percentage=None,
           max_length=None, side=None):
    """Taper window of `npts` samples (signal/window.py:74-106): ones with the
    first/last ``nmax`` samples of a periodic Hann window of length 2*nmax.
    Host-side coordinate helper (float64), like the reference."""
    wtype = wtype or 'hann'
    side = side or 'both'
    if side not in ('left', 'right', 'both'):
        raise ValueError('side should be either "left", "right" or "both"!')
    nmax = _nmax(npts, sampling_rate, max_percentage, max_length)
    win = _flanks(wtype, nmax)
    w = np.ones(npts, dtype=np.float64)
    w[0:nmax] = win[0:nmax]
    w[npts - nmax:] = win[nmax:]
    return w


def _flanks(wtype, nmax):
    """``scipy.signal.windows.<wtype>(2*nmax, False)`` (signal/window.py:92), float64."""
    import scipy.signal.windows as _w
    if not hasattr(_w, wtype):
        raise ValueError(f'unknown window type "{wtype}"')
    return np.asarray(getattr(_w, wtype)(2 * nmax, False), dtype=np.float64)


def taper(x, wtype=None, max_percentage=None, max_length=None, side=None,
          dim=None, sampling_rate=None):
    """x * window (signal/taper.py:67-69); any ``scipy.signal.windows`` type (Hann flanks are
    generated on the device, other types come from a host table)."""
    wtype = wtype or 'hann'

    def run(v, nmax):
        if wtype == 'hann':
            return _ops.taper(None, v, nmax)
        return _ops.taper_table(None, v, _flanks(wtype, nmax).astype(np.float32))
    if _is_da(x):
        dim = _check_dim(x, dim)
        if 'sampling_rate' not in x[dim].attrs:
            raise ValueError('x has no attribute "sampling_rate"!')
        fs = x[dim].attrs['sampling_rate']
        nmax = _nmax(x[dim].size, fs, max_percentage, max_length,
                     float(np.max(x[dim].values)))
        return _apply_da(x, dim, lambda v: run(v, nmax))
    if sampling_rate is None:
        raise ValueError('x has no attribute "sampling_rate"!')
    x = np.asarray(x)
    nmax = _nmax(x.shape[-1], sampling_rate, max_percentage, max_length)
    return run(x, nmax)


# -- a9: signal/filter.py:27-120 -------------------------------------------------
def butter_design(order, frequency, btype, sampling_rate):
    """``scipy.signal.butter(N, Wn, btype, output='sos', fs)`` (signal/filter.py:86-92), cached per argument set: a design
    costs 0.3 ms of host time, every call of the post-correlation chain asked for the same one."""
    wn = tuple(float(f) for f in frequency) if isinstance(frequency, (tuple, list, np.ndarray)) else float(frequency)
    return _butter_design(int(order), wn, str(btype), float(sampling_rate)).copy()


@functools.lru_cache(maxsize=64)
def _butter_design(order, wn, btype, fs):
    import scipy.signal
    return scipy.signal.butter(N=order, Wn=list(wn) if isinstance(wn, tuple) else wn, btype=btype, output='sos', fs=fs)


def filter(x, frequency, btype=None, order=2, dim=None, sampling_rate=None):
    """Zero-phase Butterworth: ``butter(order, frequency, btype, 'sos', fs)``
    designed on the host (scipy, microseconds), ``sosfiltfilt`` on the GPU
    (signal/filter.py:86-96)."""
    import scipy.signal
    if not (isinstance(frequency, float) or
            (isinstance(frequency, tuple) and len(frequency) == 2)):
        raise ValueError('Corner frequency should be a `float` or tuple-pair '
                         'with (min, max)!')
    btype = btype or 'highpass'
    if _is_da(x):
        dim = _check_dim(x, dim)
        if 'sampling_rate' not in x[dim].attrs:
            raise ValueError(f'Dimension "{dim}" has no attribute '
                             '"sampling_rate"!')
        sampling_rate = x[dim].attrs['sampling_rate']
    elif sampling_rate is None:
        raise ValueError('sampling_rate is required for array input')
    sos = butter_design(order, frequency, btype, sampling_rate)
    if _is_da(x):
        return _apply_da(x, dim, lambda v: _ops.sosfiltfilt(None, v, sos))
    return _ops.sosfiltfilt(None, x, sos)


# -- a3: signal/unbias.py:30-145 -------------------------------------------------
def unbias_weights(npts, index_min=0, index_max=None):
    """``weight(npts)[index_min:index_max]`` (signal/unbias.py:133-135)."""
    from ..util.cc import weight
    index_max = 2 * npts - 1 if index_max is None else index_max
    return weight(npts, pad=True)[index_min:index_max]


def unbias(x, npts=None, index_min=0, index_max=None, dim=None):
    """Bias-correct a cross-correlation estimate (signal/unbias.py:73-85)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        if 'unbiased' not in x.attrs:
            raise ValueError('x has no unbiased attribute flag!')
        if x.attrs['unbiased'] != 0:
            return x
        la = x[dim].attrs
        for attr in ('sampling_rate', 'delta', 'npts', 'index_min',
                     'index_max'):
            if attr not in la:
                raise ValueError(f'Lag has no attribute "{attr}"!')
        y = _apply_da(x, dim, lambda v: _ops.unbias(
            None, v, la['npts'], la['index_min'], la['index_max']))
        y.attrs['unbiased'] = np.byte(True)
        return y
    if npts is None:
        raise ValueError('npts is required for array input')
    return _ops.unbias(None, x, npts, index_min, index_max or 0)


# -- a10: signal/timeshift.py:23-117 ---------------------------------------------
def timeshift(x, delay, dtype=None, dim=None, fast=True, sampling_rate=None):
    """Shift along the last axis (or `dim`) by `delay` seconds through the
    phase ramp exp(-2 pi i delay f) (signal/timeshift.py:90-96).  `delay` is a
    float or an array broadcasting over the leading dimensions.  Like the
    reference (rfft path, signal/fft.py:357) an odd-length axis loses its last
    sample with ``fast=True``; ``fast=False`` (fft/ifft) keeps every sample and,
    for an odd length, reproduces the reference's cos(2 pi m / n) factor (its
    `ifft` applies fftshift where ifftshift is due, signal/fft.py:239-241)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        if not np.all(np.abs(x[dim].diff(dim, 2)) < 1e-10):
            raise ValueError(f'coordinate "{dim}" should be regularly spaced')
        if _is_da(delay):
            if dim in delay.dims:
                raise ValueError(f'delay cannot depend on dim "{dim}"')
            for d in delay.dims:
                if d not in x.dims:
                    raise ValueError(f'delay dim "{d}" not existing in x')
        elif not isinstance(delay, float):
            raise TypeError('delay should be a float or DataArray')
        fs = x[dim].attrs.get('sampling_rate')
        if fs is None:
            c = x[dim].values
            fs = 1.0 / float(c[1] - c[0])
        xt = x.transpose(..., dim)
        dl = delay
        if _is_da(delay):
            dl = delay.broadcast_like(xt.isel({dim: 0})).transpose(*xt.dims[:-1]).values
        if not fast:        # the fft path takes delta from the coordinate itself (fft.py:104-107)
            c = x[dim].values
            fs = 1.0 / float(np.fabs(c[1] - c[0]))
        out = _ops.timeshift(None, np.ascontiguousarray(xt.values, np.float32), dl, fs, fast=fast)
        y = xt.isel({dim: slice(0, out.shape[-1])}).copy(data=out.astype(x.dtype))
        y = y.transpose(*x.dims)
        y.attrs = x.attrs
        y.name = x.name
        return y
    if sampling_rate is None:
        raise ValueError('sampling_rate is required for array input')
    if not isinstance(delay, (float, np.ndarray, np.floating)):
        raise TypeError('delay should be a float or an array')
    x = np.asarray(x)
    out = _ops.timeshift(None, x, delay, sampling_rate, fast=fast)
    return out.astype(dtype or x.dtype)


# -- signal/rms.py:23-68, signal/absolute.py --------------------------------------
def rms(x, dim=None):
    """sqrt(mean(x**2)) along the last axis / `dim` (signal/rms.py:52-58)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        xt = x.transpose(..., dim)
        vals = rms(xt.values)
        y = xr.DataArray(vals.astype(x.dtype), dims=xt.dims[:-1],
                         coords={d: xt[d] for d in xt.dims[:-1] if d in xt.coords}, name=x.name, attrs=x.attrs)
        return y
    x = np.asarray(x)
    n = x.shape[-1]
    out = _ops.block_rms(None, np.ascontiguousarray(x, np.float32), n, n, 1)
    return out[..., 0].astype(x.dtype if x.dtype.kind == 'f' else np.float32)


def absolute(x):
    """|x| (signal/absolute.py); elementwise, stays on the host array library."""
    return abs(x)


# -- signal/hilbert.py:29-120 ---------------------------------------------------
def hilbert(x, dim=None, **kwargs):
    """Analytic signal along the last axis / `dim` (scipy.signal.hilbert, hilbert.py:69)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        xt = x.transpose(..., dim)
        y = xt.copy(data=_ops.hilbert(None, xt.values).astype('complex128' if x.dtype == np.float64 else 'complex64'))
        y = y.transpose(*x.dims)
        y.attrs, y.name = x.attrs, x.name
        return y
    return _ops.hilbert(None, np.asarray(x))


def envelope(x, dim=None, **kwargs):
    """|hilbert(x)| (signal/hilbert.py:91-120)."""
    if _is_da(x):
        dim = _check_dim(x, dim)
        return _apply_da(x, dim, lambda v: _ops.hilbert(None, v, envelope=True))
    x = np.asarray(x)
    return _ops.hilbert(None, x, envelope=True).astype(x.dtype if x.dtype.kind == 'f' else np.float32)


# -- signal/snr.py:27-168, signal/rms.py:23-68 ------------------------------------
def snr(x, signal, noise, dim=None, power=False, decibels=False, extend=False, envelope=False):
    """Signal-to-noise ratio along the last axis: max|x| (or of the Hilbert envelope,
    ``envelope=True``) in the `signal` mask over the rms of x in the `noise` mask
    (signal/snr.py:107-116).  ``extend=True`` returns (snr, s, n, argmax)."""
    if _is_da(x):
        # signal/snr.py:85-165: masks along `dim`; returns the snr DataArray, or with extend=True a
        # Dataset {snr, s, n, <dim>_s_max} like the reference
        dim = _check_dim(x, dim)
        xt = x.transpose(..., dim)
        sm = np.asarray(signal.values if _is_da(signal) else signal, dtype=bool)
        nm = np.asarray(noise.values if _is_da(noise) else noise, dtype=bool)
        r, s_, n_, a_ = snr(np.ascontiguousarray(xt.values, dtype=np.float32), sm, nm, power=power,
                            decibels=decibels, extend=True, envelope=envelope)
        lead = {d: xt[d] for d in xt.dims[:-1] if d in xt.coords}
        da = xr.DataArray(np.asarray(r).reshape(xt.shape[:-1]), dims=xt.dims[:-1], coords=lead, name='snr')
        if not extend:
            return da
        ds = xr.Dataset()
        ds['snr'] = da
        ds['s'] = xr.DataArray(np.asarray(s_).reshape(xt.shape[:-1]), dims=xt.dims[:-1], coords=lead)
        ds['n'] = xr.DataArray(np.asarray(n_).reshape(xt.shape[:-1]), dims=xt.dims[:-1], coords=lead)
        ds[f'{dim}_s_max'] = xr.DataArray(xt[dim].values[np.asarray(a_).reshape(xt.shape[:-1])],
                                          dims=xt.dims[:-1], coords=lead)
        return ds
    x = np.asarray(x)
    s, n, a = _ops.snr(None, x, np.asarray(signal), np.asarray(noise))
    if envelope:
        s, _, a = _ops.snr(None, _ops.hilbert(None, x, envelope=True), np.asarray(signal), np.asarray(noise))
    r = s / n
    if decibels:
        r = 20 * np.log10(r)
    elif power:
        r = np.square(r)
    return (r, s, n, a) if extend else r
