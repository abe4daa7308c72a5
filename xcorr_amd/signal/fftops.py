"""Labelled FFTs of ``xcorr/signal/fft.py`` (a6): numpy-level kernels on the GPU
(any length, complex64 / float32) plus the reference's coordinate rules."""
import json

import numpy as np

from .. import ops as _ops
from .ops1d import _is_da, _check_dim, xr

_recip_name = 'reciprocal_name'
_recip_attr = 'reciprocal_attrs'
_recip_zero = 'reciprocal_zero'


def fft_freq(size, delta):
    """`fft` frequency axis (signal/fft.py:104-108)."""
    return np.fft.fftshift(np.fft.fftfreq(size, d=np.round(abs(delta), decimals=10)))


def rfft_freq(coord):
    """`rfft` frequency axis (signal/fft.py:357-359)."""
    coord = np.asarray(coord, dtype=np.float64)
    n = coord.size - 1 if coord.size & 1 else coord.size
    fN = np.round(n // 2 / (coord[n - 1] - coord[0]), decimals=0)
    return np.linspace(0., fN, n // 2 + 1)


def irfft_time(freq, zero=0.0):
    """`irfft` time axis (signal/fft.py:475-479)."""
    n = 2 * len(freq) - 2
    return np.linspace(0., (n - 1) / 2 / freq[-1], n) + zero


def _complex_dtype(dtype, default):
    dtype = np.dtype(dtype or default)
    if 'complex' not in dtype.name:
        raise TypeError('dtype should be complex.')
    return dtype


def _float_dtype(dtype, default):
    dtype = np.dtype(dtype or default)
    if 'float' not in dtype.name:
        raise TypeError('dtype should be float.')
    return dtype


def _dump_recip(c):
    attrs = {a: c.attrs[a] for a in ('long_name', 'standard_name', 'units') if a in c.attrs}
    return {_recip_name: c.name, _recip_attr: json.dumps(attrs), _recip_zero: c[0].item()}


def _relabel(x, dim, new_dim, values, coord, coord_attrs):
    xt = x.transpose(..., dim)
    y = xr.DataArray(values, dims=xt.dims[:-1] + (new_dim,),
                     coords={d: xt[d] for d in xt.dims[:-1] if d in xt.coords}, name=x.name, attrs=x.attrs)
    return y.assign_coords({new_dim: xr.DataArray(coord, dims=(new_dim,), name=new_dim, attrs=coord_attrs)})


def fft(x, dim=None, new_dim=None, dtype=None, **kwargs):
    """fftshift(fft(x)) along the last axis / `dim` (signal/fft.py:37-156)."""
    dtype = _complex_dtype(dtype, 'complex64' if not _is_da(x) else 'complex128')
    if _is_da(x):
        dim = _check_dim(x, dim)
        new_dim = new_dim or x[dim].attrs.get(_recip_name) or 'freq'
        if new_dim in x.dims:
            raise ValueError(f'x already has a dimension "{new_dim}"')
        delta = abs((x[dim][1] - x[dim][0]).values.item())
        vals = fft(x.transpose(..., dim).values, dtype=dtype)
        attrs = {**x[dim].attrs, 'long_name': 'Frequency', 'standard_name': 'frequency', 'units': 's-1',
                 **_dump_recip(x[dim])}
        return _relabel(x, dim, new_dim, vals, fft_freq(x[dim].size, delta), attrs)
    F = _ops.fft(None, np.asarray(x))
    return np.fft.fftshift(F, axes=-1).astype(dtype)


def ifft(x, dim=None, new_dim=None, dtype=None, **kwargs):
    """ifft(fftshift(x)) along the last axis / `dim` (signal/fft.py:159-284)."""
    dtype = _complex_dtype(dtype, 'complex64' if not _is_da(x) else 'complex128')
    if _is_da(x):
        dim = _check_dim(x, dim)
        new_dim = new_dim or x[dim].attrs.get(_recip_name) or 'time'
        if new_dim in x.dims:
            raise ValueError(f'x already has a dimension "{new_dim}"')
        delta = np.round(abs((x[dim][1] - x[dim][0]).values.item()), decimals=10)
        size = x[dim].size
        time = np.linspace(0., (size - 1) / np.round(size * delta, decimals=6), size)
        time = time + x[dim].attrs.get(_recip_zero, 0.)
        vals = ifft(x.transpose(..., dim).values, dtype=dtype)
        attrs = {**x[dim].attrs, 'long_name': 'Time', 'standard_name': 'time', 'units': 's',
                 **_dump_recip(x[dim])}
        return _relabel(x, dim, new_dim, vals, time, attrs)
    F = np.fft.fftshift(np.asarray(x), axes=-1)
    return _ops.ifft(None, F).astype(dtype)


def rfft(x, dim=None, new_dim=None, dtype=None, **kwargs):
    """rfft over n = size - (size & 1) samples (signal/fft.py:287-402)."""
    dtype = _float_dtype(dtype, 'float32' if not _is_da(x) else 'float64')
    cdtype = np.dtype(f'complex{dtype.itemsize * 16}')
    if _is_da(x):
        dim = _check_dim(x, dim)
        new_dim = new_dim or x[dim].attrs.get(_recip_name) or 'freq'
        if new_dim in x.dims:
            raise ValueError(f'x already has a dimension "{new_dim}"')
        vals = rfft(x.transpose(..., dim).values, dtype=dtype)
        attrs = {**x[dim].attrs, 'long_name': 'Frequency', 'standard_name': 'frequency', 'units': 's-1',
                 **_dump_recip(x[dim])}
        return _relabel(x, dim, new_dim, vals, rfft_freq(x[dim].values), attrs)
    return _ops.rfft(None, np.asarray(x)).astype(cdtype)


def irfft(x, dim=None, new_dim=None, dtype=None, **kwargs):
    """irfft to 2 (size - 1) samples (signal/fft.py:405-522)."""
    dtype = _float_dtype(dtype, 'float32' if not _is_da(x) else 'float64')
    if _is_da(x):
        dim = _check_dim(x, dim)
        new_dim = new_dim or x[dim].attrs.get(_recip_name) or 'time'
        if new_dim in x.dims:
            raise ValueError(f'x already has a dimension "{new_dim}"')
        vals = irfft(x.transpose(..., dim).values, dtype=dtype)
        time = irfft_time(x[dim].values, x[dim].attrs.get(_recip_zero, 0.))
        attrs = {**x[dim].attrs, 'long_name': 'Time', 'standard_name': 'time', 'units': 's',
                 **_dump_recip(x[dim])}
        return _relabel(x, dim, new_dim, vals, time, attrs)
    return _ops.irfft(None, np.asarray(x)).astype(dtype)
