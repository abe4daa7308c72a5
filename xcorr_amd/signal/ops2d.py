"""Two-dimensional members of ``xcorr/signal``: norm2d and correlate2d."""
import numpy as np

from .. import ops as _ops
from .ops1d import _is_da, xr


def _check_dims(x, dims):
    dims = dims or x.dims[-2:]
    if not isinstance(dims, tuple) or len(dims) != 2:
        raise TypeError('dims should be a tuple of length 2')
    return dims


# -- a5: signal/normalize.py:95-155 -------------------------------------------
def norm2d(x, dims=None, **kwargs):
    """x / ||x||_F over the last two axes (or `dims`)."""
    if _is_da(x):
        dims = _check_dims(x, dims)
        for d in dims:
            if d not in x.dims:
                raise ValueError(f'x has no dimensions "{dims}"')
        xt = x.transpose(..., *dims)
        y = xt.copy(data=norm2d(xt.values).astype(x.dtype)).transpose(*x.dims)
        y.attrs, y.name = x.attrs, x.name
        return y
    x = np.asarray(x)
    if x.ndim < 2:
        raise ValueError('x should have at least two dimensions')
    flat = np.ascontiguousarray(x, np.float32).reshape(x.shape[:-2] + (-1,))
    return _ops.norm1d(None, flat).reshape(x.shape).astype(x.dtype)


def _tiles(in1, in2):
    in1 = np.asarray(in1)
    in2 = np.asarray(in2)
    if in1.shape != in2.shape:
        raise ValueError('in1 and in2 should have the same shape!')
    if in1.ndim < 2:
        raise ValueError('in1 and in2 should have at least two dimensions')
    F, L = in1.shape[-2:]
    a = np.ascontiguousarray(in1, np.float32).reshape(-1, F, L)
    b = np.ascontiguousarray(in2, np.float32).reshape(-1, F, L)
    K = a.shape[0]
    pairs = np.stack([np.arange(K), K + np.arange(K)], axis=1)
    return np.concatenate([a, b]), pairs, in1.shape[:-2], F, L


# -- a14: signal/correlate.py:162-291 -----------------------------------------
def correlate2d(in1, in2, normalize=True, dtype=None, dims=None, **kwargs):
    """Two-dimensional cross-correlation over the last two axes (or `dims`).
    Output [..., 2F-1, 2L-1], lags -(F-1)..(F-1) x -(L-1)..(L-1)."""
    if _is_da(in1):
        dims = _check_dims(in1, dims)
        for d in dims:
            if d not in in1.dims:
                raise ValueError(f'in1 has no dimensions "{dims}"')
            if d not in in2.dims:
                raise ValueError(f'in2 has no dimensions "{dims}"')
        if in1.shape != in2.shape:
            raise ValueError('in1 and in2 should have the same shape!')
        dtype = dtype or in1.dtype
        if not isinstance(dtype, np.dtype):
            raise TypeError('dtype should be a numpy.dtype')
        a = in1.transpose(..., *dims)
        b = in2.transpose(..., *dims)
        out = correlate2d(a.values, b.values, normalize=normalize, dtype=dtype)
        new_dims = (f'delta_{dims[0]}', f'delta_{dims[1]}')
        coords = {d: a[d] for d in a.dims[:-2] if d in a.coords}
        for d, nd in zip(dims, new_dims):
            c = in1[d]
            n = c.size
            delta = (c[1] - c[0]).item() if n > 1 else 1.0
            coords[nd] = np.arange(-(n - 1), n) * delta
        cc = xr.DataArray(out, dims=a.dims[:-2] + new_dims, coords=coords, name='cc2')
        cc.attrs = {
            'long_name': 'Two-dimensional Cross-correlation Estimate',
            'standard_name': '2d_cross_correlation_estimate',
            'units': '-',
            'normalize': np.byte(normalize),
            'bias_correct': np.byte(0),
            'unbiased': np.byte(0),
        }
        return cc
    dtype = np.dtype(dtype or np.asarray(in1).dtype)
    tiles, pairs, lead, F, L = _tiles(in1, in2)
    _, _, full = _ops.correlate2d(None, tiles, pairs, normalize=normalize, full=True)
    return full.reshape(lead + (2 * F - 1, 2 * L - 1)).astype(dtype)


def correlate2d_peak(in1, in2, normalize=True):
    """max and argmax of correlate2d without materialising it -- what
    scripts/timelapse.py:255-267 keeps.  Returns (cc2_max[...], index[..., 2]);
    index - (F-1, L-1) is the lag in samples."""
    tiles, pairs, lead, F, L = _tiles(np.asarray(in1), np.asarray(in2))
    peak, idx = _ops.correlate2d(None, tiles, pairs, normalize=normalize)
    return peak.reshape(lead), idx.reshape(lead + (2,))
