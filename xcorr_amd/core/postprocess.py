"""
Post-correlation chain of xcorr (core/postprocess.py:21-150) on the GPU:

    cc.where(lag window).unbias().demean().taper(max_length=5.)
      .timeshift(delay=-(pair_offset+time_offset), dim='lag', fast=True)
      .filter(frequency=3., btype='highpass', order=2).taper(max_length=3/2)

run on all valid (status == 1) pair x time rows at once.  Celerity-based lag
limits (cmin/cmax) need the pair distance, which this build's container does not
carry (util/receiver.py geodesy is out of scope): pass `lag_lim` instead.
"""
import numpy as np

from .. import ops
from ..engine import default_context
from .init import CCDataset
from ..signal.ops1d import butter_design

__all__ = ['postprocess', 'postprocess_rows']


def postprocess_rows(rows, delay, sampling_rate, npts, index_min, index_max, filter_kwargs=None, ctx=None):
    """The chain of core/postprocess.py:133-141 on CCF rows [R, L] of the lag slice
    [index_min, index_max) -- device-resident: a :class:`DeviceArray` in gives a DeviceArray
    [R, L - (L & 1)] out and nothing crosses PCIe (the rows are modified in place up to the
    time shift); a numpy array is uploaded once and the result downloaded once.

        unbias -> demean -> taper(5 s)      one fused op (xcx_post_prep), two passes over the rows
        timeshift(delay[r])                 xcx_timeshift (Dirichlet-kernel convolution)
        sosfiltfilt highpass -> taper(1.5 s)
    """
    import scipy.signal
    from ..engine import DeviceArray
    fk = {'frequency': 3., 'btype': 'highpass', 'order': 2, **(filter_kwargs or {})}
    fs = float(sampling_rate)
    ctx = ctx or (rows.ctx if isinstance(rows, DeviceArray) else default_context())
    host = not isinstance(rows, DeviceArray)
    d = ctx.array(np.ascontiguousarray(rows, dtype=np.float32)) if host else rows
    if len(d.shape) != 2 or d.shape[1] != index_max - index_min:
        raise ValueError('rows must be [R, index_max - index_min]')
    ops.post_prep(ctx, d, npts, index_min, index_max, unbiased=True,
                  taper_nmax=int(round(5. * fs)))                   # taper: timeshift phase wrapping
    y = ops.timeshift(ctx, d, np.asarray(delay, np.float64).reshape(-1), fs)
    sos = butter_design(fk['order'], fk['frequency'], fk['btype'], fs)
    ops.sosfiltfilt(ctx, y, sos)
    ops.taper(ctx, y, int(round(1.5 * fs)))                         # filter artefacts
    if host:
        out = y.to_host()
        y.free()
        d.free()
        return out
    return y


def postprocess(ds, lag_lim=None, lag_min=None, lag_max=None, filter_kwargs=None, ctx=None):
    """Returns a new :class:`CCDataset` holding the valid rows, postprocessed.

    Mirrors core/postprocess.py:81-148: rows with status != 1 are dropped
    (their cc is zero-filled here, status kept), the lag axis is cut to
    [lag_min, lag_max], and -- like the reference's rfft-based timeshift -- an
    odd number of lags loses its last sample."""
    from .init import is_xarray_dataset
    if is_xarray_dataset(ds):            # the reference's argument type (core/postprocess.py:21): convert in, convert out
        return postprocess(CCDataset.from_xarray(ds), lag_lim=lag_lim, lag_min=lag_min, lag_max=lag_max,
                           filter_kwargs=filter_kwargs, ctx=ctx).to_xarray()
    if 'postprocess' in ds.cc_attrs:
        return ds
    if lag_lim is not None:
        lag_min, lag_max = lag_lim
    fk = {'frequency': 3., 'btype': 'highpass', 'order': 2, **(filter_kwargs or {})}
    lag = ds.lag
    lag_min = lag.min() if lag_min is None else lag_min
    lag_max = lag.max() if lag_max is None else lag_max
    m = (lag >= lag_min) & (lag <= lag_max)
    if not m.any():
        raise ValueError("No data after extracting time lag")
    sel = np.flatnonzero(m)
    i0, i1 = int(sel[0]), int(sel[-1]) + 1
    la = ds.lag_attrs
    fs, npts = la['sampling_rate'], la['npts']
    imin = la['index_min'] + i0
    imax = la['index_min'] + i1
    valid = ds.status == 1
    if not valid.any():
        raise ValueError("No data after extracting valid times")
    # one gather of the valid rows' lag slice (boolean indexing of the whole rows followed by the slice copied the
    # block twice; a container without gaps needs no gather at all)
    cc2 = ds.cc.reshape(-1, ds.cc.shape[-1])
    vflat = valid.reshape(-1)
    if vflat.all():
        rows = cc2[:, i0:i1]
    else:
        rows = cc2[np.flatnonzero(vflat), i0:i1]
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    delay = -(ds.pair_offset + ds.time_offset)[valid]
    ctx = ctx or default_context()
    y = postprocess_rows(rows, delay, fs, npts, imin, imax, filter_kwargs=fk, ctx=ctx)   # one H2D, one D2H
    n = y.shape[-1]
    out = CCDataset(ds.pair, ds.time, lag[i0:i0 + n],
                    {**la, 'index_min': imin, 'index_max': imin + n},
                    {**ds.cc_attrs, 'unbiased': np.byte(1), 'postprocess': 'xcorr_amd'},
                    ds.time_attrs, dtype=ds.cc.dtype)
    out.cc[valid] = y.astype(ds.cc.dtype, copy=False)
    out.status[...] = ds.status
    out.pair_offset[...] = ds.pair_offset
    out.time_offset[...] = ds.time_offset
    out.attrs = dict(ds.attrs)
    return out
