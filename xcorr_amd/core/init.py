"""
Output container of the cross-correlation driver.

Mirrors the variables/attrs that ``xcorr.core.init`` (core/init.py:25-327)
lays out in its xarray.Dataset -- ``cc`` float32 (pair, time, lag), ``status``
int8, ``pair_offset``/``time_offset`` float64, lag attrs ``sampling_rate,
delta, npts, index_min, index_max`` -- as plain numpy arrays (xarray is an
optional dependency here; ``to_xarray()`` converts when it is importable).
"""
import numpy as np

from ..util import cc as ucc

__all__ = ['init', 'CCDataset']


class CCDataset:
    """Plain-array stand-in for the reference's ``xarray.Dataset``."""

    def __init__(self, pair, time, lag, lag_attrs, cc_attrs, time_attrs,
                 dtype=np.float32):
        self.pair = np.asarray(pair, dtype=object)
        self.time = np.asarray(time)            # window CENTRES (core/init.py:163)
        self.lag = np.asarray(lag, dtype=np.float64)
        self.lag_attrs = dict(lag_attrs)
        self.time_attrs = dict(time_attrs)
        shape = (len(self.pair), len(self.time), len(self.lag))
        self.cc = np.zeros(shape, dtype=dtype)
        self.cc_attrs = dict(cc_attrs)
        self.status = np.zeros(shape[:2], dtype=np.byte)       # -1 / 0 / 1
        self.pair_offset = np.zeros(shape[:2], dtype=np.float64)
        self.time_offset = np.zeros(shape[:2], dtype=np.float64)
        self.attrs = {'history': ''}

    def to_xarray(self):
        import xarray as xr
        ds = xr.Dataset()
        ds.coords['pair'] = self.pair
        ds.coords['time'] = self.time
        ds.coords['lag'] = ('lag', self.lag, self.lag_attrs)
        ds['cc'] = (('pair', 'time', 'lag'), self.cc, self.cc_attrs)
        ds['status'] = (('pair', 'time'), self.status)
        ds['pair_offset'] = (('pair', 'time'), self.pair_offset)
        ds['time_offset'] = (('pair', 'time'), self.time_offset)
        ds.attrs = dict(self.attrs)
        return ds


def init(pair, starttime, endtime, sampling_rate, window_length=86400.,
         window_overlap=.875, clip_lag=None, unbiased_cc=False, closed='left',
         dtype=np.float32, normalize=True):
    """Build the container (core/init.py:25-327, numerics-relevant part).

    pair : str or list of str (the reference holds exactly one pair per
        dataset, core/init.py:154; a list is allowed here so that one
        container feeds the batched engine).
    starttime, endtime : seconds (float) relative to any epoch; window centres
        are ``arange(start, end, window_length*(1-window_overlap))`` like
        ``pd.date_range(freq=...)`` at core/init.py:163-168 (`closed` =
        'left' drops the end point, 'right' the start, None keeps both).
    clip_lag : None or (a, b) seconds -> ``argmin|lag-a| : argmin|lag-b|``
        (core/init.py:176-196, end exclusive).
    """
    pairs = [pair] if isinstance(pair, str) else list(pair)
    delta = 1. / sampling_rate
    npts = int(window_length * sampling_rate)              # core/init.py:120
    step = window_length * (1 - window_overlap)
    nsteps = int(np.floor((endtime - starttime) / step + 1e-9))
    time = starttime + step * np.arange(nsteps + 1)
    if closed == 'left':
        time = time[time < endtime - 1e-9]
    elif closed == 'right':
        time = time[time > starttime + 1e-9]
    lag = ucc.lag(npts, delta, pad=True)
    if clip_lag is not None:
        msg = ('``clip_lag`` should be in seconds of type `float` or of type'
               '`tuple` with length 2 specifying start and end.')
        if isinstance(clip_lag, float):
            clip_lag = (-abs(clip_lag), abs(clip_lag))
        elif not (isinstance(clip_lag, tuple) and len(clip_lag) == 2):
            raise TypeError(msg)
        nmin = int(np.argmin(abs(lag - clip_lag[0])))
        nmax = int(np.argmin(abs(lag - clip_lag[1])))
    else:
        nmin, nmax = 0, 2 * npts - 1
    lag_attrs = {
        'long_name': 'Time lag', 'standard_name': 'time_lag', 'units': 's',
        'sampling_rate': float(sampling_rate), 'delta': float(delta),
        'npts': int(npts),
        'clip_lag': np.array(clip_lag if clip_lag is not None else []),
        'index_min': int(nmin), 'index_max': int(nmax),
    }
    cc_attrs = {
        'long_name': 'Cross-correlation Estimate',
        'standard_name': 'cross_correlation_estimate', 'units': '-',
        'normalize': np.byte(normalize), 'bias_correct': np.byte(unbiased_cc),
        'unbiased': np.byte(0),
    }
    time_attrs = {'window_length': window_length,
                  'window_overlap': window_overlap, 'closed': closed}
    return CCDataset(pairs, time, lag[nmin:nmax], lag_attrs, cc_attrs,
                     time_attrs, dtype=dtype)
