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

__all__ = ['init', 'CCDataset', 'is_xarray_dataset']


class CCDataset:
    """Plain-array stand-in for the reference's ``xarray.Dataset``."""

    def __init__(self, pair, time, lag, lag_attrs, cc_attrs, time_attrs,
                 dtype=np.float32, hash_waveforms=False):
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
        # 'n/a' until processed (core/init.py:252-255); None when hashing was not requested
        self.hash = np.full(shape[:2], 'n/a', dtype=object) if hash_waveforms else None

    # the methods of the reference's ``Dataset.xcorr`` accessor (core/accessors.py:52-110) that
    # belong to this path, on the container itself
    def process(self, client, **kwargs):
        from .process import process
        return process(self, client, **kwargs)

    def postprocess(self, **kwargs):
        from .postprocess import postprocess
        return postprocess(self, **kwargs)

    def write(self, path, **kwargs):
        from ..io import write
        return write(self, path, **kwargs)

    def merge(self, *others, **kwargs):
        from ..io import merge
        return merge(self, *others, **kwargs)

    def to_xarray(self):
        """The container as the reference's ``xarray.Dataset`` (variable, coordinate and attribute names of
        xcorr/core/init.py:153-320).  NOTE: xarray is not installable in this project's build and test
        images, so neither this method nor :meth:`from_xarray` has ever executed there; both are
        written against the reference's layout and against xarray's documented API only."""
        import xarray as xr
        ds = xr.Dataset()
        ds.coords['pair'] = self.pair
        ds.coords['time'] = (self.time * 1e9).astype('datetime64[ns]') if getattr(self, 'time_is_datetime', False) \
            else self.time
        ds.coords['lag'] = ('lag', self.lag, self.lag_attrs)
        ds['time'].attrs = dict(self.time_attrs)
        ds['pair'].attrs = dict(getattr(self, 'pair_attrs', {}))
        ds['cc'] = (('pair', 'time', 'lag'), self.cc, self.cc_attrs)
        ds['status'] = (('pair', 'time'), self.status)
        ds['pair_offset'] = (('pair', 'time'), self.pair_offset)
        ds['time_offset'] = (('pair', 'time'), self.time_offset)
        if self.hash is not None:
            ds['hash'] = (('pair', 'time'), self.hash)
        ds.attrs = dict(self.attrs)
        return ds

    @classmethod
    def from_xarray(cls, ds):
        """A :class:`CCDataset` holding the data of a reference-style ``xarray.Dataset`` (the output of
        ``xcorr.init`` / ``xcorr.read``: core/init.py:153-320).  Only duck-typed attribute access is used
        (``ds.pair.values``, ``ds.lag.attrs`` ...), so any object with that shape converts.  See the note in
        :meth:`to_xarray`: not executed in any environment available to this project."""
        t = np.asarray(ds['time'].values)
        is_dt = np.issubdtype(t.dtype, np.datetime64)
        time = t.astype('datetime64[ns]').astype(np.int64) * 1e-9 if is_dt else t.astype(np.float64)
        la = dict(ds['lag'].attrs)
        for k in ('npts', 'index_min', 'index_max'):
            la[k] = int(la[k])
        for k in ('sampling_rate', 'delta'):
            la[k] = float(la[k])
        ta = dict(ds['time'].attrs)
        cc = np.asarray(ds['cc'].values)
        out = cls([str(p) for p in np.asarray(ds['pair'].values).reshape(-1)], time, np.asarray(ds['lag'].values, np.float64),
                  la, dict(ds['cc'].attrs), ta, dtype=cc.dtype, hash_waveforms='hash' in ds)
        out.time_is_datetime = bool(is_dt)
        out.pair_attrs = dict(ds['pair'].attrs)
        out.cc[...] = cc
        out.status[...] = np.asarray(ds['status'].values)
        out.pair_offset[...] = np.asarray(ds['pair_offset'].values)
        out.time_offset[...] = np.asarray(ds['time_offset'].values)
        if out.hash is not None:
            out.hash[...] = np.asarray(ds['hash'].values)
        out.attrs = dict(ds.attrs)
        return out

    def update_xarray(self, ds):
        """Write the processed variables back into `ds` in place (what ``xcorr.process`` does to its
        argument, core/process.py:82-165): cc, status, offsets, hashes and the ``unbiased`` flag."""
        ds['cc'].values[...] = self.cc
        ds['cc'].attrs.update(self.cc_attrs)
        ds['status'].values[...] = self.status
        ds['pair_offset'].values[...] = self.pair_offset
        ds['time_offset'].values[...] = self.time_offset
        if self.hash is not None and 'hash' in ds:
            ds['hash'].values[...] = self.hash
        return ds


def is_xarray_dataset(obj):
    """True for an ``xarray.Dataset`` (without importing xarray when it is absent)."""
    import sys
    xr = sys.modules.get('xarray')
    return xr is not None and isinstance(obj, xr.Dataset)


def _seconds(t):
    """float seconds (POSIX) of a float / ISO string / datetime64 / pandas.Timestamp / datetime."""
    if isinstance(t, (int, float, np.integer, np.floating)):
        return float(t), False
    if isinstance(t, np.datetime64):
        return float((t - np.datetime64(0, 's')) / np.timedelta64(1, 'ns')) * 1e-9, True
    try:
        import pandas as pd
        ts = pd.Timestamp(t)
        if ts.tzinfo is None:
            ts = ts.tz_localize('UTC')
        return ts.timestamp(), True
    except ModuleNotFoundError:
        return float(np.datetime64(t).astype('datetime64[ns]').astype(np.int64)) * 1e-9, True


def init(pair, starttime, endtime, preprocess=None, attrs=None, sampling_rate=None,
         window_length=86400., window_overlap=.875, clip_lag=None, unbiased_cc=False,
         closed='left', dtype=np.float32, inventory=None, stationary_poi=None,
         hash_waveforms=False, normalize=True):
    """Build the container -- the reference's signature (core/init.py:25-40):

        init(pair, starttime, endtime, preprocess, attrs, sampling_rate, window_length=86400.,
             window_overlap=.875, clip_lag=None, unbiased_cc=False, closed='left',
             dtype=np.float32, inventory=None, stationary_poi=None, hash_waveforms=False)

    pair : str (the reference holds exactly one pair per dataset, core/init.py:115-116,154; a
        list is also accepted here so that one container feeds the batched engine).
    starttime, endtime : pandas.Timestamp / ISO string / numpy.datetime64 / datetime, or plain
        seconds.  Window centres are ``date_range(start, end, freq=window_length*(1-overlap))``
        (core/init.py:163-168; `closed` = 'left' drops the end point, 'right' the start, None
        keeps both).  The time axis is held in float seconds; ``to_xarray`` turns it back into
        datetime64 when it came from timestamps.
    preprocess : dict of operation lists per channel (kept verbatim in the pair attrs; the sha256
        integrity hash of stream/process.py:557-566 belongs to the storage layer, out of scope).
    attrs : dict with 'institution', 'author', 'source' (same TypeError / KeyError as
        core/init.py:106-113 when given).
    clip_lag : None or (a, b) seconds -> ``argmin|lag-a| : argmin|lag-b|`` (core/init.py:176-196,
        end exclusive); a single float c means (-|c|, |c|) (the reference raises there,
        ``tuple(-c, c)``: SURVEY a12).
    inventory, stationary_poi : pair geodesy (util/receiver.py) is out of scope: accepted, unused.
    hash_waveforms : adds the ``hash`` variable ('n/a' per (pair, time), core/init.py:252-272) that
        ``process`` fills with the sha256 of the two preprocessed traces (util/hasher.py).

    Legacy numeric form of this build, still accepted:
        init(pair, start_s, end_s, sampling_rate, window_length, window_overlap, ...).
    """
    if isinstance(preprocess, (int, float, np.integer, np.floating)) and not isinstance(preprocess, bool):
        # legacy positional form: (pair, t0, t1, sampling_rate[, window_length[, window_overlap]])
        legacy_wl = attrs if isinstance(attrs, (int, float)) else None
        legacy_ov = sampling_rate if isinstance(sampling_rate, (int, float)) and legacy_wl is not None else None
        sampling_rate, preprocess, attrs = float(preprocess), None, None
        if legacy_wl is not None:
            window_length = float(legacy_wl)
        if legacy_ov is not None:
            window_overlap = float(legacy_ov)
    if sampling_rate is None:
        raise TypeError("init() missing required argument: 'sampling_rate'")
    if attrs is not None:
        if not isinstance(attrs, dict):
            raise TypeError('attrs should be a dictionary')
        if 'institution' not in attrs:
            raise KeyError("attrs['institution'] = 'Institution, department'!")
        if 'author' not in attrs:
            raise KeyError("attrs['author'] = 'Name - E-mail'!")
        if 'source' not in attrs:
            raise KeyError("attrs['source'] = 'Data source description'!")
    if not isinstance(pair, (str, list, tuple)):
        raise TypeError("pair should be receiver pair string!")
    starttime, dt0 = _seconds(starttime)
    endtime, dt1 = _seconds(endtime)
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
        elif isinstance(clip_lag, tuple) and len(clip_lag) == 2:
            if not isinstance(clip_lag[0], float):            # core/init.py:183-187 (checks [0] twice)
                raise TypeError(msg)
        else:
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
    ds = CCDataset(pairs, time, lag[nmin:nmax], lag_attrs, cc_attrs,
                   time_attrs, dtype=dtype, hash_waveforms=hash_waveforms)
    ds.time_is_datetime = bool(dt0 or dt1)
    ds.pair_attrs = {'long_name': 'Cross-correlation receiver pair', 'standard_name': 'receiver_pair',
                     'units': '-', 'preprocess': preprocess if preprocess is not None else {}}
    if attrs is not None:                                     # global attributes of core/init.py:127-151
        ds.attrs = {
            'title': ((attrs['title'] if 'title' in attrs else '') + ' Cross-correlations').strip(),
            'institution': attrs['institution'], 'author': attrs['author'], 'source': attrs['source'],
            'history': '', 'comment': attrs['comment'] if 'comment' in attrs else 'n/a',
            'references': ('Bendat, J. Samuel, & Piersol, A. Gerald. (1971). Random data : analysis and '
                           'measurement procedures. New York (N.Y.): Wiley-Interscience.'),
            'Conventions': 'CF-1.9',
        }
    return ds
