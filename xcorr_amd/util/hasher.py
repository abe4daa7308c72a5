"""
Waveform / object hashes of ``xcorr/util/hasher.py`` without ObsPy objects.

``hash_obj`` (hasher.py:106-139) is the reference's code path verbatim in behaviour:
sha256 of ``json.dumps(obj minus the ignored keys, separators=(',', ':'),
sort_keys=True, indent=4, default=...)``.  ``hash_trace`` / ``hash_stream``
(hasher.py:142-215) take the trace header as a plain dict; the reference
serialises ObsPy's ``UTCDateTime`` start / end times through ``repr`` --
``UTCDateTime(2015, 1, 15, 0, 0)`` -- which is rebuilt here from POSIX seconds
(restated from ObsPy's documented repr, **unverified against obspy**).
"""
import datetime
import hashlib
import json

import numpy as np

__all__ = ['hash_obj', 'hash_trace', 'hash_stream', 'utcdatetime_repr', 'trace_stats']

_ignore_keys = ['sha256_hash', 'sha256_hash_metadata', 'add_offset', 'scale_factor']
_trace_keys = ['network', 'station', 'location', 'channel', 'starttime', 'endtime', 'sampling_rate', 'delta', 'npts']


def _to_serializable(obj):
    if isinstance(obj, (np.int8, np.int16, np.int32, np.int64)):
        return int(obj)
    if isinstance(obj, (np.float32, np.float64)):
        return float(obj)
    if isinstance(obj, _UTC):
        return obj.text
    return repr(obj)


class _UTC:
    """carries the text ``repr(UTCDateTime)`` would give through json's `default` hook."""
    def __init__(self, text):
        self.text = text


def to_json(obj):
    keep = {k: obj[k] for k in obj.keys() if k not in _ignore_keys}
    return json.dumps(keep, separators=(',', ':'), sort_keys=True, indent=4, skipkeys=False, default=_to_serializable)


def hash_obj(obj, hashlib_obj=None):
    h = hashlib_obj or hashlib.sha256()
    h.update(to_json(obj).encode('utf-8'))
    return None if hashlib_obj else h.hexdigest()


def utcdatetime_repr(seconds):
    """``repr(obspy.UTCDateTime(seconds))``: ``'UTCDateTime' + repr(datetime)[17:]`` with
    microsecond resolution (ObsPy's default precision 6)."""
    us = int(round(float(seconds) * 1e6))
    dt = datetime.datetime(1970, 1, 1) + datetime.timedelta(microseconds=us)
    return 'UTCDateTime' + repr(dt)[17:]


def trace_stats(seed_id, starttime, sampling_rate, npts):
    """header dict of one trace: `seed_id` = 'NET.STA.LOC.CHA', `starttime` POSIX seconds."""
    net, sta, loc, cha = (seed_id.split('.') + ['', '', '', ''])[:4]
    fs = float(sampling_rate)
    return {'network': net, 'station': sta, 'location': loc, 'channel': cha, 'starttime': float(starttime),
            'endtime': float(starttime) + (int(npts) - 1) / fs, 'sampling_rate': fs, 'delta': 1.0 / fs,
            'npts': int(npts)}


def hash_trace(stats, data, hashlib_obj=None):
    """hasher.py:174-181: json of the nine header keys, then the bytes of every sample."""
    h = hashlib_obj or hashlib.sha256()
    head = {k: stats[k] for k in _trace_keys}
    for k in ('starttime', 'endtime'):
        if not isinstance(head[k], str):
            head[k] = _UTC(utcdatetime_repr(head[k]))
        else:
            head[k] = _UTC(head[k])
    h.update(to_json(head).encode('utf-8'))
    h.update(np.ascontiguousarray(data).tobytes())          # == update(sample.tobytes()) sample by sample
    return None if hashlib_obj else h.hexdigest()


def hash_stream(traces, hashlib_obj=None):
    """hasher.py:210-215: one running sha256 over the traces in order; `traces` =
    iterable of (stats, data)."""
    h = hashlib_obj or hashlib.sha256()
    for stats, data in traces:
        hash_trace(stats, data, hashlib_obj=h)
    return None if hashlib_obj else h.hexdigest()
