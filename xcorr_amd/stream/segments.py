"""
Segment bookkeeping of the preprocess lists (xcorr/stream/process.py:50-53, 62-69,
70-77, 86-90; used as in xcorr/scripts/cc.py:48-62): ``merge``, ``trim``,
``remove_sensitivity`` and the ``NE->RT`` rotation, on plain arrays.

A segment is ``(data[npts], starttime_seconds)`` at a common sampling rate.  These
are index arithmetic and copies, not arithmetic on samples: they run on the host
and hand contiguous float32 rows to the device ops.  ObsPy is not vendored by the
reference nor installable here; the rules below restate ObsPy's documented
``Stream.merge`` / ``Trace.trim`` behaviour and are **unverified against obspy**.
"""
import math

import numpy as np

__all__ = ['merge', 'trim', 'remove_sensitivity', 'rotate_ne_rt']


def _round_away(x):
    """round half away from zero (ObsPy's ``round_away``)."""
    return int(math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def merge(segments, sampling_rate, method=1, fill_value=None, interpolation_samples=0):
    """``Stream.merge(method, fill_value, interpolation_samples)`` for the segments of ONE
    channel: sort by start time and add them cumulatively (``Trace.__add__``).

    Overlaps -- method 0: samples of the overlap that differ are discarded (gap rule
    applies to them); method 1: the later segment is used from the overlap on, with
    `interpolation_samples` samples (-1: the whole overlap) linearly interpolated
    between the last kept sample of the earlier and the later segment; a segment
    fully contained in an earlier one is dropped.
    Gaps -- ``fill_value`` None: NaN (ObsPy masks them); a number; 'latest': last sample
    before the gap; 'interpolate': straight line between the neighbouring samples.

    Returns ``(data, starttime)``; float dtype of the first segment."""
    if method not in (0, 1):
        raise NotImplementedError('merge methods 0 and 1 are built')
    segs = sorted(((np.asarray(d), float(t)) for d, t in segments if len(d)), key=lambda s: s[1])
    if not segs:
        raise ValueError('no samples to merge')
    fs = float(sampling_rate)
    cur, t0 = segs[0][0].copy(), segs[0][1]
    if cur.dtype.kind != 'f':
        cur = cur.astype(np.float64)
    for rt, rt0 in segs[1:]:
        rt = rt.astype(cur.dtype, copy=False)
        lt_end = t0 + (cur.size - 1) / fs
        rt_end = rt0 + (rt.size - 1) / fs
        delta = _round_away((rt0 - lt_end) * fs) - 1          # samples missing between the two
        if delta < 0 and rt_end > lt_end + 0.5 / fs:           # overlap, later segment reaches further
            ov = -delta
            if method == 0:
                a, b = cur[-ov:], rt[:ov]
                gap = np.where(a == b, a, np.nan)
                if fill_value is not None and np.isnan(gap).any():
                    gap = _fill(gap, cur[:-ov], rt[ov:], fill_value)
                cur = np.concatenate([cur[:-ov], gap, rt[ov:]])
            else:
                k = ov if interpolation_samples == -1 else min(int(interpolation_samples), ov)
                ls = cur[-ov - 1] if cur.size > ov else cur[0]
                if k >= rt.size:                               # behaves like a contained segment
                    continue
                ramp = np.linspace(ls, rt[k], k + 2)[1:-1].astype(cur.dtype)
                cur = np.concatenate([cur[:-ov], ramp, rt[k:]])
        elif delta < 0:                                        # contained segment
            if method == 0:
                i0 = _round_away((rt0 - t0) * fs)
                a = cur[i0:i0 + rt.size]
                mid = np.where(a == rt[:a.size], a, np.nan)
                if fill_value is not None and np.isnan(mid).any():
                    mid = _fill(mid, cur[:i0], cur[i0 + rt.size:], fill_value)
                cur = np.concatenate([cur[:i0], mid, cur[i0 + rt.size:]])
            # method 1: dropped
        elif delta == 0:
            cur = np.concatenate([cur, rt])
        else:
            gap = _fill(np.full(delta, np.nan, cur.dtype), cur, rt, fill_value)
            cur = np.concatenate([cur, gap, rt])
    return cur, t0


def _fill(gap, left, right, fill_value):
    """values for the NaN entries of `gap` between `left` and `right`."""
    gap = gap.copy()
    m = np.isnan(gap)
    if fill_value is None:
        return gap
    if isinstance(fill_value, str):
        if fill_value == 'latest':
            gap[m] = left[-1]
        elif fill_value == 'interpolate':
            line = np.linspace(left[-1], right[0], gap.size + 2)[1:-1]
            gap[m] = line[m]
        else:
            raise ValueError(f'unknown fill_value "{fill_value}"')
    else:
        gap[m] = fill_value
    return gap


def _is_device(x):
    from ..engine import DeviceArray
    return isinstance(x, DeviceArray)


def _device_columns(x, i0, i1):
    """Columns [i0, i1) of device-resident rows as a contiguous DeviceArray (row-wise device copies)."""
    import ctypes as C
    from .. import _ffi
    from ..engine import DeviceArray
    rows = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
    n, w, isz = x.shape[-1], i1 - i0, x.dtype.itemsize
    pitch = getattr(x, '_pitch', n * isz)
    y = DeviceArray(x.ctx, tuple(x.shape[:-1]) + (w,), x.dtype)
    for r in range(rows):
        _ffi.check(_ffi.lib().xcx_memcpy_d2d(x.ctx.handle, C.c_void_p(y.ptr + r * w * isz),
                                             C.c_void_p(x.ptr + r * pitch + i0 * isz), w * isz))
    return y


def trim(x, sampling_rate, t0, starttime=None, endtime=None, pad=False, nearest_sample=True, fill_value=None):
    """``Trace.trim``: cut (or, with `pad`, extend) to [starttime, endtime] seconds.
    nearest_sample=True selects the samples nearest to the bounds; False only samples
    inside them.  Padding uses `fill_value` (None: NaN).  Returns ``(y, new_t0)``."""
    dev = _is_device(x)
    if not dev:
        x = np.asarray(x)
    fs = float(sampling_rate)
    n = x.shape[-1]
    if starttime is not None and endtime is not None and starttime > endtime:
        raise ValueError('startime is larger than endtime')
    i0, i1 = 0, n                                           # keep x[..., i0:i1], indices may leave [0, n)
    if starttime is not None:
        d = (starttime - t0) * fs
        i0 = _round_away(d) if nearest_sample else -int(math.floor(round(-d, 7)))
    if endtime is not None:
        t_end = t0 + (n - 1) / fs
        if nearest_sample:
            i1 = _round_away((endtime - t0) * fs) + 1
        else:
            i1 = n + int(math.floor(round((endtime - t_end) * fs, 7)))
    if not pad:
        i0, i1 = max(i0, 0), min(i1, n)
    if dev:
        # device-resident rows: the cut is a column slice copied into a buffer of its own; padding (samples that
        # do not exist) and empty results stay host features
        if i0 < 0 or i1 > n or i1 <= i0:
            raise ValueError('trim of device-resident rows must select existing samples (pad=False, non-empty)')
        if i0 == 0 and i1 == n:
            return x, t0
        return _device_columns(x, i0, i1), t0 + i0 / fs
    if i1 <= i0:
        return x[..., :0].copy(), t0 + i0 / fs
    fill = np.nan if fill_value is None else fill_value
    lead = x.shape[:-1]
    parts = []
    if i0 < 0:
        parts.append(np.full(lead + (min(-i0, i1 - i0),), fill, x.dtype if x.dtype.kind == 'f' else np.float64))
    if i1 > 0 and i0 < n:
        parts.append(x[..., max(i0, 0):min(i1, n)])
    if i1 > n:
        parts.append(np.full(lead + (min(i1 - n, i1 - i0),), fill, x.dtype if x.dtype.kind == 'f' else np.float64))
    y = np.concatenate(parts, axis=-1) if len(parts) > 1 else parts[0].copy()
    return y, t0 + i0 / fs


def remove_sensitivity(x, sensitivity):
    """``Trace.remove_sensitivity``: divide by the overall sensitivity of the channel
    (``inventory.get_response(...).instrument_sensitivity.value``).  Device-resident rows are scaled in place."""
    if _is_device(x):
        from .. import ops as _ops
        return _ops.scale_rows(x.ctx, x, 1.0 / np.asarray(sensitivity, dtype=np.float64))
    x = np.asarray(x)
    return (x / sensitivity).astype(x.dtype if x.dtype.kind == 'f' else np.float64)


def normalize(x, norm=None):
    """``Trace.normalize(norm=None)``: divide by the absolute maximum of the trace (or by `norm`; a negative
    `norm` is used by its absolute value, as ObsPy does with a warning).  An all-zero trace is returned
    unchanged.  Documented ObsPy behaviour restated, unverified vs obspy."""
    if _is_device(x):
        from .. import ops as _ops
        if norm is None:
            m = _ops.absmax_rows(x.ctx, x).astype(np.float64)
            f = np.where(m == 0, 1.0, 1.0 / np.where(m == 0, 1.0, m))
        else:
            if float(norm) == 0:
                return x
            f = 1.0 / abs(float(norm))
        return _ops.scale_rows(x.ctx, x, f)
    x = np.asarray(x)
    if norm is None:
        norm = np.abs(x).max(axis=-1, keepdims=True) if x.size else 1.0
        norm = np.where(norm == 0, 1.0, norm)
    else:
        norm = abs(float(norm))
        if norm == 0:
            return x
    return (x / norm).astype(x.dtype if x.dtype.kind == 'f' else np.float64)


def rotate_ne_rt(n, e, back_azimuth, ctx=None):
    """``rotate('NE->RT', back_azimuth)`` (obspy.signal.rotate.rotate_ne_rt): degrees from
    north of the direction from station to source.  Returns (radial, transverse):
    r = -e sin(ba) - n cos(ba), t = -e cos(ba) + n sin(ba).  The arithmetic runs in the library
    (``xcx_rotate2``): device-resident components are rotated in place, host arrays come back as copies."""
    if tuple(n.shape) != tuple(e.shape):
        raise TypeError('North and East component have different length.')
    if back_azimuth < 0 or back_azimuth > 360:
        raise ValueError('Back Azimuth should be between 0 and 360 degrees.')
    from .. import ops as _ops
    ba = math.radians(back_azimuth)
    # (n, e) <- (r, t)
    return _ops.rotate2(ctx, n, e, -math.cos(ba), -math.sin(ba), math.sin(ba), -math.cos(ba))
