"""Array-level preprocess ops with ObsPy's parameter names (see package docstring)."""
import functools

import numpy as np
import scipy.signal

from .. import ops as _ops
from . import segments as _seg


# -- filter design (host side, scipy) -----------------------------------------
def butter_sos(ftype, sampling_rate, corners=4, freq=None, freqmin=None, freqmax=None):
    """ObsPy ``Trace.filter('highpass'|'lowpass'|'bandpass', ...)`` design: Butterworth
    ``iirfilter(corners, Wn, btype, ftype='butter', output='zpk')`` -> ``zpk2sos``.  Designs are cached: a list of
    ops runs the same few filters for every station-day, and a design costs as much host time as filtering a day."""
    key = tuple(None if v is None else float(v) for v in (sampling_rate, freq, freqmin, freqmax))
    return _butter_design(str(ftype), int(corners), key).copy()


@functools.lru_cache(maxsize=64)
def _butter_design(ftype, corners, key):
    sampling_rate, freq, freqmin, freqmax = key
    fe = 0.5 * sampling_rate
    if ftype == 'bandpass':
        wn, btype = [freqmin / fe, freqmax / fe], 'band'
    elif ftype in ('highpass', 'lowpass'):
        wn, btype = freq / fe, ftype[:-4]
    else:
        raise NotImplementedError(f'filter type "{ftype}" is not built')
    z, p, k = scipy.signal.iirfilter(corners, wn, btype=btype, ftype='butter', output='zpk')
    return scipy.signal.zpk2sos(z, p, k)


def cheby2_lowpass_sos(freq, sampling_rate, maxorder=12):
    """ObsPy ``lowpass_cheby_2`` design (the anti-alias filter of ``Trace.decimate``):
    stop band at `freq` with 96 dB attenuation, pass band lowered by 1 % steps until the
    order given by ``cheb2ord`` is <= maxorder.  Cached (2.6 ms of scipy per design: more than the decimation of a
    station-day on the device)."""
    return _cheby2_design(float(freq), float(sampling_rate), int(maxorder)).copy()


@functools.lru_cache(maxsize=64)
def _cheby2_design(freq, sampling_rate, maxorder):
    nyquist = sampling_rate * 0.5
    rp, rs, order = 1, 96, 1e99
    ws = freq / nyquist
    wp = ws
    if ws > 1:
        ws = 1.0
    wn = ws
    while order > maxorder:
        wp = wp * 0.99
        order, wn = scipy.signal.cheb2ord(wp, ws, rp, rs, analog=0)
    z, p, k = scipy.signal.cheby2(order, rs, wn, btype='low', analog=0, output='zpk')
    return scipy.signal.zpk2sos(z, p, k)


def cosine_taper_length(npts, sampling_rate, max_percentage=0.05, max_length=None):
    """Flank length of ObsPy ``Trace.taper(type='cosine', max_percentage, max_length)``."""
    wlen = int(max_percentage * npts) if max_percentage is not None else npts // 2
    if max_length is not None:
        wlen = min(wlen, int(max_length * sampling_rate))
    return min(wlen, npts // 2)


# -- operations -----------------------------------------------------------------
def detrend(x, type='demean', ctx=None):
    """``Trace.detrend('demean' | 'constant' | 'linear')``."""
    if type in ('demean', 'constant'):
        return _ops.demean(ctx, x)
    if type in ('linear', 'simple'):
        if type == 'simple':
            raise NotImplementedError('detrend type "simple" is not built')
        return _ops.detrend_linear(ctx, x)
    raise NotImplementedError(f'detrend type "{type}" is not built')


def filter(x, sampling_rate, type, corners=4, zerophase=False, ctx=None, **options):
    """``Trace.filter(type, freq= | freqmin=, freqmax=, corners=4, zerophase=False)``."""
    sos = butter_sos(type, sampling_rate, corners, **options)
    return _ops.sosfilt_zerophase(ctx, x, sos) if zerophase else _ops.sosfilt(ctx, x, sos)


def taper(x, sampling_rate, max_percentage=0.05, type='cosine', max_length=None, ctx=None):
    """``Trace.taper(max_percentage, type='cosine'|'hann', max_length)``, both sides."""
    if type not in ('cosine', 'hann'):
        raise NotImplementedError(f'taper type "{type}" is not built')
    n = x.shape[-1]
    wlen = cosine_taper_length(n, sampling_rate, max_percentage, max_length)
    if wlen < 1:
        return x
    return _ops.taper(ctx, x, wlen, wtype=type)


def decimate(x, sampling_rate, factor, no_filter=False, ctx=None):
    """``Trace.decimate(factor, no_filter=False)``: Chebyshev-II low-pass at
    0.5 * sampling_rate / factor (``lowpass_cheby_2``, maxorder 12, causal), then every
    factor-th sample.  Returns (y, new_sampling_rate)."""
    factor = int(factor)
    if factor > 16 and not no_filter:
        raise ArithmeticError('Automatic filter design is unstable for decimation factors above 16. '
                              'Manual decimation is necessary.')
    sos = None if no_filter else cheby2_lowpass_sos(0.5 * sampling_rate / factor, sampling_rate)
    return _ops.decimate(ctx, x, factor, sos), sampling_rate / float(factor)


@functools.lru_cache(maxsize=4)
def _named_window_half(name, npts):
    """First npts // 2 + 1 samples of ``ifftshift(get_window(name, npts))`` as float32 (read-only, shared)."""
    w = np.fft.ifftshift(scipy.signal.get_window(name, npts))[:npts // 2 + 1].astype(np.float32)
    w.setflags(write=False)
    return w


def resample(x, sampling_rate, new_sampling_rate, window='hann', no_filter=True, strict_length=False, ctx=None):
    """``Trace.resample(sampling_rate, window='hann', no_filter=True, strict_length=False)``: Fourier
    method -- rfft, frequency-domain window ``ifftshift(get_window(window, npts))`` (or a callable
    of ``fftfreq(npts)`` / an array of npts samples / None), linear interpolation of the spectrum
    onto the new frequency grid, irfft to ``num = int(npts / factor)`` samples, scaled by
    num / npts.  ``no_filter=False`` first applies the Chebyshev-II anti-alias low-pass of
    ``decimate`` (factor <= 16).  Returns (y, new_sampling_rate)."""
    npts = x.shape[-1]
    factor = sampling_rate / float(new_sampling_rate)
    if strict_length and npts % factor != 0.0:
        raise ValueError('End time of trace would change and strict_length=True.')
    if not no_filter:
        if factor > 16:
            raise ArithmeticError('Automatic filter design is unstable for resampling factors (current sampling '
                                  'rate/new sampling rate) above 16. Manual resampling is necessary.')
        x = _ops.sosfilt(ctx, x if not isinstance(x, np.ndarray) else x.copy(),
                         cheby2_lowpass_sos(sampling_rate * 0.5 / float(factor), sampling_rate))
    num = int(npts / factor)
    wh = None
    if window is not None:
        if callable(window):
            large_w = np.asarray(window(np.fft.fftfreq(npts)), dtype=np.float64)
        elif isinstance(window, np.ndarray):
            if window.shape != (npts,):
                raise ValueError('Window has to be the same shape as the data.')
            large_w = window.astype(np.float64)
        else:
            large_w = None
            wh = _named_window_half(str(window), npts)       # cached: 4.32 M samples of a window cost more than the transform
        if large_w is not None:
            wh = large_w[:npts // 2 + 1]
    # bin position of new frequency j: (j * new_rate / num) / (old_rate / npts)
    ratio = (float(new_sampling_rate) / num) / (float(sampling_rate) / npts)
    return _ops.resample(ctx, x, num, ratio, wh), float(new_sampling_rate)


def interpolate(x, sampling_rate, new_sampling_rate, method='lanczos', a=20, starttime_offset=0.0, npts=None,
                ctx=None):
    """``Trace.interpolate(sampling_rate, method='lanczos', a=...)``: new sample i sits at
    ``starttime_offset + i / new_sampling_rate`` seconds after the first old sample;
    default npts covers the old time span.  Returns (y, new_sampling_rate)."""
    if method != 'lanczos':
        raise NotImplementedError(f'interpolation method "{method}" is not built')
    n = x.shape[-1]
    dt_old, dt_new = 1.0 / sampling_rate, 1.0 / float(new_sampling_rate)
    if npts is None:
        span = (n - 1) * dt_old - starttime_offset
        npts = int(np.floor(span / dt_new)) + 1
    if starttime_offset < 0 or starttime_offset + (npts - 1) * dt_new > (n - 1) * dt_old + 1e-9 * dt_old:
        raise ValueError('The time span of the new samples must lie within the old one')
    y = _ops.lanczos(ctx, x, npts, starttime_offset / dt_old, dt_new / dt_old, a)
    return y, float(new_sampling_rate)


def running_rms(x, sampling_rate, window=1., ctx=None):
    """xcorr/stream/running_rms.py:73-113: rms over consecutive windows of `window`
    seconds.  ObsPy's ``Trace.slide(window, window)`` windows include both end points
    (window*fs + 1 samples, neighbours share one sample).  Returns (rms, 1/window)."""
    n = x.shape[-1]
    step = int(round(window * sampling_rate))
    win = step + 1
    nwin = (n - win) // step + 1 if n >= win else 0
    if nwin < 1:
        raise ValueError('trace shorter than one window')
    return _ops.block_rms(ctx, x, win, step, nwin), 1.0 / window


def rotate(components, method, back_azimuth=None, ctx=None, **unused):
    """``Stream.rotate(method, back_azimuth=...)`` on a component set ``{channel code: array}``
    (stream/process.py:70-73; scripts/cc.py:83-84 runs ``'->ZNE'`` then ``'NE->RT'``).

    ``'NE->RT'``: the channels ending in N and E (same band / instrument prefix) become R and T, renamed like ObsPy
    does (``BHN, BHE -> BHR, BHT``); `back_azimuth` in degrees is required (ObsPy can also take it from trace
    stats set by the client; there is no inventory here).
    ``'->ZNE'``: components that are already Z / N / E are returned as they are; anything else (``BH1``, ``BH2``)
    needs the orientations of an ObsPy inventory and raises."""
    if not isinstance(components, dict):
        raise TypeError('rotate needs a component set {channel: array} (e.g. {"BHZ": z, "BHN": n, "BHE": e})')
    if method == '->ZNE':
        odd = [c for c in components if c[-1:] not in ('Z', 'N', 'E')]
        if odd:
            raise NotImplementedError(f'rotate("->ZNE") of {odd} needs the channel orientations of an ObsPy inventory')
        return components
    if method != 'NE->RT':
        raise NotImplementedError(f'rotate method "{method}" is not built (built: "->ZNE" for ZNE input, "NE->RT")')
    if back_azimuth is None:
        raise ValueError('rotate("NE->RT") needs back_azimuth (degrees)')
    out = dict(components)
    done = 0
    for cn in [c for c in components if c.endswith('N')]:
        ce = cn[:-1] + 'E'
        if ce not in components:
            continue
        r, t = _seg.rotate_ne_rt(components[cn], components[ce], float(back_azimuth), ctx=ctx)
        del out[cn], out[ce]
        out[cn[:-1] + 'R'], out[cn[:-1] + 'T'] = r, t
        done += 1
    if not done:
        raise ValueError('rotate("NE->RT"): no pair of N / E components with a common prefix')
    return out


def select(components, channel=None, component=None, **other):
    """``Stream.select(channel=..., component=...)`` on a component set (stream/process.py:74-77; scripts/cc.py:85
    keeps ``BHR``): UNIX wildcards as in ObsPy.  Station / network / location selectors do not apply to a
    component set of one receiver and are refused."""
    import fnmatch
    if not isinstance(components, dict):
        raise TypeError('select needs a component set {channel: array}')
    extra = [k for k, v in other.items() if v is not None]
    if extra:
        raise NotImplementedError(f'select by {extra} is not built (a component set holds one receiver)')
    keep = {c: v for c, v in components.items()
            if (channel is None or fnmatch.fnmatchcase(c, channel))
            and (component is None or fnmatch.fnmatchcase(c[-1:], component))}
    if not keep:
        raise ValueError(f'select(channel={channel!r}, component={component!r}) matches none of {list(components)}')
    return keep


_NEEDS_INVENTORY = ('attach_response', 'remove_response', 'simulate')

_OPS = {'detrend': detrend, 'filter': filter, 'taper': taper, 'decimate': decimate, 'resample': resample,
        'interpolate': interpolate, 'running_rms': running_rms, 'merge': _seg.merge, 'trim': _seg.trim,
        'remove_sensitivity': _seg.remove_sensitivity, 'normalize': _seg.normalize,
        'rotate': rotate, 'select': select}


def apply_operations(x, sampling_rate, operations, ctx=None, t0=0.0, starttime=None, endtime=None,
                     sensitivity=None, return_t0=False):
    """Run a reference-style operation list ``[(name, params), ...]``
    (xcorr/stream/process.py:300-412) on an array.  Returns (y, sampling_rate), or
    (y, sampling_rate, t0) with ``return_t0``.

    `x` is one array ``[..., npts]`` starting at `t0` seconds, or -- when the list
    starts with ``merge`` -- a list of segments ``[(data, starttime), ...]``.
    `starttime` / `endtime` / `sensitivity` are the dynamic parameters the reference
    injects from the window bounds and the inventory (stream/process.py:153-196):
    ``trim`` cuts to [starttime, endtime], ``remove_sensitivity`` divides by
    `sensitivity`.

    A COMPONENT SET ``{channel code: array}`` (``{'BHZ': z, 'BHN': n, 'BHE': e}``, equal shapes and one time axis)
    takes the place of the multi-trace ``obspy.Stream`` of one receiver: per-trace operations run on every
    component, ``rotate`` and ``select`` act on the set (scripts/cc.py:70-90, the BHR list).  When ``select``
    leaves one component the result is that array, otherwise the set.  ``attach_response`` / ``remove_response`` /
    ``simulate`` need an ObsPy inventory and raise."""
    if isinstance(x, dict):
        comps = dict(x)
        if not comps:
            raise ValueError('empty component set')
        ops_left = list(operations)
        while ops_left:
            name, params = ops_left[0]
            if name in ('rotate', 'select'):
                comps = _OPS[name](comps, ctx=ctx, **params) if name == 'rotate' else select(comps, **params)
                ops_left.pop(0)
                continue
            # the run of per-trace operations up to the next set operation, component by component
            k = next((i for i, (nm, _) in enumerate(ops_left) if nm in ('rotate', 'select')), len(ops_left))
            run, ops_left = ops_left[:k], ops_left[k:]
            res = {c: apply_operations(v, sampling_rate, run, ctx=ctx, t0=t0, starttime=starttime, endtime=endtime,
                                       sensitivity=(sensitivity.get(c) if isinstance(sensitivity, dict) else sensitivity),
                                       return_t0=True) for c, v in comps.items()}
            comps = {c: r[0] for c, r in res.items()}
            rates = {r[1] for r in res.values()}
            starts = {r[2] for r in res.values()}
            if len(rates) != 1 or len(starts) != 1:
                raise ValueError('the components left the operations with different time axes')
            sampling_rate, t0 = rates.pop(), starts.pop()
        y = next(iter(comps.values())) if len(comps) == 1 else comps
        return (y, sampling_rate, t0) if return_t0 else (y, sampling_rate)
    for name, params in operations:
        if callable(name):
            x = name(x, **params)
            continue
        if name in _NEEDS_INVENTORY:
            raise NotImplementedError(f'"{name}" needs an ObsPy inventory (instrument response): not built')
        if name in ('rotate', 'select'):
            raise TypeError(f'"{name}" acts on a component set {{channel: array}}, not on a single trace')
        if name not in _OPS:
            raise NotImplementedError(f'"{name}" is not an implemented operation or callable.')
        p = dict(params)
        if name == 'merge':
            if isinstance(x, np.ndarray):
                continue                               # a single segment is already merged
            x, t0 = _seg.merge(x, sampling_rate, **p)
            if np.isnan(x).any():
                raise ValueError('merge left gaps (masked samples); pass a fill_value')
            x = np.ascontiguousarray(x, dtype=np.float32)
        elif name == 'trim':
            p.setdefault('starttime', starttime)
            p.setdefault('endtime', endtime)
            x, t0 = _seg.trim(x, sampling_rate, t0, **p)
        elif name == 'remove_sensitivity':
            if sensitivity is None:
                raise ValueError('remove_sensitivity needs the channel sensitivity (inventory)')
            x = _seg.remove_sensitivity(x, sensitivity)
        elif name == 'normalize':
            x = _seg.normalize(x, **p)
        elif name == 'detrend':
            x = detrend(x, ctx=ctx, **p)
        elif name == 'taper':
            x = taper(x, sampling_rate, ctx=ctx, **p)
        elif name == 'filter':
            x = filter(x, sampling_rate, ctx=ctx, **p)
        elif name == 'decimate':
            x, sampling_rate = decimate(x, sampling_rate, ctx=ctx, **p)
        elif name == 'resample':
            new_rate = p.pop('sampling_rate')
            x, sampling_rate = resample(x, sampling_rate, new_rate, ctx=ctx, **p)
        elif name == 'interpolate':
            new_rate = p.pop('sampling_rate')
            if 'starttime' in p:                       # ObsPy: absolute start time of the new axis
                p['starttime_offset'] = p.pop('starttime') - t0
            t0 = t0 + p.get('starttime_offset', 0.0)
            x, sampling_rate = interpolate(x, sampling_rate, new_rate, ctx=ctx, **p)
        elif name == 'running_rms':
            x, sampling_rate = running_rms(x, sampling_rate, ctx=ctx, **p)
    return (x, sampling_rate, t0) if return_t0 else (x, sampling_rate)
