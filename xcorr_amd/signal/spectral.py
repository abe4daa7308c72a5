"""``xcorr/signal/spectrogram.py`` (spectrogram / psd) on the GPU."""
import numpy as np
import scipy.signal

from .. import ops as _ops
from .ops1d import _is_da, _check_dim, xr


def _design(sampling_rate, duration, padding_factor, scaling, window):
    """win_len, nfft, window samples and scale exactly as scipy's `_spectral_helper`
    derives them from the reference's arguments (spectrogram.py:112-133)."""
    delta = 1.0 / sampling_rate
    duration = duration if (duration and duration > delta) else delta
    padding_factor = padding_factor if (padding_factor and padding_factor >= 1) else 1
    scaling = scaling or 'density'
    if scaling not in ['density', 'spectrum']:
        raise ValueError('Scaling should be either "density" or "spectrum"!')
    win_len = int(duration * sampling_rate)
    if win_len < 16:
        raise ValueError('Change duration to have at least 16 sample points!')
    nfft = int(win_len * padding_factor)
    if isinstance(window, (str, tuple)):
        win = scipy.signal.get_window(window, win_len)
    else:
        win = np.asarray(window, dtype=np.float64)
        if win.shape != (win_len,):
            raise ValueError('window must have length of nperseg')
    scale = 1.0 / (sampling_rate * (win * win).sum()) if scaling == 'density' else 1.0 / win.sum() ** 2
    return win_len, nfft, win, scale, scaling, duration, padding_factor


def spectrogram(x, duration=None, padding_factor=2, scaling=None, dim=None, sampling_rate=None,
                window=('tukey', .25), detrend='constant', **kwargs):
    """Spectrogram with one segment per sample (signal/spectrogram.py:27-196).

    numpy input ``[..., n]`` needs ``sampling_rate``; returns ``(freq, Sxx[..., nfreq, n])``.
    DataArray input follows the reference: new dimension ``freq`` before `dim`."""
    if kwargs:
        raise NotImplementedError(f'scipy.signal.spectrogram arguments {sorted(kwargs)} are not built')
    if detrend not in ('constant', False, None):
        raise NotImplementedError('only detrend="constant" or False is built')
    if _is_da(x):
        dim = _check_dim(x, dim)
        if 'sampling_rate' not in x[dim].attrs:
            raise ValueError('Dimension has no attribute "{sampling_rate}"!')
        if 'delta' not in x[dim].attrs:
            raise ValueError('Dimension has no attribute "{delta}"!')
        fs = x[dim].attrs['sampling_rate']
        xt = x.transpose(..., dim)
        freq, S = spectrogram(xt.values, duration, padding_factor, scaling, sampling_rate=fs, window=window,
                              detrend=detrend)
        y = xr.DataArray(S.astype(x.dtype), dims=xt.dims[:-1] + ('freq', dim),
                         coords={**{d: xt[d] for d in xt.dims if d in xt.coords}, 'freq': freq}, name='psd')
        y.freq.attrs = {'name': 'freq', 'long_name': 'Frequency', 'standard_name': 'frequency',
                        'units': f"{x[dim].attrs.get('units', 's')}-1"}
        y.attrs = {**x.attrs, 'scaling': scaling or 'density', 'mode': 'psd', 'duration': duration,
                   'padding_factor': padding_factor, 'centered': np.byte(True)}
        return y
    if sampling_rate is None:
        raise ValueError('sampling_rate is required for array input')
    win_len, nfft, win, scale, *_ = _design(sampling_rate, duration, padding_factor, scaling, window)
    x = np.asarray(x)
    if win_len <= 256 and nfft <= 8192:
        S = _ops.spectrogram(None, x, win_len, nfft, win, scale, detrend == 'constant')
    else:
        # segments longer than the MFMA kernel's tile: framed transforms, one per sample, then the
        # reference's zero columns at the edges (spectrogram.py:138-141)
        core = _ops.spectrogram_seg(None, x, win_len, 1, nfft, win, scale, detrend == 'constant')
        edge = int(np.rint(win_len / 2))
        npad = [(0, 0)] * (core.ndim - 1) + [(edge, x.shape[-1] - core.shape[-1] - edge)]
        S = np.pad(np.asarray(core), npad, mode='constant', constant_values=0)
    freq = np.linspace(0., sampling_rate / 2, int(nfft / 2 + 1))
    return freq, S.astype(x.dtype if x.dtype.kind == 'f' else np.float32)


psd = spectrogram


def spectrogram_mtc(x, duration=None, padding_factor=None, overlap=None, dim='lag', sampling_rate=None,
                    window=('tukey', .25), detrend='constant', **kwargs):
    """Spectrogram with manual time-segment control (signal/spectrogram.py:199-335):
    ``scipy.signal.spectrogram(nperseg=int(duration*fs), noverlap=int(nperseg*overlap),
    nfft=int(nperseg*padding_factor), scaling='density', mode='psd')``.  Defaults as the
    reference resolves them: padding_factor 1, overlap 0.9, duration at least one sample.

    numpy input ``[..., n]`` needs ``sampling_rate``; returns ``(f, t, Sxx[..., nfreq, nseg])``.
    DataArray input: dims of `x` without `dim`, then ``psd_f``, ``psd_t`` (``t`` offset by the
    first coordinate value)."""
    if kwargs:
        raise NotImplementedError(f'scipy.signal.spectrogram arguments {sorted(kwargs)} are not built')
    if detrend not in ('constant', False, None):
        raise NotImplementedError('only detrend="constant" or False is built')
    if _is_da(x):
        if dim not in x.dims:
            raise ValueError(f'x has no dimension "{dim}"')
        if 'sampling_rate' not in x[dim].attrs:
            raise ValueError('Dimension has no attribute "{sampling_rate}"!')
        if 'delta' not in x[dim].attrs:
            raise ValueError('Dimension has no attribute "{delta}"!')
        fs = x[dim].attrs['sampling_rate']
        xt = x.transpose(..., dim)
        f, t, S = spectrogram_mtc(xt.values, duration, padding_factor, overlap, sampling_rate=fs, window=window,
                                  detrend=detrend)
        t = t + x[dim].values[0]
        pf = padding_factor if (padding_factor and padding_factor >= 1) else 1
        ov = overlap if overlap and (0. < overlap < 1.) else .9
        du = duration if (duration and duration > x[dim].attrs['delta']) else x[dim].attrs['delta']
        coords = {d: xt[d] for d in xt.dims[:-1] if d in xt.coords}
        coords['psd_f'] = ('psd_f', f, {'long_name': 'Frequency', 'standard_name': 'frequency', 'units': 'Hz'})
        coords['psd_t'] = ('psd_t', t, {'long_name': 'Time', 'standard_name': 'time'})
        return xr.DataArray(S, dims=xt.dims[:-1] + ('psd_f', 'psd_t'), coords=coords, name='psd',
                            attrs={'long_name': 'Power Spectral Density', 'standard_name': 'power_spectral_density',
                                   'units': 'Hz-1', 'scaling': 'density', 'mode': 'psd', 'duration': du,
                                   'padding_factor': pf, 'overlap': ov, 'centered': np.byte(True)})
    if sampling_rate is None:
        raise ValueError('sampling_rate is required for array input')
    delta = 1.0 / sampling_rate
    padding_factor = padding_factor if (padding_factor and padding_factor >= 1) else 1
    duration = duration if (duration and duration > delta) else delta
    win_len = int(duration * sampling_rate)
    if win_len < 16:
        raise ValueError('Change duration to have at least 16 sample points!')
    overlap = overlap if overlap and (0. < overlap < 1.) else .9
    noverlap = int(win_len * overlap)
    nfft = int(win_len * padding_factor)
    win = scipy.signal.get_window(window, win_len) if isinstance(window, (str, tuple)) else np.asarray(window, float)
    if win.shape != (win_len,):
        raise ValueError('window must have length of nperseg')
    scale = 1.0 / (sampling_rate * (win * win).sum())
    x = np.asarray(x)
    hop = win_len - noverlap
    S = _ops.spectrogram_seg(None, x, win_len, hop, nfft, win, scale, detrend == 'constant')
    # scipy: freqs = rfftfreq(nfft, 1/fs); time = arange(nperseg/2, n - nperseg/2 + 1, hop) / fs
    f = np.fft.rfftfreq(nfft, 1 / sampling_rate)
    t = np.arange(win_len / 2, x.shape[-1] - win_len / 2 + 1, hop) / float(sampling_rate)
    return f, t, S.astype(x.dtype if x.dtype.kind == 'f' else np.float32)
