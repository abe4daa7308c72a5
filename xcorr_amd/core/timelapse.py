"""
The data path of xcorr/scripts/timelapse.py (BASELINE configs[4]) for ONE receiver pair:

    get_spectrogram          (:196-211)  post-processed CCF rows -> hop-1 spectrogram (2.5 s, padding 4)
    correlate_spectrogram    (:218-268)  per frequency band and per pair of times (t1 < t2):
                                         correlate2d of the two band tiles -> maximum, delta_freq, delta_lag

The CLI around it (file discovery, Dask fan-out, the NetCDF result container) stays with the
caller; this module takes arrays and returns arrays.
"""
import numpy as np

from .. import ops
from ..engine import default_context
from ..signal import spectral

__all__ = ['spectrograms', 'correlate_spectrograms']


def spectrograms(cc, sampling_rate, duration=2.5, padding_factor=4):
    """cc [T, L] (post-processed CCFs of T window times, core/postprocess.py chain with the
    timelapse settings of scripts/timelapse.py:198-205) -> (freq [nf], psd [T, nf, L])."""
    return spectral.spectrogram(np.ascontiguousarray(cc, np.float32), duration=duration,
                                padding_factor=padding_factor, sampling_rate=sampling_rate)


def correlate_spectrograms(cc, sampling_rate, freq, freq_bw, duration=2.5, padding_factor=4, pairs=None,
                           normalize=True, ctx=None):
    """For every band ``[f - bw/2, f + bw/2)`` (scripts/timelapse.py:243-249) and every pair of rows
    (t1 < t2, or `pairs` [P, 2]): the maximum of ``correlate2d(psd[t1, band], psd[t2, band])`` and where it
    sits (:252-267).  Returns a dict of arrays [nband, P]:

        cc2         the maximum (normalised by the Frobenius norms when `normalize`, signal/correlate.py:234-236)
        delta_freq  its frequency offset in Hz,  (i - (F-1)) * df
        delta_lag   its lag offset in seconds,   (j - (L-1)) / sampling_rate
        status      1 where a band holds at least one frequency bin, else 0
    plus 'pairs' [P, 2] and 'freq' of the spectrogram axis."""
    ctx = ctx or default_context()
    cc = np.ascontiguousarray(cc, np.float32)
    if cc.ndim != 2:
        raise ValueError('cc should be [times, lags]')
    T, L = cc.shape
    f, psd = spectrograms(cc, sampling_rate, duration, padding_factor)
    if pairs is None:
        pairs = np.array([(i, j) for i in range(T) for j in range(i + 1, T)], np.int32).reshape(-1, 2)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    freq = np.atleast_1d(np.asarray(freq, np.float64))
    bw = np.broadcast_to(np.asarray(freq_bw, np.float64), freq.shape)
    nb, P = len(freq), len(pairs)
    out = {'cc2': np.zeros((nb, P), np.float32), 'delta_freq': np.zeros((nb, P)), 'delta_lag': np.zeros((nb, P)),
           'status': np.zeros((nb, P), np.byte), 'pairs': pairs, 'freq': f}
    df = f[1] - f[0]
    for b in range(nb):
        sel = (f >= freq[b] - bw[b] / 2) & (f < freq[b] + bw[b] / 2)
        F = int(sel.sum())
        if F == 0 or P == 0:
            continue
        tiles = np.ascontiguousarray(psd[:, sel, :])
        peak, idx = ops.correlate2d(ctx, tiles, pairs, normalize=normalize)
        out['cc2'][b] = peak
        out['delta_freq'][b] = (idx[:, 0] - (F - 1)) * df
        out['delta_lag'][b] = (idx[:, 1] - (L - 1)) / float(sampling_rate)
        out['status'][b] = 1
    return out
