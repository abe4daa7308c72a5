"""
xcorr_amd.signal -- the starred functions of ``xcorr/signal`` on the GPU.

Every function takes either a float32 numpy array with the core dimension
last (what the reference's ``xr.apply_ufunc`` closures receive) plus the
metadata the reference reads from coordinate attrs (``sampling_rate``,
``npts``, ``index_min`` ...) as keyword arguments, or -- when xarray is
importable -- an ``xarray.DataArray`` carrying that metadata, in which case
the reference's signature ``f(x, ..., dim=None)`` applies and a DataArray is
returned.  (xarray is not installed in the build container: the DataArray
branch is exercised only where xarray exists.)
"""
from .ops1d import (correlate1d, norm1d, norm, demean, detrend, window, taper,
                    filter, unbias, unbias_weights, timeshift, snr, hilbert,
                    envelope, rms, absolute)  # noqa: F401
from .ops2d import norm2d, correlate2d, correlate2d_peak  # noqa: F401
from .fftops import fft, ifft, rfft, irfft  # noqa: F401
from .spectral import spectrogram, psd, spectrogram_mtc  # noqa: F401

__all__ = ['correlate1d', 'norm1d', 'norm', 'demean', 'detrend', 'window',
           'taper', 'filter', 'unbias', 'unbias_weights', 'timeshift', 'snr',
           'norm2d', 'correlate2d', 'correlate2d_peak', 'fft', 'ifft', 'rfft', 'irfft', 'hilbert', 'envelope', 'spectrogram', 'psd', 'spectrogram_mtc', 'rms', 'absolute']

try:                                    # B4 seam: DataArray.signal accessor
    from . import accessor  # noqa: F401
except ModuleNotFoundError:             # xarray absent
    accessor = None
