"""``DataArray.signal`` accessor for the GPU-backed functions (B4 seam,
xcorr/signal/accessor.py:52-327).  Imported only when xarray is available."""
from functools import wraps

import xarray as xr

from . import ops1d as _f
from . import ops2d as _f2
from . import fftops as _ff
from . import spectral as _fs


@xr.register_dataarray_accessor('signal')
class SignalAccessor():
    def __init__(self, dataarray: xr.DataArray):
        self._da = dataarray

    @wraps(_f.correlate1d)
    def correlate1d(self, in2, **kwargs):
        return _f.correlate1d(self._da, in2, **kwargs)

    @wraps(_f.norm1d)
    def norm1d(self, **kwargs):
        return _f.norm1d(self._da, **kwargs)

    norm = norm1d

    @wraps(_f.detrend)
    def detrend(self, **kwargs):
        return _f.detrend(self._da, **kwargs)

    @wraps(_f.demean)
    def demean(self, **kwargs):
        return _f.demean(self._da, **kwargs)

    @wraps(_f.taper)
    def taper(self, **kwargs):
        return _f.taper(self._da, **kwargs)

    @wraps(_f.filter)
    def filter(self, *args, **kwargs):
        return _f.filter(self._da, *args, **kwargs)

    @wraps(_f.unbias)
    def unbias(self, **kwargs):
        return _f.unbias(self._da, **kwargs)

    @wraps(_f.timeshift)
    def timeshift(self, **kwargs):
        return _f.timeshift(self._da, **kwargs)

    @wraps(_f.hilbert)
    def hilbert(self, **kwargs):
        return _f.hilbert(self._da, **kwargs)

    @wraps(_f.envelope)
    def envelope(self, **kwargs):
        return _f.envelope(self._da, **kwargs)

    @wraps(_f2.norm2d)
    def norm2d(self, **kwargs):
        return _f2.norm2d(self._da, **kwargs)

    @wraps(_f2.correlate2d)
    def correlate2d(self, in2, **kwargs):
        return _f2.correlate2d(self._da, in2, **kwargs)

    @wraps(_ff.fft)
    def fft(self, **kwargs):
        return _ff.fft(self._da, **kwargs)

    @wraps(_ff.ifft)
    def ifft(self, **kwargs):
        return _ff.ifft(self._da, **kwargs)

    @wraps(_ff.rfft)
    def rfft(self, **kwargs):
        return _ff.rfft(self._da, **kwargs)

    @wraps(_ff.irfft)
    def irfft(self, **kwargs):
        return _ff.irfft(self._da, **kwargs)

    @wraps(_fs.spectrogram)
    def spectrogram(self, **kwargs):
        return _fs.spectrogram(self._da, **kwargs)

    psd = spectrogram

    @wraps(_f.rms)
    def rms(self, **kwargs):
        return _f.rms(self._da, **kwargs)

    def abs(self):
        return _f.absolute(self._da)
