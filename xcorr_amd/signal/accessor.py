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

    @wraps(_fs.spectrogram_mtc)
    def spectrogram_mtc(self, **kwargs):
        return _fs.spectrogram_mtc(self._da, **kwargs)

    @wraps(_f.rms)
    def rms(self, **kwargs):
        return _f.rms(self._da, **kwargs)

    def abs(self):
        return _f.absolute(self._da)

    absolute = abs

    @wraps(_f.snr)
    def snr(self, signal, noise, **kwargs):
        return _f.snr(self._da, signal, noise, **kwargs)

    def unbias_weights(self, name=None):
        """Unbias weights of the DataArray's lag coordinate (signal/unbias.py:106-145): the
        coordinate attrs ``npts, index_min, index_max`` select ``weight(npts)[index_min:index_max]``."""
        la = self._da.attrs if 'npts' in self._da.attrs else self._da[self._da.dims[-1]].attrs
        for attr in ('npts', 'index_min', 'index_max'):
            if attr not in la:
                raise ValueError(f'Lag has no attribute "{attr}"!')
        w = _f.unbias_weights(la['npts'], la['index_min'], la['index_max'])
        dim = self._da.dims[-1]
        return xr.DataArray(w.astype(self._da.dtype), dims=(dim,), coords={dim: self._da[dim]}, name=name or 'w',
                            attrs={'long_name': 'Unbiased CC estimate scale factor', 'units': '-'})

    def window(self, **kwargs):
        """Taper window along the DataArray's (1-D) coordinate (signal/window.py:27-116)."""
        coord = self._da
        if 'sampling_rate' not in coord.attrs:
            raise ValueError('x has no attribute "sampling_rate"!')
        w = _f.window(coord.size, coord.attrs['sampling_rate'], **kwargs)
        return xr.DataArray(w, dims=coord.dims, coords=coord.coords, name='win',
                            attrs={'long_name': 'Window', 'units': '-'})
