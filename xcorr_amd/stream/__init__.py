"""
xcorr_amd.stream -- GPU counterparts of the preprocess operations that
``xcorr/stream/process.py:29-94`` dispatches to ObsPy, on plain arrays
(``[..., npts]`` float32 + sampling rate).  They can be injected as callables
into the reference's operation lists (seam B5, ``stream/process.py:264-266``).

ObsPy is not vendored by the reference and not installable here: the ObsPy-backed
ops are restated from ObsPy's documented behaviour and are **unverified against
obspy itself** (SURVEY 8c).  ``running_rms`` is the reference's own code.
"""
from .ops import (detrend, filter, taper, decimate, resample, interpolate, running_rms,   # noqa: F401
                  butter_sos, cheby2_lowpass_sos, cosine_taper_length, apply_operations)
from .segments import merge, trim, remove_sensitivity, rotate_ne_rt, normalize                       # noqa: F401

__all__ = ['merge', 'trim', 'remove_sensitivity', 'rotate_ne_rt', 'normalize', 'detrend', 'filter', 'taper', 'decimate', 'resample', 'interpolate', 'running_rms',
           'butter_sos', 'cheby2_lowpass_sos', 'cosine_taper_length', 'apply_operations']
