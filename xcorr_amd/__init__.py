"""
xcorr_amd -- MI355X-native drop-in for xcorr's cross-correlation hot path.

Python host code over a C-ABI HIP library (``xcorr_amd/lib/libxcx.so``,
``include/xcx.h``).  The package mirrors the reference's interface for this
path only: ``xcorr_amd.util.cc`` (xcorr/util/cc.py), ``xcorr_amd.signal``
(the starred functions of xcorr/signal/) and ``xcorr_amd.core.process``
(the pair x window driver of xcorr/core/process.py).
"""
from . import _ffi                      # noqa: F401
from .engine import Context, DeviceArray, CCPlan, default_context  # noqa: F401

__all__ = ['Context', 'DeviceArray', 'CCPlan', 'default_context']
__version__ = '0.1.0'
