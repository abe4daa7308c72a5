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

__all__ = ['Context', 'DeviceArray', 'CCPlan', 'default_context',
           # the reference's top-level names for this path (xcorr/__init__.py:34-46)
           'signal', 'stream', 'util', 'io', 'init', 'process', 'postprocess', 'read', 'write', 'merge', 'mfread']
__version__ = '0.2.0'

_LAZY = {'init': ('core', 'init'), 'process': ('core', 'process'), 'postprocess': ('core', 'postprocess'),
         'read': ('io', 'read'), 'write': ('io', 'write'), 'merge': ('io', 'merge'), 'mfread': ('io', 'mfread')}


def __getattr__(name):
    """``xcorr_amd.init`` / ``.process`` / ``.read`` ... like ``xcorr.init`` etc., imported on first use."""
    import importlib
    if name in ('signal', 'stream', 'util', 'io', 'core', 'ops'):
        return importlib.import_module('.' + name, __name__)
    if name in _LAZY:
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module('.' + mod, __name__), attr)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
