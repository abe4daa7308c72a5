"""
Skip-if-done resume of the production loop (xcorr/core/lazy.py:26-82) without Dask: one call per
(pair, day).  The reference wraps this in ``dask.delayed`` only to fan it out over CPU workers;
here the fan-out is the device batch inside ``core.process``.
"""
import numpy as np

from .. import io
from .init import init
from .process import process

__all__ = ['single_process', 'lazy_process']


def single_process(pair, time, init_args, client, root, force_fresh=False, step=86400., verb=0, **kwargs):
    """Process one (pair, day) file: open ``io.ncfile(pair, time, root)``; if every step is already
    processed return True without touching the GPU (core/lazy.py:52-58); otherwise initialise a new
    container or resume the partial one with ``retry_missing=True`` (:61-78), write it when it holds
    any data (:80-83).  `time`: POSIX seconds / timestamp of the day start; `step`: file span (s)."""
    nc = io.ncfile(pair, time, root)
    data = None
    if not force_fresh:
        data = io.read(nc)
        if data is not None and np.all(data.status == 1):
            return True
    if data is None:
        from .init import _seconds
        t0, _ = _seconds(time)
        data = init(pair=pair, starttime=time, endtime=t0 + step, **init_args)
    process(data, client=client, retry_missing=True, verb=verb, **kwargs)
    if np.any(data.status == 1):
        io.write(data, nc, verb=verb)
        return True
    return False


def lazy_process(pairs, times, init_args, client, root, **kwargs):
    """All (pair, day) combinations in order (core/lazy.py:85-160 builds the same list as delayed
    tasks); returns the number of files that hold processed data."""
    return sum(bool(single_process(p, t, init_args, client, root, **kwargs)) for p in pairs for t in times)
