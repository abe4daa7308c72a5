"""
xcorr_amd.util.cc -- drop-in for ``xcorr.util.cc`` (xcorr/util/cc.py).

``cc`` keeps the reference signature and error behaviour (util/cc.py:22-79)
but runs on the GPU through libxcx.so: zero-pad, FFT, conj-multiply, L2
normalisation, inverse FFT and lag rotation are one fused HIP pipeline at a
fast transform length N >= 2n-1 (identical linear cross-correlation).
``lag``/``weight`` and the arg-max helpers are coordinate bookkeeping, kept
on the host in numpy like the reference.
"""
import numpy as np

from ..engine import CCPlan, default_context

__all__ = ['cc', 'lag', 'weight', 'extract_shift', 'extract_shift_and_max',
           'compute_shift', 'compute_shift_and_max']

_plans = {}


def _plan(n, normalize, unbiased):
    key = (n, bool(normalize), bool(unbiased))
    if key not in _plans:
        if len(_plans) > 16:
            _plans.pop(next(iter(_plans))).close()
        _plans[key] = CCPlan(default_context(), n, normalize=normalize,
                             unbiased=unbiased)
    return _plans[key]


def cc(x, y, normalize=True, pad=True, unbiased=True, dtype=None):
    """Cross-correlate two vectors `x` and `y` (util/cc.py:22-79).

    Same arguments as the reference.  Differences, all documented in
    DESIGN.md: the unbiased weights are the closed form over all 2n-1 lags
    (the reference's ``weight`` is two entries short and raises);
    arithmetic is float32 on the GPU (<= 1e-5 of max|cc| from the reference).
    """
    x = np.asarray(x)
    y = np.asarray(y)
    n = len(x)
    dtype = dtype or x.dtype
    assert n == len(y), "Vectors `x` and `y` should have the same length!"
    assert x.dtype == y.dtype, (
        "Vectors `x` and `y` should have the same dtype!"
    )
    plan = _plan(n, normalize, unbiased and pad)
    r = plan.cc(np.ascontiguousarray(x, np.float32)[None],
                np.ascontiguousarray(y, np.float32)[None])[0]
    if not pad:
        # circular correlation of length n from the linear one:
        # c[k] = R[k] + R[k-n]; then fftshift like util/cc.py:78
        lin = r.astype(np.float64)
        c = lin[n - 1:].copy()                 # lags 0..n-1
        c[1:] += lin[:n - 1]                   # lags -(n-1)..-1 -> k-n
        r = np.fft.fftshift(c)
        if unbiased:
            r = r * weight(n, False)
    return r.astype(dtype)


def lag(n, delta, pad=True):
    """Lag vector in seconds (util/cc.py:82-108)."""
    nn = n * 2 - 1 if pad else n
    return np.fft.fftshift(np.fft.fftfreq(nn, 1 / (nn * delta)))


def weight(n, pad=True, clip=None):
    """Unbiased-estimate weights n/(n-|k|) for every lag of ``cc(pad=pad)``
    (util/cc.py:111-141; closed form over all lags, see module docstring)."""
    if pad:
        k = np.arange(-(n - 1), n, dtype=np.float64)
        w = n / (n - np.abs(k))
    else:
        nn = int((n + 1) / 2)
        k = np.fft.fftshift(np.fft.fftfreq(n, 1.0 / n))
        w = nn / (nn - np.abs(k))
    if clip is not None:
        w[np.where(w > clip)] = clip
    return w


def extract_shift(Rxy, delta=None):
    """Sample (or time) shift at the maximum (util/cc.py:144-172)."""
    zero_index = int(len(Rxy) / 2)
    shift = np.argmax(Rxy) - zero_index
    return shift * (delta or 1)


def extract_shift_and_max(Rxy, delta=None):
    """Shift and maximum (util/cc.py:175-206)."""
    zero_index = int(len(Rxy) / 2)
    index_max = np.argmax(Rxy)
    shift = index_max - zero_index
    return shift * (delta or 1), Rxy[index_max]


def compute_shift(x, y, delta=None, **kwargs):
    """util/cc.py:209-243."""
    Rxy = cc(x, y, **kwargs)
    return extract_shift_and_max(Rxy, delta)[0]


def compute_shift_and_max(x, y, delta=None, **kwargs):
    """util/cc.py:246-280."""
    Rxy = cc(x, y, **kwargs)
    return extract_shift_and_max(Rxy, delta)
