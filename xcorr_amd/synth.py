"""
Synthetic station records of SURVEY 8d (bench / examples): a common signal
with 0.5 s moveout per station at 50 Hz plus independent noise, an offset and
a linear trend, so that demean/detrend are observable and every pair (i, j)
peaks at lag 25*(i-j) samples.  float32 [S][T].
"""
import numpy as np

__all__ = ['synthetic_traces', 'fast_day']


def synthetic_traces(S, T, day=0, dtype=np.float32):
    c = np.random.default_rng(day * 10000).standard_normal(T + 25 * S)
    t = np.arange(T, dtype=np.float64)
    out = np.empty((S, T), dtype=dtype)
    for s in range(S):
        nse = np.random.default_rng(day * 10000 + 1000 + s).standard_normal(T)
        off = 25 * (S - 1 - s)
        out[s] = (0.5 * c[off:off + T] + nse + (3.0 + 1e-6 * t)).astype(dtype)
    return out


def fast_day(S, T, day, offset=3.0):
    """Cheap synthetic station-day for multi-day runs (float32 generator: 30 days x 64 stations in seconds
    instead of minutes; same structure as :func:`synthetic_traces` -- common signal with 25-sample moveout
    per station + independent noise + an offset -- without the linear trend).  The generator of
    ``bench.py --stack --days D`` and of ``tests/test_gpu_config4.py``."""
    rng = np.random.default_rng(777 + day)
    c = rng.standard_normal(T + 25 * S, dtype=np.float32)
    out = rng.standard_normal((S, T), dtype=np.float32)
    for s in range(S):
        off = 25 * (S - 1 - s)
        out[s] += 0.5 * c[off:off + T]
        if offset:
            out[s] += np.float32(offset)
    return out
