"""Per-call latency of the single-vector entry points (seam B1: util.cc.cc per pair and window)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd.util import cc as ucc
from xcorr_amd import signal
rng = np.random.default_rng(0)
n = 180000
x = rng.standard_normal(n).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
def t(name, fn, reps=20):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    print(f'{name:34s} {1e3*(time.perf_counter()-t0)/reps:8.3f} ms per call', flush=True)
t('util.cc.cc(x, y) n=180000', lambda: ucc.cc(x, y, normalize=True, pad=True, unbiased=False, dtype=np.float32))
X = rng.standard_normal((47, n)).astype(np.float32); Y = rng.standard_normal((47, n)).astype(np.float32)
t('signal.correlate1d [47, n]', lambda: signal.correlate1d(X, Y), reps=5)
c = rng.standard_normal((47, 2 * n - 1)).astype(np.float32)
t('signal.filter [47, 359999]', lambda: signal.filter(c, 3., 'highpass', 2, sampling_rate=50.), reps=5)
t('signal.timeshift [47, 359999]', lambda: signal.timeshift(c, 0.01, sampling_rate=50.), reps=3)
t('signal.detrend+taper [47, 359999]', lambda: signal.taper(signal.detrend(c), max_length=5., sampling_rate=50.), reps=5)
