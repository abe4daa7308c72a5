"""Post-correlation chain on the PRODUCTION lag axis (clip_lag = (0, 9 h) at 50 Hz: 1 620 000 lags per row,
scripts/cc.py:25-29): head, time shift (block-decomposed Dirichlet convolution), zero-phase filter, device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from xcorr_amd import ops
from xcorr_amd.engine import default_context
ctx = default_context()
rows, n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4320000, 1620000
imin = n - 1
x = ctx.array((np.random.default_rng(0).standard_normal((rows, L)) * 0.01).astype(np.float32))
sos = scipy.signal.butter(2, 3.0, 'highpass', output='sos', fs=50.)
delay = np.random.default_rng(1).uniform(-0.02, 0.02, rows)
delay5 = np.array([0.0, 0.0049, -0.0131, 0.02, 0.005])[np.random.default_rng(2).integers(0, 5, rows)]
y = ctx.empty((rows, L - (L & 1)))
def t(name, fn, reps=2):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:14s} {dt*1e3:9.2f} ms  {dt/rows*1e6:9.1f} us/row', flush=True)
t('post_prep', lambda: ops.post_prep(ctx, x, n, imin, imin + L, unbiased=True, taper_nmax=250))
t('timeshift', lambda: ops.timeshift(ctx, x, delay, 50., out=y))
t('timeshift/5', lambda: ops.timeshift(ctx, x, delay5, 50., out=y))
t('sosfiltfilt', lambda: ops.sosfiltfilt(ctx, y, sos))
