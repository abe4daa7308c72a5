"""Timing of the post-correlation chain (core/postprocess.py:133-141) on full-lag CCF rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from xcorr_amd import ops
from xcorr_amd.engine import default_context
ctx = default_context()
rows, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2016, 180000
L = 2 * n - 1
x = ctx.array((np.random.default_rng(0).standard_normal((rows, L)) * 0.01).astype(np.float32))
sos = scipy.signal.butter(2, 3.0, 'highpass', output='sos', fs=50.)
delay = np.random.default_rng(1).uniform(-0.02, 0.02, rows)
y = ctx.empty((rows, L - 1))
def t(name, fn, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:12s} {dt*1e3:8.2f} ms  {dt/rows*1e6:7.2f} us/row', flush=True)
t('unbias', lambda: ops.unbias(ctx, x, n))
t('demean', lambda: ops.demean(ctx, x))
t('taper', lambda: ops.taper(ctx, x, 250))
t('timeshift', lambda: ops.timeshift(ctx, x, delay, 50., out=y))
delay5 = np.array([0.0, 0.0049, -0.0131, 0.02, 0.005])[np.random.default_rng(2).integers(0, 5, rows)]
t('timeshift/5', lambda: ops.timeshift(ctx, x, delay5, 50., out=y))      # five distinct delays: kernels shared
t('timeshift/1', lambda: ops.timeshift(ctx, x, 0.0049, 50., out=y))      # scalar delay
t('sosfiltfilt', lambda: ops.sosfiltfilt(ctx, y, sos))
t('taper2', lambda: ops.taper(ctx, y, 75))
x2 = ctx.array((np.random.default_rng(0).standard_normal((rows, L)) * 0.01).astype(np.float32))
t('post_prep', lambda: ops.post_prep(ctx, x2, n, 0, L, unbiased=True, taper_nmax=250))
from xcorr_amd.core.postprocess import postprocess_rows
def chain():
    r = postprocess_rows(x2, delay, 50., n, 0, L, ctx=ctx)
    r.free()
t('whole chain', chain, reps=2)
