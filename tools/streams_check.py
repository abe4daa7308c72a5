"""Does the hot path slow down when other plans of the same context have created their own pipeline streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xcorr_amd as X
from xcorr_amd import ops, workload
from xcorr_amd.synth import fast_day
order = sys.argv[1] if len(sys.argv) > 1 else 'after'
ctx = X.Context(0)
def others():
    x = np.random.default_rng(0).standard_normal((64, 359998)).astype(np.float32)
    ops.rfft(ctx, x); ops.hilbert(ctx, x)
    y = ops.timeshift(ctx, x, np.full(64, 0.01), 50.)
    z = np.random.default_rng(0).standard_normal((64, 20001)).astype(np.float32)
    ops.rfft(ctx, z)
    ctx.sync()
if order == 'before':
    others()
S, T, n = 64, 4320000, 180000
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
sd = workload.StationDay(ctx, S, T, n, n // 2, pairs)
sd.load(fast_day(S, T, 0))
def timeit(tag):
    sd.step(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): sd.step()
    ctx.sync()
    print(tag, round((time.perf_counter() - t0) / 3 * 1e3, 2), 'ms/step', flush=True)
timeit('hot path (%s other plans exist)' % ('after' if order == 'before' else 'before any'))
if order != 'before':
    others()
    timeit('hot path after other plans ran')
