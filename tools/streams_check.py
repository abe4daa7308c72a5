"""Does the hot path slow down when other plans of the same context were created first (time shift, any-length
transforms)?  The pair phase overlaps two launch pipelines; they need hardware queues of their own.
usage: streams_check.py fresh|before|after [stations]     (fresh process each; prints `ms_per_step <x>`)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xcorr_amd as X
from xcorr_amd import ops, workload
from xcorr_amd.synth import fast_day
order = sys.argv[1] if len(sys.argv) > 1 else 'fresh'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = X.Context(0)


def others():
    x = np.random.default_rng(0).standard_normal((8, 359998)).astype(np.float32)
    ops.rfft(ctx, x); ops.hilbert(ctx, x)                 # chirp-convolution plans (stacking pair pipelines)
    ops.timeshift(ctx, x, np.full(8, 0.01), 50.)          # time-shift plan
    ops.rfft(ctx, x[:, :20002])
    ctx.sync()


if order == 'before':
    others()
T, n = 4320000, 180000
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
sd = workload.StationDay(ctx, S, T, n, n // 2, pairs)
sd.load(fast_day(S, T, 0))


def timeit():
    sd.step(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): sd.step()
    ctx.sync()
    return (time.perf_counter() - t0) / 3 * 1e3


if order == 'after':
    print('fresh', round(timeit(), 2))
    others()
print('ms_per_step', round(timeit(), 2), flush=True)
