"""Timing of the per-day front end of bench.py (copy, demean, taper, sosfilt) and of the forward transforms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from xcorr_amd import ops
from xcorr_amd.engine import default_context, CCPlan
ctx = default_context()
S, T, n, hop = 64, 4320000, 180000, 90000
raw = ctx.array(np.random.default_rng(0).standard_normal((S, T)).astype(np.float32))
work = ctx.empty((S, T))
sos = scipy.signal.butter(4, [0.1, 10.0], btype='bandpass', output='sos', fs=50.)
def t(name, fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    print(f'{name:10s} {(time.perf_counter()-t0)/reps*1e3:7.3f} ms', flush=True)
t('copy', lambda: ops.copy(ctx, work, raw))
t('demean', lambda: ops.demean(ctx, work))
t('taper', lambda: ops.taper(ctx, work, 1500, wtype='cosine'))
t('sosfilt', lambda: ops.sosfilt(ctx, work, sos))
t('ram', lambda: ops.ram_normalize(ctx, work, 250))
t('decimate5', lambda: ops.decimate(ctx, work, 5, sos))
# ---- bench.py --shard windows: the front end of ONE rank's window block ----------------
from xcorr_amd import workload
W_all = (T - n) // hop + 1
pairs = np.array([(0, 1)], np.int32)
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        w0, w1 = (W_all * rank) // world, (W_all * (rank + 1)) // world
        sd = workload.StationDay(ctx, S, T, n, hop, pairs, w_first=w0, W=w1 - w0)
        sd.raw, sd.work = raw, work
        sd.view = work
        sd.front(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5): sd.front()
        ctx.sync()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        worst = max(worst, ms)
        print(f'front end of rank {rank} of {world} (windows {w0}..{w1 - 1}): {ms:7.3f} ms', flush=True)
        sd.raw = sd.work = None
        sd.plan.close(); sd.out.free()
    print(f'  -> slowest rank of {world}: {worst:.3f} ms', flush=True)
