"""Timing of the reference's production window (scripts/cc.py:25-29: window_length = 86400 s @ 50 Hz, clip_lag = (0, 9 h))
on device-resident rows: n = 4 320 000, 1 620 000 retained lags, segmented plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xcorr_amd as X
from xcorr_amd.engine import default_context
ctx = default_context()
n, fs = 4320000, 50.
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
x = ctx.array(rng.standard_normal((rows, n), dtype=np.float32))
y = ctx.array(rng.standard_normal((rows, n), dtype=np.float32))
imin, imax = n - 1, n - 1 + int(9 * 3600 * fs)
plan = X.CCPlan(ctx, n, imin, imax, normalize=True, demean=True)
out = ctx.empty((rows, plan.L))
def run():
    plan.cc(x, y, out=out)
run(); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): run()
ctx.sync()
dt = (time.perf_counter() - t0) / 3
print(f'production day window: {rows} pair-days in {dt*1e3:.2f} ms -> {dt/rows*1e3:.3f} ms per pair-day '
      f'({rows * plan.L / dt / 1e9:.2f} G lag samples/s)')

# all pairs of S stations for one day-long window: rows that share a station share its segment transforms
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tr = ctx.array(rng.standard_normal((S, n), dtype=np.float32))
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
outp = ctx.empty((len(pairs), 1, plan.L))
def runp():
    plan.pairs(tr, pairs, 1, n, out=outp)
runp(); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): runp()
ctx.sync()
dt = (time.perf_counter() - t0) / 3
print(f'production day window, all {len(pairs)} pairs of {S} stations: {dt*1e3:.2f} ms -> {dt/len(pairs)*1e3:.3f} ms per pair-day')
