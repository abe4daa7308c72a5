"""Time xcx_cc2d on BASELINE config 5 (256 x 4096 tiles): peak-only, resident tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd import ops
from xcorr_amd.engine import default_context

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
F, L = 256, 4096
ctx = default_context()
rng = np.random.default_rng(0)
t = np.abs(rng.standard_normal((T, F, L))).astype(np.float32)
d = ctx.array(t)
pairs = [(i, j) for i in range(T) for j in range(i + 1, T)]
ops.correlate2d(ctx, d, pairs[:4])
for rep in range(3):
    t0 = time.perf_counter()
    peak, idx = ops.correlate2d(ctx, d, pairs)
    dt = time.perf_counter() - t0
    print(f'T={T} pairs={len(pairs)}: {dt*1e3:.1f} ms  -> {len(pairs)/dt:.0f} tile-pairs/s  ({dt/len(pairs)*1e6:.0f} us/pair)', flush=True)
if '--cpu' in sys.argv:
    from oracle import ref_numpy as R
    t0 = time.perf_counter()
    ref = R.correlate2d(t[0], t[1])
    print(f'numpy float64 one pair: {time.perf_counter()-t0:.2f} s')
