import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
import xcorr_amd as X
ctx = X.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4320000
rng = np.random.default_rng(2024)
c = rng.standard_normal(n + 30000).astype(np.float32)
x = (c[30000:] + 0.5 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
y = (c[:n] + 0.5 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
K = min(1620000, n - 1)
imin, imax = n - 1, n - 1 + K
plan = X.CCPlan(ctx, n, imin, imax, normalize=True)
got = plan.cc(x[None], y[None])[0].astype(np.float64)
ref = R.cc_fair(x, y, normalize=True, dtype=np.float64)[imin:imax]
d = np.abs(got - ref)
i = int(np.argmax(d))
print('N', plan.N, 'max abs diff', d.max(), 'at lag', i, 'got', got[i], 'ref', ref[i], 'max ref', np.abs(ref).max(), 'argmax ref', int(np.argmax(ref)))
print('median abs diff', np.median(d), 'p99', np.percentile(d, 99), 'count > 1e-6:', int((d > 1e-6).sum()))
bad = np.flatnonzero(d > 1e-6)
print('bad lags (first 20):', bad[:20], ' ... last', bad[-5:] if len(bad) else None)
Kb = plan.N // 2
print('bad lags mod Kb:', np.unique(bad % Kb)[:20], 'chunks', np.unique(bad // Kb))
