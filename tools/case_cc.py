"""One parity case of the pairs engine against the oracle (debug aid).
   python tools/case_cc.py n W hop S imin imax stack normalize demean onebit P [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
from xcorr_amd.engine import default_context, CCPlan
a = sys.argv[1:]
n, W, hop, S, imin, imax = [int(v) for v in a[:6]]
stack, normalize, demean, onebit = [bool(int(v)) for v in a[6:10]]
P = int(a[10]); seed = int(a[11]) if len(a) > 11 else 0
rng = np.random.default_rng(seed)
T = (W - 1) * hop + n + 3
allp = [(i, j) for i in range(S) for j in range(S)]
pairs = np.array([allp[i] for i in rng.choice(len(allp), P, replace=False)], np.int32)
tr = (rng.standard_normal((S, T)) + 0.3).astype(np.float32)
ctx = default_context()
plan = CCPlan(ctx, n, imin, imax, stack=stack, normalize=normalize, demean=demean, onebit=onebit)
got = plan.pairs(tr, pairs, W, hop)
ref = R.cc_pairs(tr, pairs, n, hop, normalize=normalize, index_min=imin, index_max=imax, onebit_=onebit, demean_=demean,
                 dtype=np.float64)[:, :W]
full = R.cc_pairs(tr, pairs, n, hop, normalize=normalize, onebit_=onebit, demean_=demean, dtype=np.float64)[:, :W]
scale = np.abs(full).max() * (W if stack else 1)
if stack:
    ref = ref.sum(axis=1)
d = np.abs(got - ref) / scale
print(f'N={plan.N} N1={plan.N1} N2={plan.N2} L={plan.L}: max err {d.max():.3e}; rows worst {np.argsort(d.reshape(d.shape[0], -1).max(-1))[-3:]}; '
      f'bad elements {(d > 1e-5).sum()} of {d.size}; first bad lags {np.argwhere(d > 1e-5)[:6].tolist()}')
