"""Randomised parity sweep of the pairs engine against the oracle (GPU box):
random n, hop, lag slices, flags, station counts and pair subsets, stack on/off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
from xcorr_amd.engine import default_context, CCPlan



def run(budget=60.0, seed=0, ctx=None):
  rng = np.random.default_rng(seed)
  ctx = ctx or default_context()
  t_end = time.time() + budget
  worst, cases = 0.0, 0
  while time.time() < t_end:
    n = int(rng.choice([rng.integers(2, 64), rng.integers(64, 3000), rng.integers(3000, 40000)]))
    W = int(rng.integers(1, 5))
    hop = int(rng.integers(1, n + 1))
    S = int(rng.integers(2, 9))
    T = (W - 1) * hop + n + int(rng.integers(0, 5))
    full = 2 * n - 1
    if rng.random() < 0.5:
        imin, imax = 0, full
    else:
        imin = int(rng.integers(0, full)); imax = int(rng.integers(imin + 1, full + 1))
    flags = dict(normalize=bool(rng.random() < 0.7), demean=bool(rng.random() < 0.4), onebit=bool(rng.random() < 0.3))
    stack = bool(rng.random() < 0.4)
    allp = [(i, j) for i in range(S) for j in range(S)]
    k = int(rng.integers(1, len(allp) + 1))
    pairs = np.array([allp[i] for i in rng.choice(len(allp), k, replace=False)], np.int32)
    tr = (rng.standard_normal((S, T)) + rng.uniform(-1, 1)).astype(np.float32)
    plan = CCPlan(ctx, n, imin, imax, stack=stack, **flags)
    if os.environ.get('FUZZ_VERBOSE'):
        print(cases, dict(n=n, W=W, hop=hop, S=S, imin=imin, imax=imax, stack=stack, P=len(pairs), N1=plan.N1, N2=plan.N2, batch=plan.batch, **flags), flush=True)
    got = plan.pairs(tr, pairs, W, hop)
    ref = R.cc_pairs(tr, pairs, n, hop, normalize=flags['normalize'], index_min=imin, index_max=imax,
                     onebit_=flags['onebit'], demean_=flags['demean'], dtype=np.float64)[:, :W]
    if stack:
        ref = ref.sum(axis=1)
    scale = np.abs(R.cc_pairs(tr, pairs, n, hop, normalize=flags['normalize'], onebit_=flags['onebit'],
                              demean_=flags['demean'], dtype=np.float64)[:, :W]).max()
    if stack:
        scale *= W
    err = float(np.abs(got - ref).max() / max(scale, 1e-30))
    worst = max(worst, err); cases += 1
    tol = 1e-5          # one-bit after demean decides its signs in float64 on the GPU, like the oracle
    if not (err < tol):
        raise AssertionError(f'{dict(n=n, W=W, hop=hop, S=S, imin=imin, imax=imax, stack=stack, P=len(pairs), **flags)}: {err}')
    plan.close()
  return cases, worst


if __name__ == '__main__':
    c, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz: {c} cases, worst relative error {w:.2e}')
