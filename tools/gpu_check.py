"""Dev aid: quick numeric check of the HIP path against the oracle on a GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import ref_numpy as R
import xcorr_amd as X

ctx = X.Context(0)
print('device:', ctx.name, flush=True)
rng = np.random.default_rng(0)
ok = True
for n in [1, 2, 8, 100, 257, 1000, 4096, 20000, 180000]:
    for norm in (True, False):
        x = rng.standard_normal((3, n)).astype(np.float32)
        y = (0.5 * np.roll(x, n // 5, axis=-1) + rng.standard_normal((3, n))).astype(np.float32)
        plan = X.CCPlan(ctx, n, normalize=norm)
        t0 = time.time()
        got = plan.cc(x, y)
        dt = time.time() - t0
        ref = R.cc_fair(x, y, normalize=norm, dtype=np.float64)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        flag = 'OK ' if err < 1e-5 else 'BAD'
        ok &= err < 1e-5
        print(f'{flag} n={n:7d} N={plan.N:7d} ({plan.N1}x{plan.N2}) norm={norm} relerr={err:.3e} t={dt*1e3:.1f}ms', flush=True)
        plan.close()

# pairs engine
S, n, hop, W = 5, 3000, 1500, 4
T = hop * (W - 1) + n
tr = R.synthetic_traces(S, T)
pairs = R.all_pairs(S)
for kw in [dict(), dict(demean=True), dict(onebit=True, demean=True), dict(index_min=1000, index_max=4500), dict(unbiased=True)]:
    okw = dict(kw)
    ub = okw.pop('unbiased', False)
    plan = X.CCPlan(ctx, n, **kw)
    got = plan.pairs(tr, pairs, W, hop)
    ref = R.cc_pairs(tr, pairs, n, hop, normalize=True, dtype=np.float64,
                     demean_=okw.get('demean', False), onebit_=okw.get('onebit', False),
                     index_min=okw.get('index_min', 0), index_max=okw.get('index_max'))
    if ub:
        ref = R.unbias(ref, n)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    flag = 'OK ' if err < 1e-5 else 'BAD'
    ok &= err < 1e-5
    print(f'{flag} pairs {kw} shape={got.shape} relerr={err:.3e}', flush=True)
    plan.close()

# stack
plan = X.CCPlan(ctx, n, stack=True)
got = plan.pairs(tr, pairs, W, hop)
ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64).sum(axis=1)
err = np.abs(got - ref).max() / np.abs(ref).max()
print(('OK ' if err < 1e-5 else 'BAD'), 'stack relerr', err, flush=True)
ok &= err < 1e-5
print('ALL OK' if ok else 'FAILURES')
sys.exit(0 if ok else 1)
