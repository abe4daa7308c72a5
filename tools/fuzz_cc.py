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
    # random whitening band (band-limited spectrum loads / stores): from a sliver to nearly the whole band,
    # zero or wide flanks; every planner size, stacked or not
    wh = None
    if rng.random() < 0.35 and n >= 16:
        fs = 50.0
        fmin = float(rng.uniform(0.0, 0.35)) * fs / 2
        fmax = float(rng.uniform(fmin / (fs / 2) + 0.02, 0.98)) * fs / 2
        ftaper = float(rng.choice([0.0, rng.uniform(0.0, 0.1) * fs / 2]))
        if flags['demean'] and fmin - ftaper <= 0.02 * fs / 2:
            # whitening maps the DC bin of a demeaned window (pure rounding noise, arbitrary phase) to +-g(0):
            # ill-conditioned in any precision, so keep DC outside the band when the window is demeaned
            fmin = ftaper + 0.02 * fs / 2
            fmax = max(fmax, fmin + 0.02 * fs / 2)
        if rng.random() < 0.3:
            # band edges a hair beside a frequency bin of the planned transform (the float edges the kernels compare
            # against once admitted one bin too many: fmax N / fs = 21377.9994 -> 21378.0f)
            probe = CCPlan(ctx, n, imin, imax)
            Np = probe.N
            probe.close()
            snap = lambda f: (round(f * Np / fs) + float(rng.choice([-1.0, 1.0])) * float(rng.uniform(1e-7, 1e-4))) * fs / Np
            fmin2, fmax2 = max(snap(fmin), 0.0), snap(fmax)
            if ftaper > 0:
                ftaper = max(snap(fmin2 + ftaper) - fmin2, 0.0)      # flank ends beside a bin as well
                # a flank narrower than the float32 resolution of a bin coordinate (1e-5 bins at bin 200) is not
                # resolved by the kernels' float edges (the ramp value of a bin INSIDE such a flank is decided as
                # 0 or 1); physically meaningless, so not generated: flanks are sharp or at least 0.01 bin wide
                if ftaper * Np / fs < 0.01:
                    ftaper = 0.0
            if fmax2 > fmin2 and fmin2 - ftaper >= 0:
                fmin, fmax = fmin2, fmax2
        wh = (fs, fmin, fmax, ftaper)
        flags['normalize'] = True
    allp = [(i, j) for i in range(S) for j in range(S)]
    k = int(rng.integers(1, len(allp) + 1))
    pairs = np.array([allp[i] for i in rng.choice(len(allp), k, replace=False)], np.int32)
    tr = (rng.standard_normal((S, T)) + rng.uniform(-1, 1)).astype(np.float32)
    if flags['demean']:
        # float32 conditioning of x - mean(x): a window whose samples nearly coincide (tiny n) loses
        # eps * max|x| / rms(x - mean) of relative accuracy in ANY float32 implementation -- skip those
        cond = 0.0
        for w in range(W):
            seg = tr[:, w * hop:w * hop + n].astype(np.float64)
            dev = np.sqrt(((seg - seg.mean(axis=1, keepdims=True)) ** 2).mean(axis=1))
            cond = max(cond, float((np.abs(seg).max(axis=1) / np.maximum(dev, 1e-300)).max()))
        if cond > 20.0:
            continue
    plan = CCPlan(ctx, n, imin, imax, stack=stack, whiten=wh, **flags)
    if wh:
        # whitening divides by |X[k]|: a bin that is (numerically) zero -- one-bit windows hit EXACT zeros at
        # bins like N/6 where the twiddles are roots of unity of low order -- gets an arbitrary phase in any
        # precision; such windows are skipped, not counted as failures
        gw = R.whiten_weights(plan.N, *wh)
        bad = False
        for w in range(W):
            seg = tr[:, w * hop:w * hop + n].astype(np.float64)
            if flags['demean']:
                seg = seg - seg.mean(axis=1, keepdims=True)
            if flags['onebit']:
                seg = np.sign(seg)
            A = np.abs(np.fft.rfft(seg, plan.N, axis=1))[:, gw > 0]
            if A.size and (A.min(axis=1) < 1e-3 * np.median(A, axis=1)).any():
                bad = True
        if bad:
            plan.close()
            continue
    if os.environ.get('FUZZ_VERBOSE'):
        print(cases, dict(n=n, W=W, hop=hop, S=S, imin=imin, imax=imax, stack=stack, P=len(pairs), N1=plan.N1, N2=plan.N2, batch=plan.batch, **flags), flush=True)
    got = plan.pairs(tr, pairs, W, hop)
    g = R.whiten_weights(plan.N, *wh) if wh else None
    kw = dict(whiten=g, nfft=plan.N) if wh else {}
    ref = R.cc_pairs(tr, pairs, n, hop, normalize=flags['normalize'], index_min=imin, index_max=imax,
                     onebit_=flags['onebit'], demean_=flags['demean'], dtype=np.float64, **kw)[:, :W]
    if stack:
        ref = ref.sum(axis=1)
    scale = np.abs(R.cc_pairs(tr, pairs, n, hop, normalize=flags['normalize'], onebit_=flags['onebit'],
                              demean_=flags['demean'], dtype=np.float64, **kw)[:, :W]).max()
    if stack:
        scale *= W
    err = float(np.abs(got - ref).max() / max(scale, 1e-30))
    worst = max(worst, err); cases += 1
    tol = 1e-5          # one-bit after demean decides its signs in float64 on the GPU, like the oracle
    if not (err < tol):
        if os.environ.get('FUZZ_DUMP'):
            np.savez(os.environ['FUZZ_DUMP'], tr=tr, pairs=pairs, got=np.asarray(got), n=n, hop=hop, W=W, imin=imin, imax=imax,
                     stack=stack, wh=np.array(wh if wh else []), N=plan.N, **{k: v for k, v in flags.items()})
        raise AssertionError(f'{dict(n=n, W=W, hop=hop, S=S, imin=imin, imax=imax, stack=stack, P=len(pairs), whiten=wh, **flags)}: {err}')
    plan.close()
  return cases, worst


if __name__ == '__main__':
    c, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz: {c} cases, worst relative error {w:.2e}')
