"""Randomised parity sweep of the segmented (long-window) plans against the ordinary single-transform plan and the
oracle, with the segmentation forced at small sizes (XCX_SEG_FORCE, block transforms of 4096 / 8192 points): window
lengths that leave ragged last blocks, lag slices on either side of zero lag or across it, every flag, batches, and
the pairs(traces, windows) form.  The distinct-segment layout of round 4 (one station per row: NS = 2 nb - 2 + nc
segments + nb blocks) is what this exercises.     python tools/fuzz_seg.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
import xcorr_amd as X
from xcorr_amd.engine import default_context


def run(budget=60.0, seed=0):
    rng = np.random.default_rng(seed)
    ctx = default_context()
    t_end = time.time() + budget
    cases, worst = 0, 0.0
    while time.time() < t_end:
        Np = int(rng.choice([4096, 8192]))
        n = int(rng.integers(Np + 7, 12 * Np))
        b = int(rng.integers(1, 5))
        L = 2 * n - 1
        kind = rng.integers(0, 5)
        if kind == 0: imin, imax = 0, L
        elif kind == 1: imin = n - 1; imax = int(rng.integers(n, L + 1))                  # non-negative lags only
        elif kind == 2: imax = n - 1; imin = int(rng.integers(0, n - 2))                  # negative lags only
        else:
            imin = int(rng.integers(0, L - 2)); imax = int(rng.integers(imin + 1, L + 1))
        kw = dict(index_min=imin, index_max=imax)
        f = rng.integers(0, 5)
        if f == 1: kw.update(normalize=False, demean=True)
        elif f == 2: kw.update(onebit=True, demean=True)
        elif f == 3: kw.update(unbiased=True)
        elif f == 4: kw.update(demean=True)
        x = (rng.standard_normal((b, n)) + rng.uniform(-1, 1)).astype(np.float32)
        y = (np.roll(x, int(rng.integers(-n // 3, n // 3)), axis=1) + rng.standard_normal((b, n))).astype(np.float32)
        ref_plan = X.CCPlan(ctx, n, **kw)
        ref = np.asarray(ref_plan.cc(x, y), np.float64)
        os.environ['XCX_SEG_FORCE'] = '1'; os.environ['XCX_SEG_N'] = str(Np)
        try:
            seg = X.CCPlan(ctx, n, **kw)
        finally:
            del os.environ['XCX_SEG_FORCE']; del os.environ['XCX_SEG_N']
        assert seg.N == Np, (seg.N, Np)
        got = np.asarray(seg.cc(x, y), np.float64)
        # errors are judged on the BIASED estimate: n / (n - |lag|) multiplies the float32 rounding of the outermost
        # lags by up to n, in every engine alike
        w = R.weight_full(n)[imin:imax] if kw.get('unbiased') else 1.0
        full = X.CCPlan(ctx, n, **{k: v for k, v in kw.items() if not k.startswith('index') and k != 'unbiased'})
        scale = float(np.abs(np.asarray(full.cc(x, y))).max())
        e = float(np.abs((got - ref) / w).max() / scale)
        # the oracle on row 0 (plain normalised cc and the demeaned raw form; the other flags are covered engine-to-engine)
        if f in (0, 3, 4):
            xo, yo = x[0].astype(np.float64), y[0].astype(np.float64)
            if f == 4: xo, yo = xo - xo.mean(), yo - yo.mean()
            o = R.cc_fair(xo, yo, normalize=True, dtype=np.float64)
            e = max(e, float(np.abs(got[0] / w - o[imin:imax]).max() / np.abs(o).max()))
        tol = 1e-5
        if rng.random() < 0.3 and not kw.get('unbiased'):
            # the pairs(traces, windows) form: rows that share a station window share its segment transforms
            S_ = int(rng.integers(2, 5)); W_ = int(rng.integers(1, 3)); hop_ = int(rng.integers(1, 900))
            tr = (rng.standard_normal((S_, n + (W_ - 1) * hop_)) + 0.3).astype(np.float32)
            pr = np.array([(int(rng.integers(0, S_)), int(rng.integers(0, S_))) for _ in range(int(rng.integers(1, 7)))], np.int32)
            a_ = np.asarray(seg.pairs(tr, pr, W_, hop_), np.float64)
            b_ = np.asarray(ref_plan.pairs(tr, pr, W_, hop_), np.float64)
            sc = float(np.abs(np.asarray(full.pairs(tr, pr, W_, hop_))).max())
            e = max(e, float(np.abs(a_ - b_).max() / sc))
        cases += 1
        worst = max(worst, e)
        if not e < tol:
            raise AssertionError(f'seg: error {e} n={n} Np={Np} b={b} slice=({imin},{imax}) kw={kw} (seed {seed}, case {cases})')
        seg.close(); ref_plan.close(); full.close()
    return cases, worst


if __name__ == '__main__':
    c, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz_seg: {c} cases, worst relative error {w:.2e}')
