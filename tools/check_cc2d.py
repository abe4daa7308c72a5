"""Randomised check of xcx_cc2d's peak value and POSITION against the oracle (GPU box): tile shapes on either side
of the padded sizes (2L - 1 lags of a padded row of 2R >= 2L points: the arg-max must skip the padding columns),
peak-only and full-map kernels, pair lists over several tile groups and -- with XCX_CC2D_PASS_MB small -- several
passes on the two pipeline streams.   python tools/check_cc2d.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
from xcorr_amd import ops
from xcorr_amd.engine import default_context


def run(budget=40.0, seed=0):
    rng = np.random.default_rng(seed)
    ctx = default_context()
    t_end = time.time() + budget
    cases, worst, ties = 0, 0.0, 0
    while time.time() < t_end:
        F = int(rng.choice([1, 2, 3, 8, 16, 17, 31, 32, 33, 64, 100, 128, 129, 256]))
        L = int(rng.choice([1, 2, 5, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 255, 256, 257, 500, 512, 513, 1000, 1024,
                            1025, 2048, 3000, 4096]))
        if F * L > 300000:
            F = max(1, 300000 // L)
        T = int(rng.integers(2, 8))
        kind = rng.choice(['noise', 'squares', 'shifted', 'negative'])
        if kind == 'noise':
            t = rng.standard_normal((T, F, L))
        elif kind == 'squares':
            t = np.square(rng.standard_normal((T, F, L)))
        elif kind == 'negative':
            t = -np.abs(rng.standard_normal((T, F, L))) - 0.1       # every lag of the CCF of (a, -b) is negative
            t[::2] *= -1
        else:
            base = rng.standard_normal((F + 8, L + 24))
            t = np.stack([base[int(rng.integers(0, 9)):, int(rng.integers(0, 25)):][:F, :L] for _ in range(T)])
        t = np.ascontiguousarray(t, dtype=np.float32)
        P = int(rng.integers(1, 40))
        prs = [(int(rng.integers(0, T)), int(rng.integers(0, T))) for _ in range(P)]
        nrm = bool(rng.random() < 0.6)
        full = bool(rng.random() < 0.4)
        res = ops.correlate2d(ctx, t, prs, normalize=nrm, full=full)
        pk, ix = res[0], res[1]
        for q, (a, b) in enumerate(prs[:6]):
            ref = R.correlate2d(t[a], t[b], normalize=nrm)
            sc = np.abs(ref).max()
            e = abs(float(pk[q]) - float(ref.max())) / sc
            worst = max(worst, e)
            assert e <= 1e-5, (F, L, T, kind, nrm, full, q, e)
            got = tuple(int(v) for v in ix[q])
            assert 0 <= got[0] < 2 * F - 1 and 0 <= got[1] < 2 * L - 1, (F, L, got)
            want = np.unravel_index(int(np.argmax(ref)), ref.shape)
            if got != tuple(int(v) for v in want):
                # a float32 near-tie: the position must hold a value within rounding of the maximum
                assert ref[got] >= ref.max() - 2e-5 * sc, (F, L, T, kind, nrm, full, q, got, want, ref[got], ref.max())
                ties += 1
            if full:
                m = res[2][q]
                m = m.to_host() if hasattr(m, 'to_host') else m
                assert np.abs(m - ref).max() <= 1e-5 * sc, (F, L, 'map')
                assert got == np.unravel_index(int(np.argmax(m)), m.shape), (F, L, 'first maximum of the map', got)
        cases += 1
    return cases, worst, ties


if __name__ == '__main__':
    c, w, t = run(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'check_cc2d: {c} calls clean, worst peak error {w:.2e} of max|cc|, {t} near-tie positions '
          f'(OVERLAP={os.environ.get("XCX_CC2D_OVERLAP", "1")}, PASS_MB={os.environ.get("XCX_CC2D_PASS_MB", "-")})')
