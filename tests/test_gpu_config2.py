"""GPU: BASELINE configs[1] at FULL size -- 16 stations x a 4 320 000-sample day,
120 pairs x 47 windows = 5 640 full-lag CCFs (8.1 GB resident), no preprocessing -- through the object
bench.py times (`StationDay(preprocess=False).step()`).

Checked on ALL 5 640 CCFs (size-independent properties, two strided device-to-host column reads):
  * the zero-lag sample equals the normalised dot product of the two windows (float64 on the host),
  * the maximum sits at lag 25 (i - j) samples (the synthetic moveout) and is below 1 + 1e-5,
and 8 full rows against the oracle's cc (first / last window, corner and neighbour pairs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R                       # noqa: E402  (checker only)
from xcorr_amd import ops, workload                     # noqa: E402
from xcorr_amd.synth import synthetic_traces            # noqa: E402

TOL = 1e-5
ROWS = [(0, 1, 0), (0, 15, 0), (14, 15, 46), (0, 15, 46), (7, 8, 23), (3, 12, 11), (5, 6, 1), (2, 13, 40)]


def test_config2_full_day(ctx):
    S, T, n, hop = 16, 4320000, 180000, 90000
    raw = synthetic_traces(S, T, day=2)
    pairs = R.all_pairs(S)
    sd = workload.StationDay(ctx, S, T, n, hop, pairs, preprocess=False).load(raw)
    try:
        sd.step()
        ctx.sync()
        P, W, L = sd.P, sd.W, sd.L
        assert (P, W, L, sd.plan.N, sd.nrows) == (120, 47, 2 * n - 1, 368640, 5640)
        # ---- zero-lag column of every CCF (one strided copy of 5 640 floats) -------------
        zl = sd.out.columns_from(n - 1).rows_to_host(0, sd.nrows, 1).reshape(P, W)
        t64 = raw.astype(np.float64)
        for w in range(W):
            seg = t64[:, w * hop:w * hop + n]
            nrm = np.linalg.norm(seg, axis=1)
            gram = (seg @ seg.T) / np.outer(nrm, nrm)
            ref = np.array([gram[i, j] for i, j in pairs])
            assert np.abs(zl[:, w] - ref).max() < TOL, f'window {w}'
        # ---- peak of every CCF: value and position, reduced on the device -----------------
        mask = np.ones(sd.ld, bool)
        mask[L:] = False                                  # the padding float of each row is not part of the CCF
        peak, _, arg = ops.snr(ctx, sd.out, mask, mask)
        peak, arg = peak.reshape(P, W), arg.reshape(P, W)
        want = np.array([n - 1 + 25 * (i - j) for i, j in pairs])
        assert (arg == want[:, None]).all()
        assert peak.max() < 1.0 + TOL and peak.min() > 0.5
        # ---- full rows against the oracle ---------------------------------------------------
        pidx = {(int(i), int(j)): k for k, (i, j) in enumerate(pairs)}
        worst = 0.0
        for i, j, w in ROWS:
            got = sd.rows(pidx[(i, j)], w)
            ref = R.cc_fair(t64[i, w * hop:w * hop + n], t64[j, w * hop:w * hop + n], dtype=np.float64)
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            assert err < TOL, f'pair ({i},{j}) window {w}: {err:.2e}'
            worst = max(worst, err)
        print(f'config 2 full day: 5640 CCFs, zero-lag and peak checks passed, 8 oracle rows worst {worst:.2e}')
    finally:
        sd.close()
