"""GPU: BASELINE configs[3] maths on ONE GPU at full pair count -- 64 stations, P = 2016 pairs,
30 station-days of hour-long half-overlapping windows accumulated in the frequency domain
(xcx_cc1d_stack_add x 30, one xcx_cc1d_stack_finish), against
  (a) the sum of the 30 per-day stacks the plain stacked call produces (sampled rows, float64 sum),
  (b) the oracle (sum over days and windows of cc) on sampled pairs.
The multi-GPU run adds only the RCCL reduce of the per-rank partial stacks
(tests/test_gpu_multirank.py); the per-rank arithmetic is exactly this loop.

Day length: 6 windows (n = 180 000, hop 90 000 -> 630 000 samples) instead of 47 keeps the
test to about a minute; window count per day does not enter the stack algebra (the 47-window
day is covered at full size by tests/test_gpu_config3.py and bench.py --stack)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R                       # noqa: E402  (checker only)
import xcorr_amd as X                                   # noqa: E402
from xcorr_amd.synth import fast_day                    # noqa: E402

TOL = 1e-5
SAMPLED = [(0, 1), (0, 63), (62, 63), (31, 32), (5, 40), (17, 18), (0, 32), (10, 53), (20, 21), (7, 8),
           (1, 62), (33, 60), (2, 3), (15, 47), (25, 26), (40, 63)]


def _day(S, T, d):
    return fast_day(S, T, d)              # the generator bench.py --stack --days uses


def test_thirty_day_stack_full_pair_count(ctx):
    S, n, hop, W, ndays = 64, 180000, 90000, 6, 30
    T = hop * (W - 1) + n
    pairs = R.all_pairs(S)
    P, L = len(pairs), 2 * n - 1
    ld = (L + 31) // 32 * 32
    pidx = {(int(i), int(j)): k for k, (i, j) in enumerate(pairs)}
    rows = [pidx[p] for p in SAMPLED]
    acc = X.CCPlan(ctx, n, stack=True, demean=True)
    per = X.CCPlan(ctx, n, stack=True, demean=True)
    assert acc.N == 368640
    dtr = ctx.empty((S, T))
    out_day = ctx.empty((P, ld))
    out_acc = ctx.empty((P, ld))
    sum_days = np.zeros((len(rows), L), np.float64)
    ref = np.zeros((3, L), np.float64)
    for d in range(ndays):
        x = _day(S, T, d)
        dtr.copy_from(x)
        acc.stack_add(dtr, pairs, W, hop)
        per.pairs(dtr, pairs, W, hop, out=out_day, ld_out=ld)
        for k, r in enumerate(rows):
            sum_days[k] += out_day.rows_to_host(r, 1, L)[0]
        for k, (i, j) in enumerate(SAMPLED[:3]):                     # oracle on three pairs
            ref[k] += R.cc_pairs(x[[i, j]], [(0, 1)], n, hop, demean_=True, nfft=acc.N,
                                 dtype=np.float64)[0].sum(axis=0)
    _, nw = acc.stack_finish(pairs, out=out_acc, ld_out=ld)
    assert nw == ndays * W
    worst = 0.0
    for k, r in enumerate(rows):
        got = out_acc.rows_to_host(r, 1, L)[0].astype(np.float64)
        e = np.abs(got - sum_days[k]).max() / np.abs(sum_days[k]).max()
        worst = max(worst, e)
        assert e < TOL, f'pair {SAMPLED[k]}: accumulated vs sum of per-day stacks {e:.2e}'
        i, j = SAMPLED[k]
        assert int(np.argmax(got)) == n - 1 + 25 * (i - j)
        if k < 3:
            eo = np.abs(got - ref[k]).max() / np.abs(ref[k]).max()
            assert eo < TOL, f'pair {SAMPLED[k]}: vs oracle {eo:.2e}'
    # stack sizes of configs[3]: the 2016 x 359 999 float32 stack is 2.90 GB
    assert abs(P * L * 4 / 1e9 - 2.90) < 0.01
    print(f'30-day stack, P=2016: accumulated vs sum of per-day stacks {worst:.2e}')
    acc.close(); per.close()
