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


def test_stack_of_full_days_through_the_chain_the_bench_times(ctx):
    """configs[3] at the chain `bench.py --stack --days D` times -- 64 stations x 47-window
    days through StationDay(preprocess=True, stack=True): demean -> cosine taper -> band-pass per station-day,
    one-bit + whitening per window, frequency-domain accumulation (xcx_cc1d_stack_add per day, ONE
    xcx_cc1d_stack_finish) -- the very loop of bench.py's step(), for 3 days at P = 2016.  Sampled pairs against
    the oracle's R.cc_pairs(onebit_=True, whiten=g) summed over the 141 windows:
      C. strict 1e-5 with the oracle fed the GPU's own float32 preprocessed rows,
      A. the composed chain (oracle preprocess in float64) with the config-3 flip rule: one-bit is discontinuous,
         every counted sign difference between the float32 and float64 rows of a pair's windows adds 5e-4 of ONE
         window's noise floor, i.e. 5e-4 / sqrt(windows) of the stack's."""
    from xcorr_amd import workload
    S, T, n, hop, ndays = 64, 4320000, 180000, 90000, 3
    pairs = R.all_pairs(S)
    sd = workload.StationDay(ctx, S, T, n, hop, pairs, preprocess=True, stack=True)
    W, L, N = sd.W, sd.L, sd.plan.N
    assert (W, sd.P, N) == (47, 2016, 368640)
    sampled = [(0, 1), (5, 40), (62, 63)]
    pidx = {(int(i), int(j)): k for k, (i, j) in enumerate(pairs)}
    stations = sorted({s for p in sampled for s in p})
    g = R.whiten_weights(N, *workload.WHITEN)
    taper = R.cosine_taper_obspy(T, workload.FS, *workload.TAPER)
    ref_c = {p: np.zeros(L) for p in sampled}
    ref_a = {p: np.zeros(L) for p in sampled}
    flips = {p: 0 for p in sampled}
    try:
        for d in range(ndays):
            raw = _day(S, T, d)
            sd.load(raw)
            sd.front()                                              # exactly bench.py's multi-day step()
            sd.plan.stack_add(sd.view, pairs, W, hop, stride_s=T)
            gpu_pre = {s: sd.work.rows_to_host(s, 1, T)[0] for s in stations}
            ora_pre = {s: R.sosfilt(workload.band_sos(), R.demean(raw[s][None].astype(np.float64)) * taper)[0]
                       for s in stations}
            for (i, j) in sampled:
                ref_c[(i, j)] += R.cc_pairs(np.stack([gpu_pre[i], gpu_pre[j]]), [(0, 1)], n, hop, onebit_=True,
                                            whiten=g, nfft=N, dtype=np.float64)[0].sum(axis=0)
                ref_a[(i, j)] += R.cc_pairs(np.stack([ora_pre[i], ora_pre[j]]), [(0, 1)], n, hop, onebit_=True,
                                            whiten=g, nfft=N, dtype=np.float64)[0].sum(axis=0)
                for s in (i, j):          # every sample is in two windows except the day's two half-window flanks
                    flips[(i, j)] += 2 * int(np.count_nonzero(np.sign(gpu_pre[s].astype(np.float64)) != np.sign(ora_pre[s])))
        _, nw = sd.plan.stack_finish(pairs, out=sd.out, ld_out=sd.ld)
    except BaseException:
        sd.plan.stack_reset()
        raise
    assert nw == ndays * W
    worst_c = worst_a = 0.0
    for (i, j) in sampled:
        got = sd.rows(pidx[(i, j)], 0).astype(np.float64)
        ec = np.abs(got - ref_c[(i, j)]).max() / np.abs(ref_c[(i, j)]).max()
        ea = np.abs(got - ref_a[(i, j)]).max() / np.abs(ref_a[(i, j)]).max()
        worst_c, worst_a = max(worst_c, ec), max(worst_a, ea)
        assert ec < TOL, f'pair ({i},{j}): stack vs oracle on the GPU rows {ec:.2e}'
        assert ea < TOL + 5e-4 * flips[(i, j)] / np.sqrt(nw), \
            f'pair ({i},{j}): composed chain {ea:.2e} with {flips[(i, j)]} one-bit sign flips'
        assert int(np.argmax(got)) == n - 1 + 25 * (i - j)
    print(f'3 full days through the bench chain, P=2016: per-window chain {worst_c:.2e}, composed {worst_a:.2e} '
          f'({sum(flips.values())} sign flips)')
    sd.close()
