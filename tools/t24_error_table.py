#!/usr/bin/env python
"""Error table of the q24 storage of the pair pass's intermediate T (24-bit fixed point, wave_kernels.h)
against float2 storage and against the float64 oracle, on whitened plans.

    python tools/t24_error_table.py > gpurun_out/t24_error_table.txt

Columns: max |gpu - oracle| / max |oracle| with float2 storage, the same with q24 storage, and
max |q24 - f32| / max |oracle|.  Inputs are chosen for the extremes of the fixed-point format:
  noise      independent white noise: the CCF is pure noise, max|cc| is smallest against the quantisation step
  common     the bench's synthetic (common signal + move-out): a CCF with a peak
  identical  the same trace twice: T reaches the bound the scale is derived from (no overflow allowed)
  spiky      white noise with a few huge spikes (one-bit removes them, plain whitening does not)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_numpy as R                      # noqa: E402
import xcorr_amd as X                                  # noqa: E402


def traces(kind, S, T, seed):
    rng = np.random.default_rng(seed)
    if kind == 'noise':
        return rng.standard_normal((S, T)).astype(np.float32)
    if kind == 'common':
        return R.synthetic_traces(S, T, day=seed).astype(np.float32)
    if kind == 'identical':
        x = rng.standard_normal(T).astype(np.float32)
        return np.tile(x, (S, 1))
    if kind == 'spiky':
        x = rng.standard_normal((S, T)).astype(np.float32)
        for s in range(S):
            x[s, rng.integers(0, T, 20)] += 1e4 * rng.standard_normal(20).astype(np.float32)
        return x
    raise ValueError(kind)


def run(ctx, tr, pairs, n, hop, W, storage, **kw):
    os.environ['XCX_T_STORAGE'] = storage
    plan = X.CCPlan(ctx, n, normalize=True, **kw)
    out = plan.pairs(tr, pairs, W, hop)
    N = plan.N
    plan.close()
    return np.asarray(out, dtype=np.float64), N


def main():
    ctx = X.Context(0)
    rows = []
    worst = 0.0
    cases = []
    for n, band in ((180000, (50., .1, 10., .05)), (180000, (50., 1., 20., .5)), (180000, (50., .01, 1., .005)),
                    (120000, (50., .1, 10., .05)), (90000, (50., .5, 5., .1)), (60000, (50., .1, 10., .05)),
                    (45000, (20., .1, 8., .05)), (30000, (50., 1., 10., .5))):
        for kind in ('noise', 'common', 'identical', 'spiky'):
            for onebit in (False, True):
                cases.append((n, band, kind, onebit))
    for ci, (n, band, kind, onebit) in enumerate(cases):
        S, W = 4, 2
        hop = n // 2
        T = hop * (W - 1) + n
        tr = traces(kind, S, T, 100 + ci)
        pairs = np.array([(0, 1), (1, 2), (0, 3), (2, 2)], np.int32)
        kw = dict(whiten=band, onebit=onebit, demean=onebit)
        f32, N = run(ctx, tr, pairs, n, hop, W, 'f32', **kw)
        q24, _ = run(ctx, tr, pairs, n, hop, W, 'q24', **kw)
        g = R.whiten_weights(N, *band)
        ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=N, onebit_=onebit, demean_=onebit)
        e32 = eq = ed = 0.0
        for p in range(len(pairs)):
            for w in range(W):
                m = np.abs(ref[p, w]).max()
                e32 = max(e32, np.abs(f32[p, w] - ref[p, w]).max() / m)
                eq = max(eq, np.abs(q24[p, w] - ref[p, w]).max() / m)
                ed = max(ed, np.abs(q24[p, w] - f32[p, w]).max() / m)
        same = bool(np.array_equal(f32, q24))
        rows.append((n, N, band, kind, onebit, e32, eq, ed, same))
        worst = max(worst, eq)
        print(f'n={n:6d} N={N:6d} band={band[1]:g}-{band[2]:g} Hz @ {band[0]:g} Hz  {kind:9s} onebit={int(onebit)}  '
              f'f32 {e32:.2e}  q24 {eq:.2e}  q24-f32 {ed:.2e}{"  (q24 not selectable: identical)" if same else ""}',
              flush=True)
    print(f'# worst q24 error against the float64 oracle: {worst:.2e} of max|cc| over {len(rows)} cases '
          f'x 4 pairs x 2 windows')
    ctx.close()


if __name__ == '__main__':
    main()
