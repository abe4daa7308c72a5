"""Randomised error sweep of the q24 storage of T (whitened plans; wave_kernels.h) against the float64 oracle:
random window lengths over every column length that has the q24 kernels, random whitening bands (slivers to
nearly everything, with and without flanks), one-bit on / off, inputs from pure noise to identical traces.
Prints the worst |gpu - oracle| / max|oracle| for q24 and, on the same cases, for float2 storage, and the worst
difference between the two storages (what the 24-bit intermediate itself adds).

    python tools/fuzz_q24.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_numpy as R                      # noqa: E402
import xcorr_amd as X                                  # noqa: E402


def run(budget=60.0, seed=0):
    rng = np.random.default_rng(seed)
    ctx = X.Context(0)
    t_end = time.time() + budget
    worst = {'q24': 0.0, 'f32': 0.0, 'q24-f32': 0.0}
    cases = used = 0
    while time.time() < t_end:
        n = int(rng.choice([rng.integers(4100, 20000), rng.integers(20000, 90000), rng.integers(90000, 184320)]))
        fs = 50.0
        fmin = float(rng.uniform(0.02, 0.4)) * fs / 2
        fmax = float(rng.uniform(fmin / (fs / 2) + 0.01, 0.98)) * fs / 2
        ftaper = float(rng.choice([0.0, rng.uniform(0.0, min(0.1 * fs / 2, fmin * 0.9))]))
        onebit = bool(rng.random() < 0.5)
        kind = rng.choice(['noise', 'identical', 'peak', 'tone'])
        S, W = 3, int(rng.integers(1, 3))
        hop = int(rng.integers(n // 4, n + 1))
        T = (W - 1) * hop + n
        tr = rng.standard_normal((S, T)).astype(np.float32)
        if kind == 'identical':
            tr[1] = tr[0]
        elif kind == 'peak':
            tr[1] = np.roll(tr[0], int(rng.integers(-50, 50))) + 0.3 * tr[1]
        elif kind == 'tone':
            tr += (5 * np.sin(2 * np.pi * (fmin + 0.5 * (fmax - fmin)) / fs * np.arange(T))).astype(np.float32)
        pairs = np.array([(0, 1), (1, 2), (2, 2)], np.int32)
        band = (fs, fmin, fmax, ftaper)
        out = {}
        for ts in ('q24', 'f32'):
            os.environ['XCX_T_STORAGE'] = ts
            plan = X.CCPlan(ctx, n, whiten=band, onebit=onebit, demean=onebit)
            sel = plan.t_storage
            out[ts] = np.asarray(plan.pairs(tr, pairs, W, hop), np.float64)
            N = plan.N
            plan.close()
            if ts == 'q24' and sel != 'q24':
                break
        cases += 1
        if len(out) < 2:
            continue                                   # a column length without the q24 kernels
        used += 1
        g = R.whiten_weights(N, *band)
        ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=N, onebit_=onebit, demean_=onebit)
        m = np.abs(ref).max(axis=-1, keepdims=True)
        out['q24-f32'] = out['q24'] - out['f32'] + ref          # the storage's own contribution, same normalisation
        for ts in out:
            e = float((np.abs(out[ts] - ref) / m).max())
            if e > worst[ts]:
                worst[ts] = e
                print(f'  {ts} {e:.2e}: n={n} N={N} band={fmin:.3f}-{fmax:.3f}+-{ftaper:.3f} onebit={int(onebit)} {kind}', flush=True)
    ctx.close()
    return cases, used, worst


if __name__ == '__main__':
    c, u, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz_q24: {u} cases with q24 storage (of {c} drawn), worst relative error q24 {w["q24"]:.2e}, '
          f'float2 storage on the same cases {w["f32"]:.2e}, worst |q24 - float2| {w["q24-f32"]:.2e} (all of max|cc|)')
