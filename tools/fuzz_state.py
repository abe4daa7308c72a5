"""Randomised SEQUENCES of calls on long-lived plans against the oracle (GPU box): what one call leaves behind
(zeroed spectrum buffers, cached descriptors and tables, prepared spectra, an open accumulation, batch size) must
never leak into the next.  Ops: set_whiten with another band, pairs with another pair list / window count /
device or host traces / prepared spectra, set_batch, stack_add ... stack_finish / stack_reset, cc.
usage: fuzz_state.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
from xcorr_amd import _ffi
from xcorr_amd._ffi import check
from xcorr_amd.engine import default_context, CCPlan

TOL = 1e-5


def rel(got, ref, scale):
    return float(np.abs(np.asarray(got) - ref).max() / scale)


def bands(rng):
    fs = 50.0
    lo = float(rng.uniform(0.5, 4.0))
    hi = float(rng.uniform(lo + 1.0, 24.0))
    tp = float(rng.choice([0.0, rng.uniform(0.05, 0.4)]))
    return (fs, lo, hi, min(tp, lo - 0.1))


def run(budget=60.0, seed=0, ctx=None):
    rng = np.random.default_rng(seed)
    ctx = ctx or default_context()
    t_end = time.time() + budget
    ops_done, worst = 0, 0.0
    while time.time() < t_end:
        n = int(rng.choice([rng.integers(200, 1500), rng.integers(1500, 9000)]))
        hop = int(rng.integers(n // 4, n + 1))
        S = int(rng.integers(3, 9))
        Wmax = 5
        T = (Wmax - 1) * hop + n
        whiten = bool(rng.random() < 0.6)
        stack = bool(rng.random() < 0.5)
        wh = bands(rng) if whiten else None
        flags = dict(onebit=False)          # (one-bit windows can hit exactly-zero bins, where whitening is ill-posed: tools/fuzz_cc.py)
        plan = CCPlan(ctx, n, stack=stack, whiten=wh, **flags)
        days = [(rng.standard_normal((S, T)) + 0.0).astype(np.float32) for _ in range(3)]
        ddays = [ctx.array(d) for d in days]
        allp = [(i, j) for i in range(S) for j in range(S)]

        def oracle(tr, pairs, W):
            kw = dict(whiten=R.whiten_weights(plan.N, *wh), nfft=plan.N) if wh else {}
            r = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, onebit_=flags['onebit'], **kw)[:, :W]
            return r, np.abs(r).max()

        log = []
        try:
            for _ in range(int(rng.integers(4, 12))):
                op = rng.choice(['pairs', 'pairs', 'pairs_dev', 'whiten', 'batch', 'accumulate', 'cc'])
                k = int(rng.integers(1, len(allp) + 1))
                pairs = np.array([allp[i] for i in rng.choice(len(allp), min(k, 24), replace=False)], np.int32)
                W = int(rng.integers(1, Wmax + 1))
                d = int(rng.integers(0, 3))
                log.append((str(op), len(pairs), W, d, wh))
                if op == 'whiten':
                    if not whiten:
                        continue
                    wh = bands(rng)
                    check(_ffi.lib().xcx_plan_set_whiten(plan.handle, *wh))
                elif op == 'batch':
                    check(_ffi.lib().xcx_plan_set_batch(plan.handle, int(rng.integers(1, 30))))
                elif op == 'pairs':
                    got = plan.pairs(days[d], pairs, W, hop)
                    ref, sc = oracle(days[d], pairs, W)
                    e = rel(got, ref.sum(axis=1) if stack else ref, sc * (W if stack else 1))
                    worst = max(worst, e); assert e < TOL, (e, log)
                elif op == 'pairs_dev':
                    if not stack and rng.random() < 0.5 and W > 1:
                        plan.prepare(ddays[d], W, hop)
                        if rng.random() < 0.3:          # a preparation that is never consumed as such
                            if whiten and rng.random() < 0.5:
                                wh = bands(rng); check(_ffi.lib().xcx_plan_set_whiten(plan.handle, *wh))
                            else:
                                d = (d + 1) % 3
                    L = plan.L
                    nrows = len(pairs) if stack else len(pairs) * W
                    out = ctx.empty((nrows, L))
                    plan.pairs(ddays[d], pairs, W, hop, out=out)
                    got = out.to_host().reshape((len(pairs), L) if stack else (len(pairs), W, L))
                    out.free()
                    ref, sc = oracle(days[d], pairs, W)
                    e = rel(got, ref.sum(axis=1) if stack else ref, sc * (W if stack else 1))
                    worst = max(worst, e); assert e < TOL, (e, log)
                elif op == 'accumulate':
                    if not stack:
                        continue
                    total, nw, sc = 0.0, 0, 0.0
                    adds = int(rng.integers(1, 7))
                    for a in range(adds):
                        Wa = int(rng.integers(1, Wmax + 1)); da = int(rng.integers(0, 3))
                        plan.stack_add(ddays[da] if rng.random() < 0.5 else days[da], pairs, Wa, hop)
                        r, s = oracle(days[da], pairs, Wa)
                        total = total + r.sum(axis=1); nw += Wa; sc += s * Wa
                    if rng.random() < 0.25:
                        plan.stack_reset()
                    else:
                        got, cnt = plan.stack_finish(pairs)
                        assert cnt == nw
                        e = rel(got, total, sc)
                        worst = max(worst, e); assert e < TOL, (e, log)
                else:
                    if stack:
                        continue
                    x, y = days[d][:2, :n], days[(d + 1) % 3][:2, hop:hop + n]
                    got = plan.cc(x, y)
                    pr = np.array([(0, 1)], np.int32)
                    for b in range(2):
                        ref, sc = oracle(np.stack([x[b], y[b]]), pr, 1)
                        e = rel(got[b], ref[0, 0], sc)
                        worst = max(worst, e); assert e < TOL, (e, log)
                ops_done += 1
        finally:
            plan.close()
            for b in ddays:
                b.free()
    return ops_done, worst


if __name__ == '__main__':
    c, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz_state: {c} operations, worst relative error {w:.2e}')
