"""GPU: the driver-timed workload itself (BASELINE configs[2]) checked
end to end against the oracle at FULL size: 64 stations x a 4 320 000-sample day,
demean -> cosine taper -> band-pass sosfilt per station-day, one-bit + whitening per
window, full-lag CCFs at N = 368 640 -- through exactly the calls bench.py times
(xcorr_amd.workload.StationDay.step).

Three comparisons on 18 sampled (pair, window) rows:
  A. the composed chain: oracle preprocess (float64) -> oracle per-window chain vs the GPU rows.
     One-bit makes this discontinuous: a sample whose band-passed value lies within float32
     rounding of zero may take the other sign on the GPU (its preprocess stores float32 rows,
     the oracle keeps float64).  Such flips are COUNTED from the downloaded GPU rows; with none
     the bar is the strict 1e-5, each flip adds 5e-4 (a unit impulse in one of 180 000 samples
     moves a whitened CCF by ~2/sqrt(n) of its noise floor).
  B. preprocess alone: GPU station-day rows vs the float64 oracle (covers xcx_sosfilt on
     4.32 M-sample rows).
  C. per-window chain alone, strict: oracle fed with the GPU's own preprocessed float32 rows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R                       # noqa: E402  (checker only)
from xcorr_amd import workload                          # noqa: E402
from xcorr_amd.synth import synthetic_traces            # noqa: E402

TOL = 1e-5
FS = workload.FS

# (i, j, window): first/last windows (inside the day taper), corner pairs, neighbours, far pairs
SAMPLES = [(0, 1, 0), (0, 63, 0), (62, 63, 46), (0, 1, 46), (31, 32, 23), (5, 40, 11), (17, 18, 1),
           (0, 32, 45), (10, 53, 30), (20, 21, 22), (7, 8, 8), (1, 62, 40), (33, 60, 5), (2, 3, 35),
           (15, 47, 17), (25, 26, 44), (0, 2, 12), (40, 63, 28)]


def _rel(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / np.abs(ref).max())


def _oracle_preprocess(raw_rows, T):
    x = raw_rows.astype(np.float64)
    x = R.demean(x)
    x = x * R.cosine_taper_obspy(T, FS, *workload.TAPER)
    return R.sosfilt(workload.band_sos(), x)


@pytest.fixture(scope='module')
def day(ctx):
    S, T, n, hop = 64, 4320000, 180000, 90000
    raw = synthetic_traces(S, T, day=0)
    pairs = R.all_pairs(S)
    sd = workload.StationDay(ctx, S, T, n, hop, pairs).load(raw)
    sd.step()
    sd.step()                      # a second step from the kept raw rows must give the same answer
    ctx.sync()
    yield sd, raw, pairs
    sd.close()


def test_config3_geometry(day):
    sd, raw, pairs = day
    assert (sd.P, sd.W, sd.L, sd.plan.N) == (2016, 47, 359999, 368640)
    assert sd.nrows == 94752


def test_config3_composed_chain_full_size(day):
    sd, raw, pairs = day
    S, T, n, hop, N = sd.S, sd.T, sd.n, sd.hop, sd.plan.N
    stations = sorted({s for i, j, _ in SAMPLES for s in (i, j)})
    ref_pre = {s: _oracle_preprocess(raw[s][None], T)[0] for s in stations}
    gpu_pre = {s: sd.work.rows_to_host(s, 1, T)[0] for s in stations}
    # ---- B: preprocess of 4.32 M-sample rows ------------------------------------------
    for s in stations:
        assert _rel(gpu_pre[s], ref_pre[s]) < 2e-6, f'station {s}'
    g = R.whiten_weights(N, *workload.WHITEN)
    pidx = {(int(i), int(j)): k for k, (i, j) in enumerate(pairs)}
    worst_a = worst_c = 0.0
    total_flips = 0
    for i, j, w in SAMPLES:
        got = sd.rows(pidx[(i, j)], w)
        assert np.all(np.isfinite(got))
        sl = slice(w * hop, w * hop + n)

        def chain(xi, xj):
            return R.cc_pairs(np.stack([xi[sl], xj[sl]]), [(0, 1)], n, n, onebit_=True, whiten=g, nfft=N,
                              dtype=np.float64)[0, 0]
        # ---- C: strict, same float32 rows in ------------------------------------------
        ec = _rel(got, chain(gpu_pre[i], gpu_pre[j]))
        assert ec < TOL, f'per-window chain, pair ({i},{j}) window {w}: {ec:.2e}'
        # ---- A: composed, oracle preprocess in float64 --------------------------------
        flips = int(sum(np.count_nonzero(np.sign(gpu_pre[s][sl].astype(np.float64)) != np.sign(ref_pre[s][sl]))
                        for s in (i, j)))
        ea = _rel(got, chain(ref_pre[i], ref_pre[j]))
        assert ea < TOL + 5e-4 * flips, f'composed chain, pair ({i},{j}) window {w}: {ea:.2e} with {flips} sign flips'
        worst_a, worst_c, total_flips = max(worst_a, ea), max(worst_c, ec), total_flips + flips
        # the synthetic moveout puts the peak of pair (i, j) at lag 25 (i - j) samples
        assert int(np.argmax(got)) == n - 1 + 25 * (i - j)
    print(f'config 3 full size: composed {worst_a:.2e} ({total_flips} one-bit sign flips), per-window {worst_c:.2e}')


def test_config3_padding_lags_untouched(ctx, day):
    """ld_out = 360 000: the one padding float of every sampled row is never written."""
    sd, raw, pairs = day
    sd.out.zero()
    sd.step()
    ctx.sync()
    for r in (0, 1, sd.nrows // 2, sd.nrows - 1):
        row = sd.out.rows_to_host(r, 1, sd.ld)[0]
        assert np.all(row[sd.L:] == 0) and np.any(row[:sd.L] != 0)
