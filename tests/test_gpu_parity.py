"""GPU: the HIP cross-correlation path through the C ABI against (a) vectors
frozen from the reference's own util/cc.py, (b) the oracle on seeded inputs,
(c) size-independent properties at BASELINE.json's full size.

Tolerance (north_star): 1e-5 relative float32, scaled by max|cc|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402  (checker only)
import xcorr_amd as X                      # noqa: E402
from xcorr_amd._ffi import XcxError        # noqa: E402

TOL = 1e-5


def relerr(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300))


def test_native_library_loaded():
    from xcorr_amd import _ffi
    assert _ffi.lib() is not None
    import os
    maps = open(f'/proc/{os.getpid()}/maps').read()
    assert 'libxcx.so' in maps


def test_reference_golden_vectors(ctx, golden):
    """Outputs of the reference's cc() (float64 FFT path), all fixture sizes."""
    for n in golden['sizes']:
        n = int(n)
        x, y = golden[f'x_{n}'], golden[f'y_{n}']
        for nrm in (0, 1):
            plan = X.CCPlan(ctx, n, normalize=bool(nrm))
            got = plan.cc(x[None], y[None])[0]
            plan.close()
            assert got.shape == (2 * n - 1,) and got.dtype == np.float32
            assert relerr(got, golden[f'cc_{n}_norm{nrm}']) < TOL, (n, nrm)


def test_reference_kat(ctx, golden):
    x = golden['kat_x'].astype(np.float32)
    y = golden['kat_y'].astype(np.float32)
    got = X.CCPlan(ctx, 4, normalize=False).cc(x[None], y[None])[0]
    np.testing.assert_allclose(got, [0, .5, 2, 3.5, 5, 4, 0], atol=5e-6)
    got = X.CCPlan(ctx, 4, normalize=True).cc(x[None], y[None])[0]
    np.testing.assert_allclose(got, golden['kat_norm'], atol=1e-6)


def test_reference_full_size_digest(ctx, golden):
    """n = 180000 (1 h @ 50 Hz): digest of the reference run."""
    n = int(golden['big_n'])
    rng = np.random.default_rng(12345)
    a = rng.standard_normal(n).astype(np.float32)
    b = (np.roll(a, 250) + 0.5 * rng.standard_normal(n).astype(np.float32)).astype(np.float32)
    plan = X.CCPlan(ctx, n)
    assert plan.N == 368640
    r = plan.cc(a[None], b[None])[0]
    mx = float(golden['big_max'])
    assert r.shape == (359999,)
    assert int(np.argmax(r)) == int(golden['big_argmax'])
    assert abs(float(r.max()) - mx) < TOL * mx
    assert np.abs(r[golden['big_probe_idx']] - golden['big_probe_val']).max() < TOL * mx
    assert abs(float(np.sqrt((r.astype(np.float64) ** 2).sum())) - float(golden['big_l2'])) < 1e-4
    # full vector against the oracle
    assert relerr(r, R.cc_fair(a, b, dtype=np.float64)) < TOL


@pytest.mark.parametrize('n', [1, 2, 3, 7, 64, 255, 256, 257, 1000, 4097, 20000, 65536, 131072, 200000, 262144,
                               300000, 393216])
def test_sizes_and_planner(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((2, n)).astype(np.float32)
    y = (0.5 * np.roll(x, max(1, n // 5), axis=-1) + rng.standard_normal((2, n))).astype(np.float32)
    for nrm in (True, False):
        plan = X.CCPlan(ctx, n, normalize=nrm)
        assert plan.N >= 2 * n - 1 and plan.N == plan.N1 * plan.N2
        got = plan.cc(x, y)
        plan.close()
        assert relerr(got, R.cc_fair(x, y, normalize=nrm, dtype=np.float64)) < TOL


def test_unsupported_and_invalid_arguments(ctx):
    from xcorr_amd._ffi import XcxError
    # beyond the largest single transform: a segmented plan, except for flags it cannot serve
    big = X.CCPlan(ctx, 400000)
    assert big.N == 368640 and big.L == 799999
    big.close()
    with pytest.raises(XcxError, match='no supported transform length'):
        X.CCPlan(ctx, 400000, stack=True)
    with pytest.raises(XcxError, match='no supported transform length'):
        X.CCPlan(ctx, 400000, nfft=1000000)
    with pytest.raises(XcxError, match='lag slice'):
        X.CCPlan(ctx, 100, index_min=150, index_max=100)
    with pytest.raises(XcxError, match='n must be'):
        X.CCPlan(ctx, 0)
    plan = X.CCPlan(ctx, 100)
    with pytest.raises(XcxError, match='outside'):
        plan.pairs(np.zeros((2, 100), np.float32), [(0, 2)], 1, 100)
    with pytest.raises(ValueError):
        plan.cc(np.zeros((1, 100), np.float32), np.zeros((1, 99), np.float32))
    # XCX_WHITEN without a band is an error, not a silent no-op
    import ctypes as C
    from xcorr_amd import _ffi
    h = C.c_void_p()
    _ffi.check(_ffi.lib().xcx_plan_cc1d(ctx.handle, 100, 0, 0, 0, _ffi.NORMALIZE | _ffi.WHITEN, C.byref(h)))
    x = np.zeros((1, 100), np.float32)
    o = np.zeros((1, 199), np.float32)
    with pytest.raises(XcxError, match='xcx_plan_set_whiten'):
        _ffi.check(_ffi.lib().xcx_cc1d(h, _ffi.ptr(x), _ffi.ptr(x), 1, _ffi.ptr(o), 0))
    _ffi.lib().xcx_plan_destroy(h)


def test_zero_and_constant_inputs(ctx):
    n = 500
    z = np.zeros((1, n), np.float32)
    o = np.ones((1, n), np.float32)
    plan = X.CCPlan(ctx, n, normalize=True)
    r = plan.cc(z, o)[0]
    assert np.all(np.isnan(r))                     # 0/0 like the reference
    r = X.CCPlan(ctx, n, normalize=False).cc(z, o)[0]
    assert np.all(r == 0)
    r = plan.cc(o, o)[0]                           # triangle, peak 1 at zero lag
    ref = (n - np.abs(np.arange(-(n - 1), n))) / n
    np.testing.assert_allclose(r, ref, atol=TOL)


CASES = [
    dict(),
    dict(demean=True),
    dict(onebit=True, demean=True),
    dict(index_min=1000, index_max=4500),
    dict(index_min=2999, index_max=3000),
    dict(normalize=False),
    dict(whiten=(50., 1., 10., .5)),
    dict(whiten=(50., 1., 10., .5), onebit=True, demean=True),
]


@pytest.mark.parametrize('kw', CASES, ids=[str(sorted(c)) for c in CASES])
def test_pairs_engine(ctx, kw):
    S, n, hop, W = 5, 3000, 1500, 4
    T = hop * (W - 1) + n
    tr = R.synthetic_traces(S, T)
    pairs = R.all_pairs(S)
    plan = X.CCPlan(ctx, n, **kw)
    got = plan.pairs(tr, pairs, W, hop)
    wh = kw.get('whiten')
    g = R.whiten_weights(plan.N, *wh) if wh else None
    ref = R.cc_pairs(tr, pairs, n, hop, normalize=kw.get('normalize', True), dtype=np.float64,
                     demean_=kw.get('demean', False), onebit_=kw.get('onebit', False),
                     index_min=kw.get('index_min', 0), index_max=kw.get('index_max'),
                     whiten=g, nfft=plan.N)
    assert got.shape == ref.shape
    assert relerr(got, ref) < TOL
    plan.close()


def test_whitening_band_edge_next_to_a_bin(ctx):
    # band edges a hair inside/outside a frequency bin (1-2e-5 bins: below float32 resolution):
    # the float edges the kernels compare against must keep the double-precision decision for every bin
    S, n, hop, W = 3, 3000, 1500, 2
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    N = X.CCPlan(ctx, n).N
    fs = 50.
    for lo_bin, hi_bin in ((300 + 1e-5, 1500 - 2e-5), (300 - 1e-5, 1500 + 2e-5)):
        wh = (fs, lo_bin * fs / N, hi_bin * fs / N, 0.)
        plan = X.CCPlan(ctx, n, whiten=wh)
        assert plan.N == N
        g = R.whiten_weights(N, *wh)
        assert np.count_nonzero(g) == (1199 if lo_bin > 300 else 1201)
        assert np.float32(lo_bin) == 300 and np.float32(hi_bin) == 1500
        ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=N)
        assert relerr(plan.pairs(tr, pairs, W, hop), ref) < TOL
        plan.close()


def test_rewhitening_a_plan_with_another_band(ctx):
    # xcx_plan_set_whiten again on the same plan: narrow band, wide band, narrow again.  The forward pass does not
    # store above the band and the pair pass reads "stored zeros" there -- what a wider band wrote in between must
    # not survive (it did: error 0.93 before the zeroed-buffer records were dropped on a band change)
    from xcorr_amd import _ffi
    from xcorr_amd._ffi import check
    S, n, hop, W = 4, 3000, 1500, 3
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    narrow, wide = (50., 1., 5., 0.), (50., 1., 12., .5)
    for stack in (False, True):
        plan = X.CCPlan(ctx, n, whiten=narrow, stack=stack)
        for wh in (narrow, wide, narrow, wide, narrow):
            check(_ffi.lib().xcx_plan_set_whiten(plan.handle, *wh))
            got = plan.pairs(tr, pairs, W, hop)
            g = R.whiten_weights(plan.N, *wh)
            ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=plan.N)
            if stack:
                ref = ref.sum(axis=1)
            assert relerr(got, ref) < TOL, (stack, wh)
        plan.close()


def test_pairs_unbiased_and_order(ctx):
    S, n, hop, W = 4, 2000, 1000, 3
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = np.array([(3, 0), (1, 1), (0, 2), (2, 1)], np.int32)   # arbitrary order, auto-pair, i>j
    ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64)
    got = X.CCPlan(ctx, n).pairs(tr, pairs, W, hop)
    assert relerr(got, ref) < TOL
    # unbiased: errors scale with the weight n/(n-|lag|) (up to n at the ends),
    # so compare after dividing the weight out again
    gotu = X.CCPlan(ctx, n, unbiased=True).pairs(tr, pairs, W, hop)
    assert relerr(gotu / R.weight_full(n), ref) < TOL
    assert relerr(gotu[..., n - 1], ref[..., n - 1]) < TOL        # weight 1 at zero lag


def test_stack_equals_sum_of_windows(ctx):
    S, n, hop, W = 4, 3000, 1500, 5
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    per = X.CCPlan(ctx, n).pairs(tr, pairs, W, hop)
    st = X.CCPlan(ctx, n, stack=True).pairs(tr, pairs, W, hop)
    assert st.shape == (len(pairs), 2 * n - 1)
    assert relerr(st, per.astype(np.float64).sum(axis=1)) < TOL
    assert relerr(st, R.cc_pairs(tr, pairs, n, hop, dtype=np.float64).sum(axis=1)) < TOL


def test_device_resident_padded_rows_and_batches(ctx):
    """Device pointers in/out, padded output pitch, batch sizes that do not
    divide the pair count (ragged last batch)."""
    S, n, hop, W = 6, 2500, 1250, 3
    T = hop * (W - 1) + n
    tr = R.synthetic_traces(S, T)
    pairs = R.all_pairs(S)                      # 15 pairs
    ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64)
    L = 2 * n - 1
    ld = (L + 31) // 32 * 32
    dtr = ctx.array(tr)
    for batch in (1, 4, 7, 64):
        plan = X.CCPlan(ctx, n, batch=batch)
        out = ctx.empty((len(pairs) * W, ld)).zero()
        plan.pairs(dtr, pairs, W, hop, out=out, ld_out=ld)
        got = out.rows_to_host(0, len(pairs) * W, L).reshape(len(pairs), W, L)
        assert relerr(got, ref) < TOL, batch
        pad = out.to_host()[:, L:]
        assert np.all(pad == 0)                 # padding is never written
        launches, pw = plan.counters()
        assert pw == len(pairs) * W
        plan.close()


def test_properties_full_size(ctx):
    """BASELINE size (n = 180000, N = 368640): symmetry, autocorrelation peak,
    shift theorem, linearity -- no oracle needed."""
    n = 180000
    rng = np.random.default_rng(7)
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    raw = X.CCPlan(ctx, n, normalize=False)
    nrm = X.CCPlan(ctx, n, normalize=True)
    rxy = raw.cc(x[None], y[None])[0].astype(np.float64)
    ryx = raw.cc(y[None], x[None])[0].astype(np.float64)
    scale = np.abs(rxy).max()
    assert np.abs(rxy - ryx[::-1]).max() < TOL * scale          # R_xy[k] = R_yx[-k]
    rxx = nrm.cc(x[None], x[None])[0]
    assert int(np.argmax(rxx)) == n - 1 and abs(float(rxx[n - 1]) - 1.0) < TOL
    assert np.abs(rxx - rxx[::-1]).max() < TOL
    # linearity in the first argument
    z = rng.standard_normal(n).astype(np.float32)
    rzy = raw.cc(z[None], y[None])[0].astype(np.float64)
    rsum = raw.cc((x + z)[None], y[None])[0].astype(np.float64)
    assert np.abs(rsum - (rxy + rzy)).max() < 2 * TOL * np.abs(rsum).max()
    # shift: delaying x by d samples moves the peak of cc(x_d, x) to lag +d
    d = 250
    xd = np.concatenate([np.zeros(d, np.float32), x[:-d]])
    r = nrm.cc(xd[None], x[None])[0]
    assert int(np.argmax(r)) == n - 1 + d
    # zero-lag value equals the plain dot product
    dot = float(np.dot(x.astype(np.float64), y.astype(np.float64)))
    assert abs(rxy[n - 1] - dot) < TOL * scale


def test_baseline_config_checksum(ctx):
    """configs[1] geometry on a reduced record: 16 stations, n = 180000,
    50 % overlap, 3 windows; every pair-window's zero-lag sample and energy
    against direct dot products (size-independent check of all 360 CCFs)."""
    S, n, hop, W = 16, 180000, 90000, 3
    T = hop * (W - 1) + n
    tr = R.synthetic_traces(S, T)
    pairs = R.all_pairs(S)
    L = 2 * n - 1
    ld = 360000
    plan = X.CCPlan(ctx, n)
    dtr = ctx.array(tr)
    out = ctx.empty((len(pairs) * W, ld))
    plan.pairs(dtr, pairs, W, hop, out=out, ld_out=ld)
    got = out.rows_to_host(0, len(pairs) * W, L).reshape(len(pairs), W, L)
    t64 = tr.astype(np.float64)
    for w in range(W):
        seg = t64[:, w * hop:w * hop + n]
        nrm = np.linalg.norm(seg, axis=1)
        gram = (seg @ seg.T) / np.outer(nrm, nrm)
        zero_lag = np.array([gram[i, j] for i, j in pairs])
        assert np.abs(got[:, w, n - 1] - zero_lag).max() < TOL
    # common signal with 25-sample moveout per station: peak at lag 25*(i-j)
    # (raw traces carry a DC offset, so use the demeaned plan for the peak)
    plan_d = X.CCPlan(ctx, n, demean=True)
    plan_d.pairs(dtr, pairs, W, hop, out=out, ld_out=ld)
    got = out.rows_to_host(0, len(pairs) * W, L).reshape(len(pairs), W, L)
    for p in (0, 17, 60, 119):
        i, j = pairs[p]
        assert int(np.argmax(got[p, 1])) == n - 1 + 25 * (i - j)
    # three full rows against the oracle
    for p, w in ((0, 0), (59, 1), (119, 2)):
        i, j = pairs[p]
        a = t64[i, w * hop:w * hop + n]
        b = t64[j, w * hop:w * hop + n]
        ref = R.cc_fair(a - a.mean(), b - b.mean(), dtype=np.float64)
        assert relerr(got[p, w], ref) < TOL


@pytest.mark.parametrize('env', [{'XCX_DUAL': '0'}, {'XCX_FLOW': '1'}, {'XCX_GROUPS': '0'}],
                         ids=['two-launch', 'flow-one-launch', 'tile-order'])
def test_alternative_engines_agree(ctx, env, monkeypatch):
    """The alternative kernel paths (separate I1/I2 launches, the opt-in one-launch flow kernel with
    its in-memory batch dependencies, the plain station-tile processing order) stay parity-green."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv('XCX_CHECK', '1')
    S, n, hop, W = 6, 3000, 1500, 3
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64)
    plan = X.CCPlan(ctx, n, batch=4)
    got = plan.pairs(tr, pairs, W, hop)
    assert relerr(got, ref) < TOL
    st = X.CCPlan(ctx, n, stack=True).pairs(tr, pairs, W, hop)
    assert relerr(st, ref.sum(axis=1)) < TOL
    plan.close()


def test_rccl_stack_reduce_single_rank(ctx):
    """RCCL communicator on one rank: init from a unique id, all-reduce and
    reduce-scatter of a partial stack are identities."""
    uid = X.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 1, 0)
    assert ctx.comm_info() == (1, 0)          # ncclCommCount / ncclCommUserRank of the communicator
    a = np.arange(4096, dtype=np.float32).reshape(4, 1024)
    d = ctx.array(a)
    ctx.stack_reduce(d)
    ctx.sync()
    np.testing.assert_array_equal(d.to_host(), a)
    ctx.stack_reduce(d, reduce_scatter=True)
    ctx.sync()
    np.testing.assert_array_equal(d.to_host(), a)


def test_stack_gram_many_stations_and_odd_pairs(ctx, monkeypatch):
    """Time-stack through the Gram (batched rank-W update) kernel: 11 stations
    (two 8-station tiles, ragged), pairs in arbitrary order with i > j, an
    auto-pair and a duplicate; also the two-window stack that is summed in the pair column pass."""
    S, n, hop, W = 11, 1200, 600, 5
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    rng = np.random.default_rng(5)
    pairs = np.array([(int(a), int(b)) for a, b in rng.integers(0, S, size=(40, 2))] + [(3, 3), (10, 0), (10, 0)],
                     np.int32)
    ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64).sum(axis=1)
    got = X.CCPlan(ctx, n, stack=True).pairs(tr, pairs, W, hop)
    assert got.shape == (len(pairs), 2 * n - 1)
    assert relerr(got, ref) < TOL
    ref2 = R.cc_pairs(tr[:, :hop + n], pairs, n, hop, dtype=np.float64).sum(axis=1)          # W = 2: no Gram pass
    got2 = X.CCPlan(ctx, n, stack=True).pairs(tr[:, :hop + n], pairs, 2, hop)
    assert relerr(got2, ref2) < TOL
    # with one-bit + whitening the stack is still the sum of the per-window CCFs
    kw = dict(onebit=True, demean=True, whiten=(50., 1., 10., .5))
    per = X.CCPlan(ctx, n, **kw).pairs(tr, pairs, W, hop).astype(np.float64).sum(axis=1)
    st = X.CCPlan(ctx, n, stack=True, **kw).pairs(tr, pairs, W, hop)
    assert relerr(st, per) < TOL
    # the band cannot change between the adds of one accumulation (transformed days wait for their Gram update)
    from xcorr_amd import _ffi
    acc = X.CCPlan(ctx, n, stack=True, **kw)
    acc.stack_add(tr, pairs, W, hop)
    with pytest.raises(XcxError, match='stack accumulation'):
        _ffi.check(_ffi.lib().xcx_plan_set_whiten(acc.handle, 50., 1., 5., 0.))
    got3, nw = acc.stack_finish(pairs)
    assert nw == W and relerr(got3, per) < TOL
    _ffi.check(_ffi.lib().xcx_plan_set_whiten(acc.handle, 50., 1., 5., 0.))       # fine between accumulations
    acc.close()


def test_randomised_sweep(ctx):
    """tools/fuzz_cc.py: random n / hop / lag slices / flags / pair subsets / stacking for a
    few seconds (a 90 s run covers ~5700 cases; worst error seen 1.2e-6)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'fuzz_cc', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_cc.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases, worst = mod.run(budget=6.0, seed=123, ctx=ctx)
    assert cases > 50 and worst < 1e-5


def test_randomised_call_sequences(ctx):
    """tools/fuzz_state.py: random SEQUENCES of calls on long-lived plans (another whitening band, pair list,
    window count, batch size, prepared spectra, open / abandoned accumulations) -- nothing a call leaves behind may
    leak into the next (80 s runs: 21 000 operations, worst error 5.5e-6)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'fuzz_state', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_state.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ops_done, worst = mod.run(budget=6.0, seed=321, ctx=ctx)
    assert ops_done > 100 and worst < 1e-5


def test_clipped_plans_use_shorter_transforms(ctx):
    """N >= n + max|lag| is enough for exact retained lags (clip_lag of core/init.py:176-209):
    a +-600 s slice of a 3600 s window runs at N = 245760 instead of 368640."""
    from xcorr_amd.engine import CCPlan
    n = 180000
    full = CCPlan(ctx, n)
    assert full.N == 368640
    clip = CCPlan(ctx, n, n - 1 - 30000, n + 30000)
    assert clip.N == 245760 and clip.L == 60001
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, n)).astype(np.float32)
    y = (np.roll(x, 12345, axis=-1) + 0.3 * rng.standard_normal((2, n))).astype(np.float32)
    got = clip.cc(x, y)
    ref = full.cc(x, y)[:, n - 1 - 30000:n + 30000]
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    one_sided = CCPlan(ctx, n, n - 1, n + 5000)            # lags 0 .. 5000
    assert one_sided.N < 245760
    got1 = one_sided.cc(x, y)
    assert np.abs(got1 - full.cc(x, y)[:, n - 1:n + 5000]).max() <= 2e-6 * np.abs(ref).max()
    for p in (full, clip, one_sided):
        p.close()


def test_stack_plan_reuse_keeps_descriptors_consistent(ctx):
    """one stacking plan reused with W=3 (Gram path), W=1 (no Gram),
    W=3 again, and with xcx_plan_set_batch between calls -- descriptors and the Gram row
    map must be rebuilt together (stale rows read the wrong cross-spectra / out of bounds)."""
    from xcorr_amd import _ffi
    S, n, hop = 7, 2000, 1000
    tr = R.synthetic_traces(S, hop * 2 + n)
    pairs = R.all_pairs(S)
    per = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64)          # [P, 3, L]
    ref3, ref1 = per.sum(axis=1), per[:, 0]
    plan = X.CCPlan(ctx, n, stack=True)
    assert relerr(plan.pairs(tr, pairs, 3, hop), ref3) < TOL
    assert relerr(plan.pairs(tr, pairs, 1, hop), ref1) < TOL
    assert relerr(plan.pairs(tr, pairs, 3, hop), ref3) < TOL
    _ffi.check(_ffi.lib().xcx_plan_set_batch(plan.handle, 5))
    assert relerr(plan.pairs(tr, pairs, 3, hop), ref3) < TOL
    sub = pairs[::2]                                               # another pair list on the same plan
    assert relerr(plan.pairs(tr, sub, 3, hop), ref3[::2]) < TOL
    assert relerr(plan.pairs(tr, pairs, 3, hop), ref3) < TOL
    plan.close()


def test_stack_add_finish_equals_sum_of_days(ctx):
    """xcx_cc1d_stack_add / _finish: cross-spectra of several calls ("days", different W)
    accumulate in the frequency domain; the result is the sum of the per-day stacks."""
    S, n, hop = 6, 2500, 1250
    pairs = np.array([(0, 1), (4, 2), (3, 3), (0, 5), (4, 2), (1, 5)], np.int32)   # duplicate + auto pair
    days = [(R.synthetic_traces(S, hop * (W - 1) + n, day=d), W) for d, W in enumerate((3, 1, 4))]
    ref = sum(R.cc_pairs(tr, pairs, n, hop, dtype=np.float64).sum(axis=1) for tr, _ in days)
    plan = X.CCPlan(ctx, n, stack=True)
    for k, (tr, W) in enumerate(days):
        plan.stack_add(ctx.array(tr) if k == 1 else tr, pairs, W, hop)    # device and host traces
    with pytest.raises(XcxError):
        plan.pairs(days[0][0], pairs, 3, hop)                     # refused while accumulating
    with pytest.raises(XcxError):
        plan.stack_add(days[0][0], pairs[:3], 3, hop)             # another pair list
    got, nw = plan.stack_finish(pairs)
    assert nw == 8 and relerr(got, ref) < TOL
    with pytest.raises(XcxError):
        plan.stack_finish(pairs)                                  # accumulator is empty again
    # an abandoned accumulation (driver error between adds): reset, then the plan works again
    plan.stack_add(days[0][0], pairs, days[0][1], hop)
    plan.stack_reset()
    with pytest.raises(XcxError):
        plan.stack_finish(pairs)
    one = plan.pairs(days[1][0], pairs, days[1][1], hop)
    assert relerr(one, R.cc_pairs(days[1][0], pairs, n, hop, dtype=np.float64).sum(axis=1)) < TOL
    # padded device output; the plan is reusable for a plain stacked call afterwards
    L = 2 * n - 1
    ld = (L + 31) // 32 * 32
    out = ctx.empty((len(pairs), ld)).zero()
    for tr, W in days[:2]:
        plan.stack_add(tr, pairs, W, hop)
    plan.stack_finish(pairs, out=out, ld_out=ld)
    ref2 = sum(R.cc_pairs(tr, pairs, n, hop, dtype=np.float64).sum(axis=1) for tr, _ in days[:2])
    assert relerr(out.rows_to_host(0, len(pairs), L), ref2) < TOL
    assert np.all(out.to_host()[:, L:] == 0)
    st = plan.pairs(days[2][0], pairs, 4, hop)
    assert relerr(st, R.cc_pairs(days[2][0], pairs, n, hop, dtype=np.float64).sum(axis=1)) < TOL
    with pytest.raises(XcxError):
        X.CCPlan(ctx, n).stack_add(days[0][0], pairs, 3, hop)     # not a stacking plan
    plan.close()


def test_output_buffers_are_validated(ctx):
    """an undersized / wrong-typed `out` is refused before the library writes."""
    S, n, hop, W = 3, 1000, 500, 2
    tr = R.synthetic_traces(S, hop * (W - 1) + n)
    pairs = R.all_pairs(S)
    plan = X.CCPlan(ctx, n)
    with pytest.raises(ValueError):
        plan.pairs(tr, pairs, W, hop, out=np.empty((len(pairs), W, plan.L - 1), np.float32))
    with pytest.raises(ValueError):
        plan.pairs(tr, pairs, W, hop, out=np.empty((len(pairs), W, plan.L), np.float64))
    with pytest.raises(ValueError):
        plan.pairs(ctx.array(tr), pairs, W, hop, out=ctx.empty((len(pairs) * W - 1, plan.L)))
    with pytest.raises(ValueError):
        plan.cc(tr[:1, :n], tr[1:2, :n], out=np.empty((1, plan.L))[:, ::2])
    plan.close()


def test_tiny_windows_single_station(ctx):
    """Edges of the pair / window API: windows of 1, 2, 3 and 5 samples, a single station (auto-pair only), a
    single window, raw / normalised / stacked -- and the two-input form cc(x, y) at the same sizes."""
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 5):
        for S, W in ((1, 1), (2, 1), (3, 2)):
            hop = max(1, n // 2)
            tr = rng.standard_normal((S, (W - 1) * hop + n)).astype(np.float32) + 0.5
            pairs = np.array([(i, j) for i in range(S) for j in range(i, S)], np.int32)
            for kw in (dict(normalize=True), dict(normalize=False), dict(normalize=True, stack=True)):
                plan = X.CCPlan(ctx, n, **kw)
                got = plan.pairs(tr, pairs, W, hop)
                ref = R.cc_pairs(tr, pairs, n, hop, normalize=kw['normalize'], dtype=np.float64)
                if kw.get('stack'):
                    ref = ref.sum(axis=1)
                assert got.shape == ref.shape and relerr(got, ref) < TOL, (n, S, W, kw)
                plan.close()
            x, y = tr[:1, :n], tr[-1:, :n]
            assert np.abs(X.CCPlan(ctx, n).cc(x, y)[0] - R.cc(x[0], y[0], normalize=True, dtype=np.float64)).max() < TOL


def test_q24_storage_of_the_intermediate_on_whitened_plans(ctx, monkeypatch):
    """The pair pass of a WHITENED plan keeps its intermediate T as 2 x 24-bit fixed point (include/xcx.h,
    xcx_plan_t_storage; wave_kernels.h): the scale comes from the bound |T| <= scale^2 sum gain^2, so nothing can
    overflow -- not even identical traces, which reach the bound -- and the arithmetic stays float32.  What the
    24-bit intermediate itself adds is bounded against float2 storage ON THE SAME INPUTS: <= 2.5e-6 of max|cc|, the
    worst case being a CCF that is pure noise (profiles/r05_t24_error_table.txt: 2.0e-6); both storages meet the
    suite's 1e-5 against the float64 oracle (a one-bit window with a nearly empty in-band bin costs either of them
    several 1e-6: float32 whitening, not storage).  Plans without whitening, stacked plans and column lengths
    without the lane-pair store keep float2."""
    rng = np.random.default_rng(77)
    for n, band in ((180000, (50., .1, 10., .05)), (180000, (50., 1., 20., .5)), (60000, (50., .1, 10., .05)),
                    (90000, (50., .5, 5., .1))):
        S, W, hop = 3, 2, n // 2
        T = hop * (W - 1) + n
        tr = rng.standard_normal((S, T)).astype(np.float32)
        tr[2] = tr[0]                                              # pair (0, 2): identical traces, T at its bound
        pairs = np.array([(0, 1), (0, 2), (1, 1)], np.int32)
        res = {}
        for ts in ('f32', 'q24'):
            monkeypatch.setenv('XCX_T_STORAGE', ts)
            plan = X.CCPlan(ctx, n, whiten=band, onebit=True, demean=True)
            assert plan.t_storage == ts
            res[ts] = np.asarray(plan.pairs(tr, pairs, W, hop), np.float64)
            N = plan.N
            plan.close()
        g = R.whiten_weights(N, *band)
        ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=N, onebit_=True, demean_=True)
        for p in range(len(pairs)):
            for w in range(W):
                m = np.abs(ref[p, w]).max()
                assert np.abs(res['f32'][p, w] - ref[p, w]).max() / m < TOL, (n, band, p, w)
                assert np.abs(res['q24'][p, w] - ref[p, w]).max() / m < TOL, (n, band, p, w)
                assert np.abs(res['q24'][p, w] - res['f32'][p, w]).max() / m < 2.5e-6, (n, band, p, w)
        assert not np.array_equal(res['f32'], res['q24'])          # the storage really differed
    monkeypatch.delenv('XCX_T_STORAGE')
    assert X.CCPlan(ctx, 180000, whiten=(50., .1, 10., .05)).t_storage == 'q24'          # the default of whitened plans
    assert X.CCPlan(ctx, 180000).t_storage == 'f32'                                       # no whitening: no bound
    assert X.CCPlan(ctx, 180000, whiten=(50., .1, 10., .05), stack=True).t_storage == 'f32'   # stacks read a Gram matrix
    assert X.CCPlan(ctx, 30000, whiten=(50., 1., 10., .5)).t_storage == 'f32'             # N1 = 256: no lane-pair store


def _zskip_expected(N, N1, band_hi):
    """Restatement of zskip_need / zskip_pick (csrc/wave_kernels.h): the instantiated symmetric block range inside the
    zero stretch of EVERY column of a band-limited spectrum."""
    N2 = N // N1
    if band_hi >= N:
        return 0
    za = max((band_hi - k2) // N2 + 1 if band_hi >= k2 else 0 for k2 in range(N2 // 2 + 1))
    zb = min((N - band_hi - k2 + N2 - 1) // N2 - 1 for k2 in range(N2 // 2 + 1))
    nb = N1 // 64
    first, last = -(-za // 64), (zb + 1) // 64 - 1
    need = max(first, nb - 1 - last)
    if need < 1 or 2 * need >= nb:
        return 0
    for v in {1536: (3, 5, 7, 9), 1024: (2, 4, 6), 768: (2, 3, 4), 512: (1, 2)}.get(N1, ()):
        if v >= need and 2 * v < nb:
            return v
    return 0


def test_zero_stretch_skip_of_band_limited_columns(ctx, monkeypatch):
    """q24 plans do not load the 64-point blocks of a spectrum column that the whitening band leaves zero in EVERY
    column (xcx_plan_zero_skip; template parameter ZB0 of the column body): every instantiated range of every column
    length is reached through a band chosen for it, the plan reports the range the restatement above expects, and the
    results meet the oracle like float2 storage (whose kernel loads everything) does -- the skipped products are
    0 * x, so nothing but the q24 rounding may differ."""
    rng = np.random.default_rng(5)
    seen = set()
    cases = [(180000, (50., .1, f, .05)) for f in (5.9, 10., 14., 18.5, 24.)] + \
            [(120000, (50., .1, f, .05)) for f in (5., 11., 17.)] + \
            [(90000, (50., .1, f, .05)) for f in (7., 11., 15.)] + \
            [(60000, (50., .1, f, .05)) for f in (5., 11.)]
    for n, band in cases:
        S, W, hop = 3, 2, n // 2
        tr = rng.standard_normal((S, hop * (W - 1) + n)).astype(np.float32)
        pairs = np.array([(0, 1), (1, 2), (2, 2)], np.int32)
        res = {}
        for ts in ('f32', 'q24'):
            monkeypatch.setenv('XCX_T_STORAGE', ts)
            plan = X.CCPlan(ctx, n, whiten=band, onebit=True, demean=True)
            N, N1 = plan.N, plan.N1
            zs = plan.zero_skip
            if ts == 'f32':
                assert zs[0] == 0
            else:
                band_hi = int(np.floor((band[2] + band[3]) * N / band[0]))      # the last bin the gain can be non-zero in
                assert zs == (_zskip_expected(N, N1, band_hi), N1 // 64), (n, band, zs)
                seen.add((N1, zs[0]))
            res[ts] = np.asarray(plan.pairs(tr, pairs, W, hop), np.float64)
            plan.close()
        g = R.whiten_weights(N, *band)
        ref = R.cc_pairs(tr, pairs, n, hop, dtype=np.float64, whiten=g, nfft=N, onebit_=True, demean_=True)
        for p in range(len(pairs)):
            for w in range(W):
                m = np.abs(ref[p, w]).max()
                assert np.abs(res['q24'][p, w] - ref[p, w]).max() / m < TOL, (n, band, p, w)
                assert np.abs(res['q24'][p, w] - res['f32'][p, w]).max() / m < 2.5e-6, (n, band, p, w)
    monkeypatch.delenv('XCX_T_STORAGE')
    # every instantiated range was exercised (and a band too wide for any: the generic kernel)
    assert {(1536, 3), (1536, 5), (1536, 7), (1536, 9), (1536, 0), (1024, 2), (1024, 4), (1024, 6),
            (768, 2), (768, 3), (768, 4), (512, 1), (512, 2)} <= seen, sorted(seen)


def test_stack_finish_reduce_in_the_frequency_domain(ctx, monkeypatch):
    """configs[3] across ranks (xcx_cc1d_stack_finish_reduce): the ranks' cross-spectrum accumulators are summed with
    one reduce-scatter of their IN-BAND elements and every rank inverse-transforms only its share of the pairs.
    One GPU can check (a) the whole path through RCCL on a one-rank communicator: bit for bit what
    xcx_cc1d_stack_finish gives, all pairs owned; (b) the partition for 3 and 5 ranks without RCCL
    (XCX_STACK_REDUCE_FAKE: the other ranks contribute nothing): every pair is owned by exactly one rank, a rank's
    rows equal the plain finish, rows it does not own are never written.  Band-limited and full-band plans, a pair
    list in arbitrary order with a duplicate."""
    S, n, hop, W = 9, 2400, 1200, 4
    tr = [R.synthetic_traces(S, hop * (W - 1) + n, day=d) for d in range(2)]
    pairs = R.all_pairs(S)[::-1].copy()
    pairs = np.concatenate([pairs, pairs[3:4]])                    # a duplicate shares its accumulator row
    P = len(pairs)
    own_ctx = X.Context(0)
    own_ctx.comm_init(X.Context.comm_unique_id(), 1, 0)
    try:
        for kw in (dict(whiten=(50., 1., 10., .5), onebit=True, demean=True), dict()):
            plan = X.CCPlan(own_ctx, n, stack=True, **kw)
            L = plan.L

            def accumulate():
                for x in tr:
                    plan.stack_add(x, pairs, W, hop)
            accumulate()
            ref = own_ctx.empty((P, L))
            plan.stack_finish(pairs, out=ref)
            want = ref.to_host()
            g = R.whiten_weights(plan.N, *kw['whiten']) if kw else None
            orc = sum(R.cc_pairs(x, pairs, n, hop, dtype=np.float64, nfft=plan.N, whiten=g, onebit_=bool(kw),
                                 demean_=bool(kw)).sum(axis=1) for x in tr)
            assert relerr(want, orc) < TOL
            accumulate()                                            # (a) through RCCL, one rank
            out = own_ctx.empty((P, L))
            owned, nw = plan.stack_finish_reduce(pairs, out)
            assert owned.all() and nw == 2 * W
            assert np.array_equal(out.to_host(), want)
            for world in (3, 5):                                    # (b) the partition, no RCCL
                count = np.zeros(P, int)
                for rank in range(world):
                    monkeypatch.setenv('XCX_STACK_REDUCE_FAKE', f'{world},{rank}')
                    accumulate()
                    out.copy_from(np.full((P, L), np.nan, np.float32))
                    owned, _ = plan.stack_finish_reduce(pairs, out)
                    got = out.to_host()
                    assert np.array_equal(got[owned], want[owned])
                    assert np.isnan(got[~owned]).all()
                    count += owned
                monkeypatch.delenv('XCX_STACK_REDUCE_FAKE')
                assert (count == 1).all()
            plan.close()
        bare = X.Context(0)                                         # a context without a communicator refuses
        try:
            plain = X.CCPlan(bare, n, stack=True)
            plain.stack_add(tr[0], pairs, W, hop)
            with pytest.raises(XcxError, match='xcx_comm_init'):
                plain.stack_finish_reduce(pairs, bare.empty((P, plain.L)))
            plain.stack_reset()
            plain.close()
        finally:
            bare.close()
    finally:
        own_ctx.close()
