"""GPU: two-dimensional cross-correlation (xcorr/signal/correlate.py:162-291)
against the float64 oracle; tolerance 1e-5 of max|cc2| (float32 path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R          # noqa: E402
from xcorr_amd import ops, signal          # noqa: E402
from xcorr_amd._ffi import XcxError        # noqa: E402

TOL = 1e-5


def tiles_of(T, F, L, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((F, L))
    out = []
    for t in range(T):
        out.append(np.roll(base, (t, 3 * t), axis=(0, 1)) + 0.3 * rng.standard_normal((F, L)))
    return np.abs(np.asarray(out)).astype(np.float32)      # spectrogram-like: non-negative


@pytest.mark.parametrize('F,L', [(1, 1), (1, 7), (5, 1), (3, 4), (8, 16), (9, 33), (17, 100), (40, 129),
                                 (64, 1000), (100, 257), (129, 64)])
@pytest.mark.parametrize('normalize', [True, False])
def test_full_output_matches_oracle(ctx, F, L, normalize):
    t = tiles_of(3, F, L, seed=F * 1000 + L)
    pairs = [(0, 1), (2, 0), (1, 1)]
    peak, idx, full = ops.correlate2d(ctx, t, pairs, normalize=normalize, full=True)
    assert full.shape == (3, 2 * F - 1, 2 * L - 1)
    for p, (a, b) in enumerate(pairs):
        ref = R.correlate2d(t[a], t[b], normalize=normalize)
        scale = np.abs(ref).max()
        assert np.abs(full[p] - ref).max() <= TOL * scale, (F, L, p)
        # reported peak is the maximum of the output the GPU produced itself
        assert peak[p] == full[p].max()
        assert tuple(idx[p]) == np.unravel_index(np.argmax(full[p]), full[p].shape)
        assert abs(peak[p] - ref.max()) <= TOL * scale


def test_definition_small(ctx):
    t = tiles_of(2, 4, 6, seed=5)
    _, _, full = ops.correlate2d(ctx, t, [(0, 1)], normalize=False, full=True)
    ref = R.correlate2d_direct(t[0], t[1])
    assert np.abs(full[0] - ref).max() <= TOL * np.abs(ref).max()


def test_autocorrelation_peak_and_known_shift(ctx):
    F, L = 32, 400
    rng = np.random.default_rng(2)
    big = rng.standard_normal((F + 8, L + 40))
    a = big[4:4 + F, 20:20 + L]
    b = big[4 + 3:4 + 3 + F, 20 - 11:20 - 11 + L]           # b[u, v] = a[u + 3, v - 11]
    t = np.stack([a, b]).astype(np.float32)
    peak, idx = ops.correlate2d(ctx, t, [(0, 0), (0, 1), (1, 0)])
    assert abs(peak[0] - 1.0) < 1e-5 and tuple(idx[0]) == (F - 1, L - 1)
    # cc[i, j] = sum a[u + di, v + dj] b[u, v] peaks at (di, dj) = (3, -11)
    assert tuple(idx[1]) == (F - 1 + 3, L - 1 - 11)
    assert tuple(idx[2]) == (F - 1 - 3, L - 1 + 11)
    assert peak[1] == pytest.approx(peak[2], rel=1e-5)


def test_timelapse_tile_size_peak(ctx):
    """BASELINE config 5 shape: 256 x 4096 spectrogram tiles; numpy finishes one
    pair in a few seconds."""
    F, L = 256, 4096
    t = tiles_of(3, F, L, seed=9)
    peak, idx = ops.correlate2d(ctx, t, [(0, 1), (0, 2), (2, 1)])
    for p, (a, b) in enumerate([(0, 1), (0, 2)]):
        ref = R.correlate2d(t[a], t[b])
        assert abs(peak[p] - ref.max()) <= TOL * np.abs(ref).max()
        ri = np.unravel_index(np.argmax(ref), ref.shape)
        assert ref[tuple(idx[p])] >= ref.max() - 2 * TOL * np.abs(ref).max()
        assert tuple(idx[p]) == ri
    # antisymmetry of the index map: cc(b, a)[i, j] = cc(a, b)[2F-2-i, 2L-2-j]
    p2, i2 = ops.correlate2d(ctx, t, [(1, 0)])
    assert p2[0] == pytest.approx(peak[0], rel=2e-5)
    assert tuple(i2[0]) == (2 * F - 2 - idx[0][0], 2 * L - 2 - idx[0][1])


def test_device_resident_and_many_pairs(ctx):
    F, L = 16, 200
    t = tiles_of(12, F, L, seed=3)
    pairs = [(i, j) for i in range(12) for j in range(12)]
    d = ctx.array(t)
    peak, idx, full = ops.correlate2d(ctx, d, pairs, full=True)
    host = full.to_host() if hasattr(full, 'to_host') else full.numpy()
    peak_h, idx_h = ops.correlate2d(ctx, t, pairs)
    assert np.array_equal(peak, peak_h) and np.array_equal(idx, idx_h)
    # the pair pass runs in tile-group order (4 x 4 tile blocks): every map and peak lands at the caller's index
    for p, (a, b) in enumerate(pairs):
        ref = R.correlate2d(t[a], t[b])
        assert np.abs(host[p] - ref).max() <= TOL * np.abs(ref).max(), (p, a, b)
        assert abs(peak[p] - ref.max()) <= TOL * abs(ref.max())
        assert tuple(idx[p]) == np.unravel_index(int(np.argmax(host[p])), host[p].shape)


def test_pair_list_over_several_passes():
    """The same with the intermediate capped at 1 MiB (XCX_CC2D_PASS_MB, read once per process): one tile group
    per pass, shuffled and repeated pairs -- the slot tables of all passes are uploaded before the first launch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import numpy as np
from oracle import ref_numpy as R
import xcorr_amd as X
from xcorr_amd import ops
rng = np.random.default_rng(8)
ctx = X.Context(0)
T, F, L = 11, 16, 200
t = np.abs(rng.standard_normal((T, F, L))).astype(np.float32)
pairs = [(int(a), int(b)) for a, b in rng.integers(0, T, (150, 2))]
peak, idx, full = ops.correlate2d(ctx, t, pairs, full=True)
worst = 0.0
for p, (a, b) in enumerate(pairs):
    ref = R.correlate2d(t[a], t[b])
    worst = max(worst, float(np.abs(full[p] - ref).max() / np.abs(ref).max()))
    assert tuple(idx[p]) == np.unravel_index(int(np.argmax(full[p])), full[p].shape), p
print('WORST', worst)
assert worst < 1e-5
'''
    env = dict(os.environ, XCX_CC2D_PASS_MB='1', PYTHONPATH=root)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-1500:]


def test_randomised_peak_positions():
    """tools/check_cc2d.py: peak value and POSITION against numpy's arg-max of the oracle map over random tile shapes on
    either side of the padded sizes (the row kernel keeps a running maximum by value and looks the position up only
    in the threads that hold it; the padding columns L <= u <= P2 - L must never win), all-negative maps, peak-only
    and full-map kernels.  A short run in this process (two passes in flight on the pipeline streams where the pair
    list needs several) and one with a single pass at a time, 1 MiB passes (XCX_CC2D_OVERLAP=0)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import check_cc2d
    cases, worst, _ = check_cc2d.run(6.0, seed=5)
    assert cases > 20 and worst <= 1e-5
    env = dict(os.environ, XCX_CC2D_PASS_MB='1', XCX_CC2D_OVERLAP='0', PYTHONPATH=root)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_cc2d.py'), '5', '6'], capture_output=True,
                         text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0 and 'calls clean' in out.stdout, out.stdout[-500:] + out.stderr[-1500:]


def test_signal_mirror_and_errors(ctx):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 3, 10, 50)).astype(np.float32)
    y = rng.standard_normal((2, 3, 10, 50)).astype(np.float32)
    cc2 = signal.correlate2d(x, y)
    assert cc2.shape == (2, 3, 19, 99) and cc2.dtype == np.float32
    ref = R.correlate2d(x, y)
    assert np.abs(cc2 - ref).max() <= TOL * np.abs(ref).max()
    pk, ix = signal.correlate2d_peak(x, y)
    assert pk.shape == (2, 3) and ix.shape == (2, 3, 2)
    assert np.array_equal(pk, cc2.max(axis=(-2, -1)))
    n2 = signal.norm2d(x)
    assert np.abs(n2 - R.norm2d(x)).max() <= TOL * np.abs(R.norm2d(x)).max()
    with pytest.raises(ValueError):
        signal.correlate2d(x, y[:, :, :, :49])
    with pytest.raises(XcxError):
        ops.correlate2d(ctx, np.zeros((1, 257, 8), np.float32), [(0, 0)])
    with pytest.raises(XcxError):
        ops.correlate2d(ctx, np.zeros((1, 8, 4097), np.float32), [(0, 0)])
    with pytest.raises(XcxError):
        ops.correlate2d(ctx, np.ones((2, 8, 8), np.float32), [(0, 2)])


def test_timelapse_data_path(ctx):
    """scripts/timelapse.py:196-268 for one pair: spectrogram of every time, then per band and time
    pair the maximum of correlate2d and its (delta_freq, delta_lag), against scipy / numpy."""
    from xcorr_amd.core import timelapse
    rng = np.random.default_rng(17)
    fs, T, L = 50., 4, 601
    t = (np.arange(L) - L // 2) / fs
    cc = np.stack([0.02 * rng.standard_normal(L) + np.exp(-0.5 * ((t - 0.04 * k) / 1.5) ** 2) * np.cos(2 * np.pi * (4. + 0.1 * k) * t)
                   for k in range(T)]).astype(np.float32)
    freq, bw = np.array([4., 8.]), np.array([4., 4.])
    res = timelapse.correlate_spectrograms(cc, fs, freq, bw, duration=0.64, padding_factor=2, ctx=ctx)
    f, psd = R.spectrogram(cc, fs, 0.64, 2)
    assert np.allclose(res['freq'], f) and res['cc2'].shape == (2, 6) and (res['status'] == 1).all()
    df = f[1] - f[0]
    for b in range(2):
        sel = (f >= freq[b] - bw[b] / 2) & (f < freq[b] + bw[b] / 2)
        for k, (i, j) in enumerate(res['pairs']):
            full = R.correlate2d(psd[i][sel], psd[j][sel], normalize=True)
            a0, a1 = np.unravel_index(np.argmax(full), full.shape)
            assert abs(res['cc2'][b, k] - full.max()) <= 1e-5 * abs(full.max())
            assert np.isclose(res['delta_freq'][b, k], (a0 - (sel.sum() - 1)) * df)
            assert np.isclose(res['delta_lag'][b, k], (a1 - (L - 1)) / fs)
    empty = timelapse.correlate_spectrograms(cc, fs, [24.9], [0.01], duration=0.64, padding_factor=2, ctx=ctx)
    assert (empty['status'] == 0).all()
