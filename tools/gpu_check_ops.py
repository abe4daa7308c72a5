"""Dev aid: ops (row-wise signal ops + SOS filters) against the oracle."""
import sys
import numpy as np
import scipy.signal
sys.path.insert(0, '.')
from oracle import ref_numpy as R
import xcorr_amd as X
from xcorr_amd import ops

ctx = X.Context(0)
rng = np.random.default_rng(1)
ok = True

def rep(name, got, ref, tol=1e-5):
    global ok
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30)
    good = err < tol
    ok &= good
    print(('OK ' if good else 'BAD'), name, f'relerr={err:.3e}', flush=True)

x = (rng.standard_normal((7, 5001)) + 3.0 + 1e-3 * np.arange(5001)).astype(np.float32)
rep('norm1d', ops.norm1d(ctx, x), R.norm1d(x))
rep('demean', ops.demean(ctx, x), R.demean(x))
rep('detrend', ops.detrend_linear(ctx, x), R.detrend_linear(x), 2e-5)
rep('taper hann', ops.taper(ctx, x, 250), R.taper(x, 50.0, max_length=5.0))
rep('taper cosine', ops.taper(ctx, x, 250, 'cosine'), x * R.cosine_taper_obspy(5001, 50.0, 0.05).astype(np.float32))
rep('onebit', ops.onebit(ctx, x - 3.5), R.onebit(x - 3.5))
n = 2501
c = rng.standard_normal((4, 2 * n - 1)).astype(np.float32)
rep('unbias', ops.unbias(ctx, c[:, 100:4000], n, 100, 4000) / R.weight_full(n)[100:4000], c[:, 100:4000])

for (order, freq, btype) in [(2, 3.0, 'highpass'), (4, 1.5, 'highpass'), (4, (0.1, 10.0), 'bandpass'), (4, 20.0, 'lowpass')]:
    sos = R.butter_sos(order, freq, btype, 50.0)
    rep(f'sosfiltfilt {btype} o{order} rows', ops.sosfiltfilt(ctx, x, sos), R.sosfiltfilt(sos, x))
    rep(f'sosfilt {btype} o{order} rows', ops.sosfilt(ctx, x, sos), R.sosfilt(sos, x))
# few long rows -> chunked scan path
xl = (rng.standard_normal((3, 300000)) + 1.0).astype(np.float32)
for (order, freq, btype) in [(4, (0.1, 10.0), 'bandpass'), (2, 0.05, 'highpass')]:
    sos = R.butter_sos(order, freq, btype, 50.0)
    rep(f'sosfilt long {btype}', ops.sosfilt(ctx, xl, sos), R.sosfilt(sos, xl))
    rep(f'sosfiltfilt long {btype}', ops.sosfiltfilt(ctx, xl, sos), R.sosfiltfilt(sos, xl))
# many rows
xm = rng.standard_normal((300, 999)).astype(np.float32)
sos = R.butter_sos(2, 3.0, 'highpass', 50.0)
rep('sosfiltfilt many rows', ops.sosfiltfilt(ctx, xm, sos), R.sosfiltfilt(sos, xm))
print('ALL OK' if ok else 'FAILURES')
sys.exit(0 if ok else 1)
