"""Device-resident timing of the entry points no other tool times: the cc(x, y) / correlate1d row form, snr, the
any-hop spectrogram, Lanczos interpolation and Fourier resampling of day-long rows, zero-phase filtering."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
import xcorr_amd as X
from xcorr_amd import ops
from xcorr_amd.engine import default_context
ctx = default_context()
rng = np.random.default_rng(0)


def t(name, fn, unit, nunits, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:46s} {dt*1e3:9.3f} ms  {dt/nunits*1e6:9.2f} us/{unit}', flush=True)


n, rows = 180000, 1024
x = ctx.array(rng.standard_normal((rows, n), dtype=np.float32))
y = ctx.array(rng.standard_normal((rows, n), dtype=np.float32))
plan = X.CCPlan(ctx, n, normalize=True)
out = ctx.empty((rows, plan.L))
t('cc(x, y) rows [1024][180000] full lag', lambda: plan.cc(x, y, out=out), 'row', rows)
L = 2 * n - 1
c = out
sig = np.zeros(L, bool); sig[n - 3000:n + 3000] = True
noi = np.zeros(L, bool); noi[:50000] = True
t('snr [1024][359999]', lambda: ops.snr(ctx, c, sig, noi), 'row', rows)
w = scipy.signal.get_window(('tukey', .25), 125)
cs = ctx.array(rng.standard_normal((64, 8191), dtype=np.float32))
t('spectrogram hop 1 [64][8191] (MFMA)', lambda: ops.spectrogram(ctx, cs, 125, 500, w, 1.0 / (50. * (w * w).sum())), 'row', 64)
t('spectrogram_seg hop 25 [64][8191]', lambda: ops.spectrogram_seg(ctx, cs, 125, 25, 500, w, 1.0 / (50. * (w * w).sum())), 'row', 64)
T = 4320000
d = ctx.array(rng.standard_normal((16, T), dtype=np.float32))
sos = scipy.signal.butter(4, [0.1, 10.], 'bandpass', output='sos', fs=50.)
t('sosfilt_zerophase [16][4320000]', lambda: ops.sosfilt_zerophase(ctx, d, sos), 'row', 16)
t('sosfiltfilt [16][4320000]', lambda: ops.sosfiltfilt(ctx, d, sos), 'row', 16)
t('lanczos 50 -> 20 Hz [16][4320000]', lambda: ops.lanczos(ctx, d, int(T / 2.5) - 10, 0.0, 2.5, 20), 'row', 16)
from xcorr_amd import stream
t('stream.resample 50 -> 20 Hz [16][4320000]', lambda: stream.resample(d, 50., 20., ctx=ctx), 'row', 16, reps=1)
t('stream.decimate 5 [16][4320000]', lambda: stream.decimate(d, 50., 5, ctx=ctx), 'row', 16)
