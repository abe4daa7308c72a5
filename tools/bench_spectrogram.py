"""Time xcx_spectrogram (duration 2.5 s @ 50 Hz, padding 4: win 125, nfft 500, 251 bins)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal as ss
from xcorr_amd import ops
from xcorr_amd.engine import default_context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8191
ctx = default_context()
rng = np.random.default_rng(0)
x = rng.standard_normal((B, n)).astype(np.float32)
d = ctx.array(x)
w = ss.get_window(('tukey', .25), 125)
scale = 1.0 / (50. * (w * w).sum())
S = ops.spectrogram(ctx, d, 125, 500, w, scale); ctx.sync()
from xcorr_amd import _ffi
from xcorr_amd._ffi import ptr, check
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    check(_ffi.lib().xcx_spectrogram(ctx.handle, ptr(d), ptr(S), B, n, 125, 500, ptr(np.ascontiguousarray(w)), scale, 1, 1))
    ctx.sync(); dt = time.perf_counter() - t0
    flops = 2.0 * 512 * 128 * (n - 124) * B
    print(f'B={B} n={n}: {dt*1e3:.2f} ms  {B/dt:.0f} rows/s  {flops/dt/1e12:.1f} TFLOP/s (padded GEMM)  '
          f'{B*251*n*4/dt/1e12:.2f} TB/s written', flush=True)
if '--cpu' in sys.argv:
    t0 = time.perf_counter()
    ss.spectrogram(x[:2].astype(np.float64), fs=50., nperseg=125, noverlap=124, nfft=500, mode='psd')
    print(f'scipy float64: {(time.perf_counter()-t0)/2*1e3:.1f} ms per row')
