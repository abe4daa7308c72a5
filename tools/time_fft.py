"""Cost of the any-length labelled transforms (a6: xcorr/signal/fft.py:357-374, 475-494; hilbert.py:69) on the axes the
post-processing scripts use: the 359 998-sample lag axis of a full-lag CCF (2 x 179 999, a prime) and a day-long
4 320 000-sample row.  Device-resident rows (on_device = 1): the transform itself, no PCIe.
usage: python tools/time_fft.py [rows]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd import _ffi
from xcorr_amd._ffi import check
from xcorr_amd.engine import default_context, ptr

ctx = default_context()
lib = _ffi.lib()


def run(name, fn, rows, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:34s} {dt * 1e3:9.2f} ms  {dt / rows * 1e6:9.2f} us/row', flush=True)


def case(n, rows):
    rng = np.random.default_rng(0)
    x = ctx.array(rng.standard_normal((rows, n)).astype(np.float32))
    m = n // 2 + 1
    X = ctx.empty((rows, 2 * m))                      # complex64 as float pairs
    y = ctx.empty((rows, 2 * (m - 1)))
    z = ctx.empty((rows, 2 * n))
    e = ctx.empty((rows, n))
    tag = f'n={n}'
    run(f'rfft     {tag}', lambda: check(lib.xcx_rfft(ctx.handle, ptr(x), ptr(X), rows, n, 1)), rows)
    run(f'irfft    {tag}', lambda: check(lib.xcx_irfft(ctx.handle, ptr(X), ptr(y), rows, m, 1)), rows)
    run(f'hilbert  {tag}', lambda: check(lib.xcx_hilbert(ctx.handle, ptr(x), ptr(z), rows, n, 0, 1)), rows)
    run(f'envelope {tag}', lambda: check(lib.xcx_hilbert(ctx.handle, ptr(x), ptr(e), rows, n, 1, 1)), rows)
    for b in (x, X, y, z, e):
        b.free()


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
case(359998, rows)                                    # lag axis of the headline workload (postprocess drops one lag)
case(360000, rows)                                    # a 5-smooth neighbour, for comparison
case(4320000, 8)                                      # day-long rows
