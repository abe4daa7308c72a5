import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from xcorr_amd import _ffi
from xcorr_amd._ffi import check
from xcorr_amd.engine import default_context, ptr
ctx = default_context(); lib = _ffi.lib()
rows, n = 128, 359998
x = ctx.array(np.random.default_rng(0).standard_normal((rows, n)).astype(np.float32))
m = n // 2 + 1
X = ctx.empty((rows, 2 * m))
for _ in range(3):
    check(lib.xcx_rfft(ctx.handle, ptr(x), ptr(X), rows, n, 1))
ctx.sync()
z = ctx.array(np.random.default_rng(0).standard_normal((rows, 2*179999)).astype(np.float32))
Z = ctx.empty((rows, 2 * 179999))
for _ in range(3):
    check(lib.xcx_fft(ctx.handle, ptr(z), ptr(Z), rows, 179999, 1))
ctx.sync()
