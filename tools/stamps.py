"""Where does a pair-pass wave spend its life?  Runs one step of the bench workload on the
DIAGNOSTIC library (make -C xcorr_amd/csrc STAMPS=1; in-kernel s_memtime stamps at the phase
boundaries of k_wpair_dual, product build unaffected) and prints per-phase medians.

    XCX_LIB_PATH=xcorr_amd/lib/libxcx_stamps.so python tools/stamps.py [--launch 400] [--stations 64]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('XCX_LIB_PATH', os.path.join(ROOT, 'xcorr_amd', 'lib', 'libxcx_stamps.so'))

ap = argparse.ArgumentParser()
ap.add_argument('--launch', type=int, default=400)
ap.add_argument('--stations', type=int, default=64)
ap.add_argument('--seconds', type=float, default=86400.)
a = ap.parse_args()
os.environ['XCX_STAMP_LAUNCH'] = str(a.launch)

import xcorr_amd as X                      # noqa: E402
from xcorr_amd import _ffi, workload       # noqa: E402
from xcorr_amd.synth import synthetic_traces  # noqa: E402

ctx = X.Context(0)
S, T, n, hop = a.stations, int(a.seconds * 50), 180000, 90000
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
sd = workload.StationDay(ctx, S, T, n, hop, pairs).load(synthetic_traces(S, T))
sd.step()
ctx.sync()
lib = _ffi.lib()
fn = C.CDLL(_ffi.LIB_PATH).xcx_debug_stamps
fn.restype = C.c_long
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
NS = 12
maxw = 8 * 4 * 64 * 64
buf = np.zeros((maxw, NS), np.uint64)
got = fn(sd.plan.handle, buf.ctypes.data_as(C.c_void_p), maxw)
buf = buf[:max(got, 0)]
live = buf[buf[:, 9] != 0]
print(f'launch {a.launch}: {len(live)} waves recorded of {got} slots (cycle counts; us at an assumed {2.1} GHz)')
if not len(live):
    sys.exit(1)
t = live.astype(np.int64)
kind = (live[:, 10] >> np.uint64(32)).astype(np.int64)
# s_memtime counts shader cycles but every XCD has its own counter base: only differences inside
# one wave are meaningful (cycles; ~2.1-2.2 GHz under this load)
ghz = 2.1
names = {1: ['launch -> inputs arrived (desc + twiddle + 48 loads, product)', 'stage 0 (3 x radix 8)', 'LDS exchange 1',
             'stage 1 (twiddle chain + 3 x radix 8)', 'LDS exchange 2', 'stage 2 (twiddle chain + radix 24)',
             'four-step twiddle + T stores issued', 'T stores acknowledged', 'tail'],
         2: ['launch -> stage 0 done (desc + 34 T loads in two batches, untangle, radix-8 pairs)', 'LDS exchange (2 rounds)',
             'stage 1 (twiddle chain + 2 x radix 15)', 'staging + cooperative CCF stores issued (4 barriers)',
             'CCF stores acknowledged', 'tail']}
for k, label in ((1, 'COLUMN waves'), (2, 'ROW waves')):
    w = t[kind == k]
    if not len(w):
        continue
    print(f'\n{label}: {len(w)}   lifetime median {np.median(w[:, 9] - w[:, 0]) / (ghz * 1e3):.2f} us '
          f'(p10 {np.percentile(w[:, 9] - w[:, 0], 10) / (ghz * 1e3):.2f}, p90 {np.percentile(w[:, 9] - w[:, 0], 90) / (ghz * 1e3):.2f})')
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9] if k == 1 else [0, 2, 3, 4, 7, 8, 9]      # the row body has no stamp 1
    labs = names[k]
    for (i0, i1), nm in zip(zip(idx[:-1], idx[1:]), labs):
        d = (w[:, i1] - w[:, i0]).astype(np.float64)
        print(f'  {nm:62s} median {np.median(d):8.0f} cyc = {np.median(d) / (ghz * 1e3):6.2f} us   '
              f'(p10 {np.percentile(d, 10):7.0f}, p90 {np.percentile(d, 90):7.0f})')
# slot occupancy of this launch: stamp 11 = s_memrealtime (100 MHz) at the wave's end, its life from the cycle stamps
end_us = t[:, 11] / 100.0
life_us = (t[:, 9] - t[:, 0]) / (ghz * 1e3)
start_us = end_us - life_us
span = end_us.max() - start_us.min()
print(f'\nlaunch span {span:.1f} us, sum of wave lifetimes {life_us.sum() / 1e3:.2f} ms -> {life_us.sum() / span:.0f} wave slots held on '
      f'average (of 4096 on the chip; the other pipeline\'s launch holds the rest of what is busy)')
grid = np.linspace(start_us.min(), end_us.max(), 21)
print('waves alive at 5 % steps of the span:', ' '.join(str(int(((start_us <= x) & (end_us > x)).sum())) for x in grid))
np.save(os.path.join(ROOT, 'gpurun_out', 'stamps_last.npy'), live)
