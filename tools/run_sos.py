"""One forward sosfilt over 64 day-long rows (profiling target for tools/pmc_any.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from xcorr_amd import ops
from xcorr_amd.engine import default_context
ctx = default_context()
work = ctx.array(np.random.default_rng(0).standard_normal((64, 4320000)).astype(np.float32))
sos = scipy.signal.butter(4, [0.1, 10.0], btype='bandpass', output='sos', fs=50.)
for _ in range(2):
    ops.sosfilt(ctx, work, sos)
ctx.sync()
