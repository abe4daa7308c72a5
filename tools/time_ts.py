import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd import ops
from xcorr_amd.engine import default_context
ctx = default_context()
rows, L = 512, 359999
x = ctx.array((np.random.default_rng(0).standard_normal((rows, L)) * 0.01).astype(np.float32))
y = ctx.empty((rows, L - 1))
d = np.random.default_rng(1).uniform(-0.02, 0.02, rows)
for _ in range(3):
    ops.timeshift(ctx, x, d, 50., out=y)
ctx.sync()
