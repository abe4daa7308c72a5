"""GPU, >= 2 devices: the only exchange step of the path -- xcx_comm_init + xcx_stack_reduce
(RCCL all-reduce / reduce-scatter over xGMI) with one PROCESS per GPU, launched before any GPU
call in the parent, ids exchanged through the file rendezvous (no torch).  Skipped on a
one-GPU box (two ranks cannot share a device under RCCL: `ncclCommInitRank ... invalid usage`)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK = r'''
import os, sys, numpy as np
sys.path.insert(0, {root!r})
import xcorr_amd as X
from xcorr_amd.rendezvous import FileRendezvous
rank, world, key, rdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
rd = FileRendezvous(rank, world, key, root=rdir, timeout=120)
ctx = X.Context(rank)
uid = rd.broadcast(X.Context.comm_unique_id() if rank == 0 else b'')
ctx.comm_init(uid, world, rank)
assert ctx.comm_info() == (world, rank), ctx.comm_info()
S, n, hop, W = 6, 2000, 1000, 3
from xcorr_amd.synth import synthetic_traces
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
P = len(pairs) + (-len(pairs)) % world            # rows padded so that the count divides
plan = X.CCPlan(ctx, n, stack=True)
L = plan.L
out = ctx.empty((P, L)).zero()
days = [d for d in range(5) if d % world == rank]
for d in days:
    plan.stack_add(synthetic_traces(S, hop * (W - 1) + n, day=d), pairs, W, hop)
plan.stack_finish(pairs, out=out)
part = out.to_host().copy()
ctx.stack_reduce(out)                             # all-reduce: every rank holds the total
ctx.sync()
total = out.to_host().copy()
out.copy_from(part)
ctx.stack_reduce(out, reduce_scatter=True)        # each rank keeps its share of the rows
ctx.sync()
share = out.to_host().reshape(world, -1)[rank].copy()
np.save(os.path.join(rdir, f'total_{{rank}}.npy'), total)
np.save(os.path.join(rdir, f'share_{{rank}}.npy'), share)
# the same sum in the frequency domain: one reduce-scatter of the accumulators, each rank inverts its pairs
for d in days:
    plan.stack_add(synthetic_traces(S, hop * (W - 1) + n, day=d), pairs, W, hop)
fout = ctx.empty((len(pairs), L)).zero()
owned, _ = plan.stack_finish_reduce(pairs, fout)
ctx.sync()
np.save(os.path.join(rdir, f'freq_{{rank}}.npy'), fout.to_host())
np.save(os.path.join(rdir, f'owned_{{rank}}.npy'), owned)
rd.barrier()
'''


def test_two_rank_rccl_stack_reduce(tmp_path):
    import ctypes as C
    import subprocess
    from xcorr_amd import _ffi
    n = C.c_int(0)
    _ffi.check(_ffi.lib().xcx_device_count(C.byref(n)))      # counts devices, touches no GPU state
    if n.value < 2:
        pytest.skip(f'needs >= 2 GPUs for one RCCL rank per device (found {n.value})')
    world = 2
    script = tmp_path / 'rank.py'
    script.write_text(_RANK.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), 'rccl', str(tmp_path)], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    from oracle import ref_numpy as R
    from xcorr_amd.synth import synthetic_traces
    S, nn, hop, W = 6, 2000, 1000, 3
    pairs = R.all_pairs(S)
    ref = sum(R.cc_pairs(synthetic_traces(S, hop * (W - 1) + nn, day=d), pairs, nn, hop, dtype=np.float64).sum(axis=1)
              for d in range(5))
    tot = [np.load(tmp_path / f'total_{r}.npy') for r in range(world)]
    for t in tot:
        assert np.abs(t[:len(pairs)] - ref).max() / np.abs(ref).max() < 1e-5
    assert np.array_equal(tot[0], tot[1])
    shares = np.concatenate([np.load(tmp_path / f'share_{r}.npy') for r in range(world)])
    assert np.array_equal(shares.reshape(tot[0].shape), tot[0])
    # xcx_cc1d_stack_finish_reduce: every pair owned by exactly one rank, its row = the total
    owned = [np.load(tmp_path / f'owned_{r}.npy') for r in range(world)]
    assert (sum(o.astype(int) for o in owned) == 1).all()
    for r in range(world):
        f = np.load(tmp_path / f'freq_{r}.npy')
        assert np.abs(f[owned[r]] - ref[owned[r]]).max() / np.abs(ref).max() < 1e-5


_RANK_WINDOWS = r'''
import os, sys, numpy as np
sys.path.insert(0, {root!r})
import ctypes as C
import xcorr_amd as X
from xcorr_amd import _ffi
from xcorr_amd.rendezvous import FileRendezvous
from xcorr_amd.synth import synthetic_traces
rank, world, key, rdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
rd = FileRendezvous(rank, world, key, root=rdir, timeout=120)
nd = C.c_int(0)
_ffi.check(_ffi.lib().xcx_device_count(C.byref(nd)))
ctx = X.Context(rank % max(nd.value, 1))          # ranks share the device when there are fewer GPUs
S, n, hop, W = 6, 3000, 1500, 7
tr = synthetic_traces(S, hop * (W - 1) + n)
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
w0, w1 = (W * rank) // world, (W * (rank + 1)) // world        # bench.py --shard windows
plan = X.CCPlan(ctx, n, demean=True)
blk = plan.pairs(np.ascontiguousarray(tr[:, w0 * hop:(w1 - 1) * hop + n]), pairs, w1 - w0, hop)
np.save(os.path.join(rdir, f'block_{{rank}}.npy'), np.asarray(blk))
rd.barrier()
'''


def test_window_blocks_over_rank_processes(tmp_path):
    """bench.py's N > 1 default (strong scaling, no collective): every rank PROCESS correlates its
    block of windows with the engine; the blocks concatenated equal the oracle on the whole
    record.  Runs on one GPU too (the ranks then share the device)."""
    import subprocess
    world = 3
    script = tmp_path / 'rank_w.py'
    script.write_text(_RANK_WINDOWS.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), 'win', str(tmp_path)], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    from oracle import ref_numpy as R
    from xcorr_amd.synth import synthetic_traces
    S, n, hop, W = 6, 3000, 1500, 7
    tr = synthetic_traces(S, hop * (W - 1) + n)
    ref = R.cc_pairs(tr, R.all_pairs(S), n, hop, demean_=True, dtype=np.float64)
    got = np.concatenate([np.load(tmp_path / f'block_{r}.npy') for r in range(world)], axis=1)
    assert got.shape == ref.shape == (15, W, 2 * n - 1)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


_RANK_WINDOWS_PRE = r'''
import os, sys, numpy as np
sys.path.insert(0, {root!r})
import ctypes as C
import xcorr_amd as X
from xcorr_amd import _ffi, workload
from xcorr_amd.rendezvous import FileRendezvous
from xcorr_amd.synth import synthetic_traces
rank, world, key, rdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
rd = FileRendezvous(rank, world, key, root=rdir, timeout=120)
nd = C.c_int(0)
_ffi.check(_ffi.lib().xcx_device_count(C.byref(nd)))
ctx = X.Context(rank % max(nd.value, 1))
S, n, hop, W = 5, 30000, 15000, 12
T = hop * (W - 1) + n
raw = synthetic_traces(S, T, day=3)
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
w0, w1 = (W * rank) // world, (W * (rank + 1)) // world        # bench.py --shard windows
sd = workload.StationDay(ctx, S, T, n, hop, pairs, w_first=w0, W=w1 - w0).load(raw)
sd.step()
sd.step()
ctx.sync()
blk = sd.out.rows_to_host(0, sd.nrows, sd.L).reshape(len(pairs), w1 - w0, sd.L)
np.save(os.path.join(rdir, f'pre_block_{{rank}}.npy'), blk)
np.save(os.path.join(rdir, f'pre_rows_{{rank}}.npy'), sd.work.rows_to_host(0, S, T)[:, w0 * hop:(w1 - 1) * hop + n])
open(os.path.join(rdir, f'pre_warm_{{rank}}.txt'), 'w').write(str(sd.warm))
rd.barrier()
'''


def test_window_blocks_with_the_preprocess_chain(ctx, tmp_path):
    """--shard windows with the config-3 front end: a rank demeans with the mean of the WHOLE day but
    tapers / band-passes only its own block of windows plus the filter's warm-up (xcx_sosfilt_cols), so
    the replicated part of an N-rank step is one read of the day, not the whole front end.  The blocks of
    three rank processes must reproduce the single-process station-day: the preprocessed samples to the
    rounding noise of the float64 recursion, the CCFs to 1e-5 (one-bit sign flips counted)."""
    import subprocess
    from xcorr_amd import workload
    from xcorr_amd.synth import synthetic_traces
    world = 3
    script = tmp_path / 'rank_p.py'
    script.write_text(_RANK_WINDOWS_PRE.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), 'winp', str(tmp_path)], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    S, n, hop, W = 5, 30000, 15000, 12
    T = hop * (W - 1) + n
    raw = synthetic_traces(S, T, day=3)
    pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
    sd = workload.StationDay(ctx, S, T, n, hop, pairs).load(raw)
    sd.step()
    ctx.sync()
    ref = sd.out.rows_to_host(0, sd.nrows, sd.L).reshape(len(pairs), W, sd.L)
    ref_rows = sd.work.rows_to_host(0, S, T)
    sd.close()
    warm = int((tmp_path / 'pre_warm_1.txt').read_text())
    assert 0 < warm < (W // world) * hop                     # the later blocks really start from a warm-up, not from 0
    flips = 0
    for r in range(world):
        w0, w1 = (W * r) // world, (W * (r + 1)) // world
        rows = np.load(tmp_path / f'pre_rows_{r}.npy')
        want = ref_rows[:, w0 * hop:(w1 - 1) * hop + n]
        assert np.abs(rows - want).max() <= 2e-7 * np.abs(want).max()          # isolated last-bit differences at most
        flips += int(np.count_nonzero(np.sign(rows) != np.sign(want)))
    got = np.concatenate([np.load(tmp_path / f'pre_block_{r}.npy') for r in range(world)], axis=1)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5 + 5e-4 * flips
