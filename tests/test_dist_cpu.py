"""CPU, world_size 2, gloo: the N>1 path.  Pair-windows shard with no
data-path collective; the only exchange is the time-stack sum.  Each rank
computes its shard with the oracle (the GPU engine is exercised by -m gpu
tests); what is tested here is that the sharding + reduce reassemble the
single-process result."""
import os

import numpy as np
import pytest


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import ref_numpy as R
    from xcorr_amd import shard
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    S, n, hop, W = 5, 300, 150, 4
    T = hop * (W - 1) + n
    pairs = shard.all_pairs(S)
    # (a) pair-sharded, one day: ranks own disjoint pairs, gather = full result
    tr = R.synthetic_traces(S, T, day=0)
    mine, idx = shard.shard_pairs(pairs, rank, world)
    part = R.cc_pairs(tr, mine, n, hop, dtype=np.float32)
    full = torch.zeros((len(pairs), W, 2 * n - 1))
    full[torch.from_numpy(idx)] = torch.from_numpy(part)
    dist.all_reduce(full)                      # disjoint rows: sum == gather
    # (b) day-sharded time-stack: every rank stacks its days, one all-reduce
    days = shard.shard_days(3, rank, world)
    st = np.zeros((len(pairs), 2 * n - 1), np.float64)
    for d in days:
        st += R.cc_pairs(R.synthetic_traces(S, T, day=d), pairs, n, hop, dtype=np.float64).sum(axis=1)
    stt = torch.from_numpy(st)
    dist.all_reduce(stt)
    if rank == 0:
        q.put((full.numpy(), stt.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import ref_numpy as R
    from xcorr_amd import shard
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, stack = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    S, n, hop, W = 5, 300, 150, 4
    T = hop * (W - 1) + n
    pairs = shard.all_pairs(S)
    ref = R.cc_pairs(R.synthetic_traces(S, T, day=0), pairs, n, hop, dtype=np.float32)
    np.testing.assert_array_equal(full, ref)
    ref_stack = sum(R.cc_pairs(R.synthetic_traces(S, T, day=d), pairs, n, hop, dtype=np.float64).sum(axis=1)
                    for d in range(3))
    np.testing.assert_allclose(stack, ref_stack, atol=1e-12)
