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


def _rdzv_worker(rank, world, key, root, q):
    from xcorr_amd.rendezvous import FileRendezvous
    r = FileRendezvous(rank, world, key, root=root, timeout=60)
    r.barrier()
    uid = r.broadcast(bytes(range(128)) if rank == 0 else b'')
    slow = r.max(1.0 + rank)
    got = r.allgather(bytes([rank]) * 3)
    r.barrier()
    q.put((rank, uid, slow, got))
    r.close()


def test_file_rendezvous_three_ranks(tmp_path):
    """bench.py's start/stop rendezvous (no torch): barrier, max over ranks, broadcast of the
    128-byte RCCL id, all-gather -- three processes meeting in a directory."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, 'test', str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, uid, slow, got in res:
        assert uid == bytes(range(128)) and slow == float(world)
        assert got == [bytes([r]) * 3 for r in range(world)]


def test_file_rendezvous_ignores_a_stale_directory(tmp_path):
    """A crashed earlier launch with the same key left round files, a join / ack pair and an in_ file
    behind: the next launch must neither pass a barrier early nor read the old broadcast."""
    import multiprocessing as mp
    d = tmp_path / 'xcx_rdzv_stale'
    d.mkdir()
    for name, payload in [('r0_0', b''), ('r0_1', b''), ('b1', b'OLD' * 40), ('deadbeefdeadbeef_r0_1', b''),
                          ('deadbeefdeadbeef_b1', b'OLD' * 40), ('join_1_0123456789abcdef', b''),
                          ('ack_1_0123456789abcdef', b'deadbeefdeadbeef'), ('in_deadbeefdeadbeef_1', b'')]:
        (d / name).write_bytes(payload)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, 'stale', str(tmp_path), q)) for r in range(world)]
    procs[1].start()                   # the non-zero rank first: it must wait for THIS launch's rank 0
    import time
    time.sleep(0.5)
    procs[0].start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, uid, slow, got in res:
        assert uid == bytes(range(128)) and slow == float(world)
        assert got == [bytes([r]) * 3 for r in range(world)]
    assert not d.exists()              # rank 0 removed it after every rank said bye


def test_window_blocks_cover_the_day():
    """bench.py --shard windows: contiguous window blocks per rank, disjoint, covering all W."""
    for W_all in (47, 48, 5):
        for world in (1, 2, 4, 8):
            blocks = [((W_all * r) // world, (W_all * (r + 1)) // world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == W_all
            assert all(b[1] == c[0] for b, c in zip(blocks, blocks[1:]))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_rendezvous_directory_must_be_ours_and_private(tmp_path, monkeypatch):
    """The rendezvous directory has a predictable name: a symlink or a directory that other users can enter
    is refused (it would hand them the acks and the RCCL unique id), one of our own with a loose mode is
    tightened; an unusable XDG_RUNTIME_DIR falls back to the next root, an explicit root fails loudly."""
    import stat
    from xcorr_amd.rendezvous import FileRendezvous
    real = tmp_path / 'elsewhere'
    real.mkdir()
    link = tmp_path / 'xcx_rdzv_k1'
    link.symlink_to(real, target_is_directory=True)
    with pytest.raises(PermissionError):
        FileRendezvous._claim(str(link))
    loose = tmp_path / 'xcx_rdzv_k2'
    loose.mkdir(mode=0o755)
    os.chmod(loose, 0o755)
    FileRendezvous._claim(str(loose))
    assert stat.S_IMODE(os.lstat(loose).st_mode) == 0o700
    # implicit roots: a dead XDG_RUNTIME_DIR is skipped
    monkeypatch.delenv('XCX_RDZV_DIR', raising=False)
    (tmp_path / 'file_not_dir').write_text('x')            # makedirs below it fails with NotADirectoryError
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path / 'file_not_dir' / 'deeper'))
    r = FileRendezvous(0, 1, f'k3_{os.getpid()}')
    assert r.dir.startswith('/tmp/')
    r.close()
    # explicit root that cannot be used: error, no silent fallback
    blocker = tmp_path / 'file'
    blocker.write_text('x')
    with pytest.raises(OSError):
        FileRendezvous(0, 1, 'k4', root=str(blocker))


# ---------------------------------------------------------------------------
# bench.py --gpus N without a launcher: xcorr_amd/launch.py starts the ranks
# (reference analogue: the Dask fan-out of core/lazy.py:352-361)
# ---------------------------------------------------------------------------
_RANK_SCRIPT = '''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
assert os.environ["XCX_RDZV_KEY"].startswith("self_")
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail" and rank != 1:
    time.sleep(60)          # would wait for rank 1 in a rendezvous: the launcher must stop us
if rank == 0:
    print(json.dumps({"n_gpus": world, "key": os.environ["XCX_RDZV_KEY"]}), flush=True)
else:
    print("noise from rank", rank, flush=True)
'''


def _load_launch():
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('_xcx_launch_t', os.path.join(here, 'xcorr_amd', 'launch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_starts_ranks_and_relays_rank0_only(tmp_path):
    import io
    import json
    launch = _load_launch()
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT)
    out = io.StringIO()
    with open(tmp_path / 'err.txt', 'w') as err:
        rc = launch.spawn_ranks(3, [str(script), 'ok'], out=out, err=err)
    assert rc == 0
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1                                   # the bench's ONE JSON line stays alone on stdout
    assert json.loads(lines[0])['n_gpus'] == 3
    assert 'noise from rank' in (tmp_path / 'err.txt').read_text()
    # two launches never share a rendezvous key
    out2 = io.StringIO()
    with open(tmp_path / 'err2.txt', 'w') as err:
        assert launch.spawn_ranks(2, [str(script), 'ok'], out=out2, err=err) == 0
    assert json.loads(out2.getvalue())['key'] != json.loads(lines[0])['key']


def test_launcher_fails_when_a_rank_fails_and_stops_the_rest(tmp_path):
    import io
    import time
    launch = _load_launch()
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT)
    t0 = time.monotonic()
    with open(tmp_path / 'err.txt', 'w') as err:
        rc = launch.spawn_ranks(3, [str(script), 'fail'], out=io.StringIO(), err=err, grace=5.0)
    assert rc == 7
    assert time.monotonic() - t0 < 30                        # the sleeping ranks were terminated, not waited for
    assert 'rank 1 exited with status 7' in (tmp_path / 'err.txt').read_text()


def test_gpus_flag_never_runs_a_silent_single_rank():
    launch = _load_launch()
    assert launch.needs_launch(1, {}) == 'run'
    assert launch.needs_launch(8, {}) == 'spawn'             # no launcher: start the ranks ourselves
    assert launch.needs_launch(8, {'WORLD_SIZE': '8'}) == 'run'
    with pytest.raises(SystemExit):
        launch.needs_launch(8, {'WORLD_SIZE': '1'})          # never n_gpus: 1 for --gpus 8
    with pytest.raises(SystemExit):
        launch.needs_launch(1, {'WORLD_SIZE': '2'})


def test_bench_refuses_more_ranks_than_devices_without_oversubscribe():
    """On a box without a GPU (this container) the device probe fails or reports 0: `bench.py --gpus 2` must
    refuse loudly before starting any rank; on a GPU box with >= 2 devices this test does not apply."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(here, 'bench.py'), '--gpus', '64', '--steps', '1'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert '--gpus 64' in r.stderr and 'n_gpus' not in r.stdout
