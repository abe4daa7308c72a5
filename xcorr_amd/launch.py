"""
Start the one-process-per-GPU ranks of ONE node from a plain `python bench.py --gpus N`.

The reference fans its pair x day jobs out over Dask workers (`core/lazy.py:352-361`,
`LocalCluster(threads_per_worker=1)`); here a rank is a process that owns one GPU.  The
parent that calls `spawn_ranks` must NOT have touched the GPU (no HIP call, no
`xcorr_amd` context): it only starts N fresh children with RANK / LOCAL_RANK /
WORLD_SIZE and a launch-unique rendezvous key in their environment, relays rank 0's
stdout, and reports failure if any child fails.  A failed child is never retried and no
process is ever replaced in place (`os.exec*`).
"""
import os
import subprocess
import sys
import time

__all__ = ['spawn_ranks', 'visible_devices', 'launch_key', 'needs_launch']


def needs_launch(gpus, environ=None):
    """What a script called with `--gpus N` has to do, from its environment alone:
    'run'   : this process is a rank (a launcher set WORLD_SIZE == N) or N == 1 without a launcher,
    'spawn' : N > 1 and no launcher set WORLD_SIZE -> start the ranks ourselves.
    A launcher whose WORLD_SIZE differs from --gpus is an error (never a silent one-rank run)."""
    env = os.environ if environ is None else environ
    ws = env.get('WORLD_SIZE')
    if ws is None or ws == '':
        return 'spawn' if gpus > 1 else 'run'
    if int(ws) != gpus:
        raise SystemExit(f'--gpus {gpus} but the launcher set WORLD_SIZE={ws}: refusing to run a job of another size')
    return 'run'


def launch_key():
    """Rendezvous key unique to this launch: pid, start time and 8 random bytes."""
    return f'self_{os.getpid()}_{time.time_ns():x}_{os.urandom(8).hex()}'


def visible_devices(python=None, timeout=300.0):
    """Number of GPUs `libxcx.so` sees, asked from a CHILD process so that the caller stays free of any
    GPU state.  Raises RuntimeError when the probe cannot run (no library, no driver)."""
    code = ('import ctypes, sys\n'
            'from xcorr_amd import _ffi\n'
            'n = ctypes.c_int(0)\n'
            '_ffi.check(_ffi.lib().xcx_device_count(ctypes.byref(n)))\n'
            'sys.stdout.write(str(n.value))\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run([python or sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError('device probe failed: ' + (r.stderr.strip().splitlines() or ['?'])[-1])
    return int(r.stdout.strip() or 0)


def spawn_ranks(n, argv, env=None, python=None, poll=0.05, grace=10.0, out=None, err=None):
    """Run `python argv...` as ranks 0..n-1 and wait for all of them.

    Returns the exit status for the caller to exit with: 0 when every rank returned 0, otherwise the
    first non-zero status seen (a signal -> 128 + signo).  As soon as one rank fails the others are
    terminated (SIGTERM, SIGKILL after `grace` seconds) -- they would wait for it in the rendezvous.
    Rank 0's stdout is relayed to `out` (default: this process's stdout) line by line; the other
    ranks' stdout goes to `err` so that the one JSON line of the bench stays alone on stdout."""
    if n < 1:
        raise ValueError('n must be >= 1')
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    base.setdefault('MASTER_PORT', '29500')
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    base['WORLD_SIZE'] = str(n)
    base['XCX_RDZV_KEY'] = base.get('XCX_RDZV_KEY') or launch_key()
    base['XCX_SELF_LAUNCHED'] = '1'
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    procs = []
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=e,
                                          stdout=subprocess.PIPE if r == 0 else err, stderr=err,
                                          text=True if r == 0 else None))
        import threading

        def relay():
            for line in procs[0].stdout:
                out.write(line)
                out.flush()

        t = threading.Thread(target=relay, daemon=True)
        t.start()
        status = 0
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and status == 0:
                    status = 128 - rc if rc < 0 else rc
                    err.write(f'[launch] rank {r} exited with status {rc}; stopping the other ranks\n')
                    err.flush()
            if status and alive:
                _stop([procs[r] for r in alive], grace)
                alive.clear()
            if alive:
                time.sleep(poll)
        t.join(timeout=5.0)
        return status
    except BaseException:
        _stop([p for p in procs if p.poll() is None], grace)
        raise


def _stop(procs, grace):
    """Terminate exactly the processes we started (never by name or pattern)."""
    for p in procs:
        try:
            p.terminate()
        except OSError:
            pass
    t0 = time.monotonic()
    for p in procs:
        try:
            p.wait(timeout=max(0.1, grace - (time.monotonic() - t0)))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
                p.wait(timeout=5.0)
            except (OSError, subprocess.TimeoutExpired):
                pass
