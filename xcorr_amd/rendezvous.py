"""
Start/stop rendezvous of the one-process-per-GPU ranks of ONE node without any
framework: a directory of small files under /tmp (or $XCX_RDZV_DIR).

The data path needs no collective (pair-windows are independent);
the ranks only have to (a) start and stop the clock together, (b) agree on the
slowest rank's time, (c) pass the 128-byte RCCL unique id from rank 0 to the
others for the time-stack reduce.  A launcher (torchrun, mpirun, a shell loop)
provides RANK / WORLD_SIZE / MASTER_PORT; the launcher's pid (the common parent
of all ranks) and the port make the directory unique per launch.
"""
import os
import struct
import time

__all__ = ['FileRendezvous', 'from_env']


class FileRendezvous:
    def __init__(self, rank, world, key, root=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        root = root or os.environ.get('XCX_RDZV_DIR') or '/tmp'
        self.dir = os.path.join(root, f'xcx_rdzv_{key}')
        os.makedirs(self.dir, exist_ok=True)
        self.seq = 0

    # every collective is one "round": each rank publishes one file, then reads what it needs
    def _publish(self, name, payload=b''):
        tmp = os.path.join(self.dir, f'.{name}.{self.rank}.tmp')
        with open(tmp, 'wb') as fh:
            fh.write(payload)
        os.replace(tmp, os.path.join(self.dir, name))        # atomic: readers never see a partial file

    def _wait(self, name):
        path = os.path.join(self.dir, name)
        t0 = time.monotonic()
        nap = 2e-4
        while not os.path.exists(path):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f'rendezvous: rank {self.rank} waited {self.timeout:.0f} s for {name} in {self.dir}')
            time.sleep(nap)
            nap = min(nap * 1.5, 5e-3)
        with open(path, 'rb') as fh:
            return fh.read()

    def allgather(self, payload=b''):
        k = self.seq
        self.seq += 1
        self._publish(f'r{k}_{self.rank}', payload)
        return [self._wait(f'r{k}_{r}') for r in range(self.world)]

    def barrier(self):
        self.allgather()

    def max(self, value):
        return max(struct.unpack('<d', b)[0] for b in self.allgather(struct.pack('<d', float(value))))

    def broadcast(self, payload, src=0):
        k = self.seq
        self.seq += 1
        if self.rank == src:
            self._publish(f'b{k}', payload)
            return payload
        return self._wait(f'b{k}')

    def close(self):
        """Leave; the last rank out removes the directory."""
        try:
            self.allgather()
            if self.rank == 0:
                time.sleep(0.05)
                for f in os.listdir(self.dir):
                    try:
                        os.unlink(os.path.join(self.dir, f))
                    except OSError:
                        pass
                os.rmdir(self.dir)
        except (OSError, TimeoutError):
            pass


class _Single:
    rank, world = 0, 1

    def barrier(self):
        pass

    def max(self, value):
        return float(value)

    def broadcast(self, payload, src=0):
        return payload

    def allgather(self, payload=b''):
        return [payload]

    def close(self):
        pass


def from_env():
    """Rendezvous for the process's RANK / WORLD_SIZE (a no-op object for one rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return _Single()
    rank = int(os.environ['RANK'])
    key = os.environ.get('XCX_RDZV_KEY') or \
        f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return FileRendezvous(rank, world, key)
