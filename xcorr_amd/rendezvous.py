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
import stat
import struct
import time

__all__ = ['FileRendezvous', 'from_env']


class FileRendezvous:
    """Rounds of small files in one directory.

    The directory name is predictable (launcher pid + port) and may survive a crashed run or be
    reused by the next launch from the same shell, so nothing in it is trusted by name alone:
    rank 0 draws a fresh session nonce S and hands it to the others through a join / ack / in
    handshake, and every later file carries S in its name.

      rank r > 0 : publish  join_<r>_<own nonce>   (re-published while waiting: rank 0 may wipe stale files)
                   wait for ack_<r>_<own nonce>    (content = S)
                   publish  in_<S>_<r>
      rank 0     : wipe the directory, then acknowledge every join file it sees (stale ones too:
                   nobody reads those acks) until in_<S>_<r> exists for every r -- S is fresh, so
                   an in_ file can only come from a rank of THIS launch.

    close(): every rank's last action is bye_<S>_<r>; rank 0 waits for all of them (bounded) and
    then removes the directory, so no rank can still be polling a file that disappears.
    """

    def __init__(self, rank, world, key, root=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        # a per-user runtime directory when the session has one; /tmp is the last resort.  An explicit root
        # (argument / XCX_RDZV_DIR) is used or fails loudly; the implicit candidates are tried in order
        # (XDG_RUNTIME_DIR is often set but absent inside containers).
        explicit = root or os.environ.get('XCX_RDZV_DIR')
        roots = [explicit] if explicit else [r for r in (os.environ.get('XDG_RUNTIME_DIR'), '/tmp') if r]
        last = None
        for r in roots:
            try:
                self.dir = self._claim(os.path.join(r, f'xcx_rdzv_{key}'))
                break
            except OSError as e:           # PermissionError included
                last = e
        else:
            raise last
        self.seq = 0
        self.session = self._handshake()

    @staticmethod
    def _claim(path):
        """Create (or adopt) the rendezvous directory.  The name is predictable, so only a real directory of our
        own that nobody else can enter is accepted: a directory or symlink planted by another user would hand
        them the acks and the RCCL unique id."""
        os.makedirs(path, mode=0o700, exist_ok=True)
        st = os.lstat(path)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid():
            raise PermissionError(f'rendezvous: {path} is not a directory owned by uid {os.geteuid()}')
        if st.st_mode & 0o077:
            os.chmod(path, 0o700)                            # ours, created under an older umask: tighten or fail
            if os.lstat(path).st_mode & 0o077:
                raise PermissionError(f'rendezvous: {path} is accessible to other users')
        return path

    def _path(self, name):
        return os.path.join(self.dir, name)

    def _handshake(self):
        t0 = time.monotonic()

        def expired():
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f'rendezvous: rank {self.rank} could not join {self.dir} within {self.timeout:.0f} s')

        if self.rank == 0:
            for f in os.listdir(self.dir):                   # leftovers of an earlier run
                try:
                    os.unlink(self._path(f))
                except OSError:
                    pass
            S = os.urandom(8).hex()
            acked = set()
            while True:
                for f in os.listdir(self.dir):
                    if f.startswith('join_') and f not in acked:
                        self._publish('ack_' + f[5:], S.encode())
                        acked.add(f)
                if all(os.path.exists(self._path(f'in_{S}_{r}')) for r in range(1, self.world)):
                    return S
                expired()
                time.sleep(1e-3)
        mine = f'{self.rank}_{os.urandom(8).hex()}'
        while True:
            if not os.path.exists(self._path('join_' + mine)):
                try:
                    self._publish('join_' + mine)
                except OSError:                              # rank 0's wipe took the temporary file: try again
                    pass
            try:
                with open(self._path('ack_' + mine), 'rb') as fh:
                    S = fh.read().decode()
                if S:
                    break
            except OSError:
                pass
            expired()
            time.sleep(1e-3)
        self._publish(f'in_{S}_{self.rank}')
        return S

    # every collective is one "round": each rank publishes one file, then reads what it needs
    def _publish(self, name, payload=b''):
        tmp = self._path(f'.{name}.{self.rank}.tmp')
        with open(tmp, 'wb') as fh:
            fh.write(payload)
        os.replace(tmp, self._path(name))                    # atomic: readers never see a partial file

    def _wait(self, name, timeout=None):
        path = self._path(name)
        timeout = self.timeout if timeout is None else timeout
        t0 = time.monotonic()
        nap = 2e-4
        while not os.path.exists(path):
            if time.monotonic() - t0 > timeout:
                raise TimeoutError(f'rendezvous: rank {self.rank} waited {timeout:.0f} s for {name} in {self.dir}')
            time.sleep(nap)
            nap = min(nap * 1.5, 5e-3)
        with open(path, 'rb') as fh:
            return fh.read()

    def allgather(self, payload=b''):
        k = self.seq
        self.seq += 1
        self._publish(f'{self.session}_r{k}_{self.rank}', payload)
        return [self._wait(f'{self.session}_r{k}_{r}') for r in range(self.world)]

    def barrier(self):
        self.allgather()

    def max(self, value):
        return max(struct.unpack('<d', b)[0] for b in self.allgather(struct.pack('<d', float(value))))

    def broadcast(self, payload, src=0):
        k = self.seq
        self.seq += 1
        if self.rank == src:
            self._publish(f'{self.session}_b{k}', payload)
            return payload
        return self._wait(f'{self.session}_b{k}')

    def close(self):
        """Leave.  A rank's last action is its bye file; rank 0 removes the directory once it has every rank's."""
        try:
            self._publish(f'bye_{self.session}_{self.rank}')
            if self.rank == 0:
                for r in range(1, self.world):
                    self._wait(f'bye_{self.session}_{r}', timeout=min(self.timeout, 60.0))
                for f in os.listdir(self.dir):
                    try:
                        os.unlink(self._path(f))
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
    # unique per launch: the launcher's own ids where it provides them (torchrun: run id + port; Slurm / MPI:
    # job and step ids), always the launcher's pid and its start time -- two launches from one shell differ in
    # the latter even when port and run id are defaults
    ppid = os.getppid()
    try:
        with open(f'/proc/{ppid}/stat', 'rb') as fh:
            born = fh.read().rsplit(b')', 1)[1].split()[19].decode()     # field 22: start time in clock ticks
    except (OSError, IndexError):
        born = 'x'
    ids = [os.environ.get(v, '') for v in ('MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID', 'SLURM_STEP_ID',
                                           'OMPI_MCA_ess_base_jobid', 'PMI_JOBID')]
    key = os.environ.get('XCX_RDZV_KEY') or '_'.join([v for v in ids if v] + [str(ppid), born])
    key = ''.join(ch if ch.isalnum() or ch in '-_.' else '-' for ch in key)[:120]    # one path component
    return FileRendezvous(rank, world, key)
