"""In-memory waveform client with the one method ``core.process`` calls on the
reference's ``xcorr.stream.Client`` (stream/client.py:1116-1149).  File and
network access (SDS/FDSN) is out of scope for this build; this client serves
windows of already preprocessed station arrays and is what tests and
benchmarks use."""
import numpy as np

__all__ = ['ArrayClient']


class ArrayClient:
    """traces: dict receiver -> float32[T]; `t0`: time of sample 0 (POSIX seconds, or
    anything ``init`` accepts as a start time: ISO string, datetime64, Timestamp)."""

    def __init__(self, traces, sampling_rate, t0=0.0, separator='-'):
        from .init import _seconds
        self.traces = {k: np.ascontiguousarray(v, dtype=np.float32)
                       for k, v in traces.items()}
        self.sampling_rate = float(sampling_rate)
        self.t0 = _seconds(t0)[0]
        self.separator = separator

    def window(self, receiver, center, duration):
        """Samples of [center - duration/2, center + duration/2) or None if
        the record does not cover it (-> status -1, core/process.py:122-132)."""
        x = self.traces.get(receiver)
        if x is None:
            return None
        n = int(duration * self.sampling_rate)
        i0 = int(round((center - duration / 2 - self.t0) * self.sampling_rate))
        if i0 < 0 or i0 + n > len(x):
            return None
        return x[i0:i0 + n], self.t0 + i0 / self.sampling_rate

    def get_pair_processed_waveforms(self, pair, time, duration, **kwargs):
        """Returns [(data_a, starttime_a), (data_b, starttime_b)] or None."""
        a, b = str(pair).split(self.separator)
        wa = self.window(a, time, duration)
        wb = self.window(b, time, duration)
        if wa is None or wb is None:
            return None
        return [wa, wb]
