"""Per-kernel HIP-event timing on the launch stream (bench.py's roofline leg).

Op wrappers call `region(name, algorithmic_bytes)` around a launch; when the clock is disabled
(the default) that is a no-op.  Events are recorded on torch's current stream, which is the
stream every launcher in this package is given (ao_amd/_lib.py:stream_ptr)."""
import contextlib

import torch


class KernelClock:
    def __init__(self):
        self.enabled = False
        self.records = {}  # name -> list of (start_event, stop_event, bytes)

    def reset(self):
        self.records = {}

    @contextlib.contextmanager
    def region(self, name, algorithmic_bytes=0):
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.records.setdefault(name, []).append((a, b, algorithmic_bytes))

    def summary(self):
        """{name: dict(launches, total_ms, avg_us, bytes_per_launch)} -- call after a device sync."""
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = dict(launches=len(recs), total_ms=sum(ms), avg_us=1e3 * sum(ms) / len(recs),
                             bytes_per_launch=sum(r[2] for r in recs) / len(recs))
        return out


clock = KernelClock()
