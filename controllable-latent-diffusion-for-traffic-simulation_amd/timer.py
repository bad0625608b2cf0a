"""Per-phase timers of the rollout loop, same call surface as the reference's `Timers`
(`src/tbsim/utils/timer.py:41-64`: tic / toc / timed / __str__, keys "obs", "to_torch", "network", "env_step", "step" in
`src/tbsim/utils/env_utils.py:268-298`).  The reference reads the host clock around asynchronous GPU work; here every
tic/toc also records a HIP event on the current stream, so `gpu_ms(key)` gives the device time of a phase without
synchronising inside the loop (events are resolved when the numbers are read)."""
from __future__ import annotations

import time
from contextlib import contextmanager

import torch


class Timer:
    def __init__(self, device=None):
        self.total_time, self.calls, self.start_time, self.diff, self.average_time, self.times = 0.0, 0, 0.0, 0.0, 0.0, []
        self._gpu = device is not None and torch.cuda.is_available()
        self._events, self._open = [], None

    def tic(self):
        self.start_time = time.time()
        if self._gpu:
            self._open = torch.cuda.Event(enable_timing=True)
            self._open.record()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.times.append(self.diff)
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        if self._gpu and self._open is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self._events.append((self._open, end))
            self._open = None
        return self.average_time if average else self.diff

    def gpu_ms(self):
        """Mean device time per call (waits for the recorded events)."""
        if not self._events:
            return 0.0
        self._events[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    @contextmanager
    def timed(self):
        self.tic()
        yield
        self.toc()


class Timers:
    def __init__(self, device=None):
        self._timers, self._device = {}, device

    def tic(self, key):
        if key not in self._timers:
            self._timers[key] = Timer(self._device)
        self._timers[key].tic()

    def toc(self, key):
        self._timers[key].toc()

    @contextmanager
    def timed(self, key):
        self.tic(key)
        yield
        self.toc(key)

    def gpu_ms(self, key):
        return self._timers[key].gpu_ms() if key in self._timers else 0.0

    def __str__(self):
        return ", ".join("%s: %f" % (k, v.average_time) for k, v in self._timers.items())
