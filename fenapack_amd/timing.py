"""DOLFIN-style named timers (``dolfin.Timer`` / ``dolfin.timed`` /
``dolfin.list_timings``), which the reference wraps around its hot path and
its set-ups: ``@timed("FENaPack: PCDPC_BRM1 apply")``
(``fenapack/preconditioners.py:98,148,219,264``) and
``with Timer("FENaPack: {} setup")`` (``fenapack/field_split.py:89,104,143``;
``field_split_backend.py:138,254,262``).  Same task names here, so a
``list_timings()`` dump of a run reads like the reference's.

What a timer around ``apply`` measures: the engine *enqueues* the fused apply
on its HIP stream and returns, so - like every host-side timer around
asynchronous device work - the wall time is the launch cost unless the caller
synchronises.  ``FENAPACK_TIMERS_SYNC=1`` makes ``timed`` call
``torch.cuda.synchronize()`` before it stops the clock (diagnostics only: it
serialises the host with the device).
"""

import functools
import os
import time

_TIMINGS = {}          # task -> [reps, wall seconds]


class Timer(object):
    """``with Timer("task"):`` or ``t = Timer("task"); ...; t.stop()``."""

    def __init__(self, task=None):
        self.task = task
        self._t0 = time.perf_counter()
        self._elapsed = None

    def start(self):
        self._t0 = time.perf_counter()
        self._elapsed = None

    def stop(self):
        if self._elapsed is None:
            self._elapsed = time.perf_counter() - self._t0
            if self.task is not None:
                rec = _TIMINGS.setdefault(self.task, [0, 0.0])
                rec[0] += 1
                rec[1] += self._elapsed
        return self._elapsed

    def elapsed(self):
        return (time.perf_counter() - self._t0 if self._elapsed is None
                else self._elapsed)

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def _device_sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed(task):
    """Decorator: accumulate wall time and call count under ``task``."""
    def wrap(fn):
        @functools.wraps(fn)
        def inner(*args, **kwargs):
            t = Timer(task)
            try:
                return fn(*args, **kwargs)
            finally:
                if os.environ.get("FENAPACK_TIMERS_SYNC") == "1":
                    _device_sync()
                t.stop()
        return inner
    return wrap


def timings(clear=False):
    """{task: (reps, total wall seconds)}."""
    out = {k: (v[0], v[1]) for k, v in _TIMINGS.items()}
    if clear:
        _TIMINGS.clear()
    return out


def list_timings(clear=False, file=None):
    """Print the table ``dolfin.list_timings(TimingClear, [TimingType.wall])``
    prints: task, repetitions, average and total wall time."""
    rows = sorted(timings(clear).items())
    width = max([len(k) for k, _ in rows] + [4])
    lines = ["%-*s  |  %8s  %12s  %12s" % (width, "task", "reps",
                                           "wall avg", "wall tot"),
             "-" * (width + 42)]
    for k, (n, t) in rows:
        lines.append("%-*s  |  %8d  %12.6f  %12.6f"
                     % (width, k, n, t / max(n, 1), t))
    text = "\n".join(lines)
    print(text, file=file)
    return text
