"""Per-kernel timing with HIP events on the stream the kernels are launched on.

``ops`` brackets every C-ABI launch with ``record(label)`` when a timer is active.  Events are
``torch.cuda.Event`` objects recorded on ``torch.cuda.current_stream`` -- the very stream whose handle is
passed to the C ABI -- so they time exactly the kernels of that launch.
"""
import contextlib

import torch

_active = None
_replay_label = None      # label whose next launch is kept (closure over its arguments) for an isolated re-run
_replay = {}


class KernelTimer:
    def __init__(self, only=None):
        self.only = only            # label filter (None = every launch)
        self.events = {}            # label -> list of (start, stop)
        self.sequence = []          # labels in launch order

    @contextlib.contextmanager
    def span(self, label):
        if self.only is not None and label != self.only:
            yield
            return
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.events.setdefault(label, []).append((a, b))
        self.sequence.append(label)

    def summary(self):
        """label -> (calls, total_ms, mean_ms); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for label, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[label] = (len(ms), float(sum(ms)), float(sum(ms) / len(ms)))
        return out


@contextlib.contextmanager
def kernel_timer(only=None):
    global _active
    prev, _active = _active, KernelTimer(only)
    try:
        yield _active
    finally:
        _active = prev


def span(label):
    if _active is None or torch.cuda.is_current_stream_capturing():
        return contextlib.nullcontext()
    return _active.span(label)


def request_replay(label):
    """Ask ops to keep the next launch carrying `label` (bench.py re-runs the dominant kernel alone on the GPU)."""
    global _replay_label
    _replay_label = label
    _replay.pop(label, None)


def offer_replay(label, fn):
    global _replay_label
    if label == _replay_label:
        _replay[label] = fn
        _replay_label = None


def replay_of(label):
    return _replay.get(label)
