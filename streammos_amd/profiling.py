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


def family_of(label, family=None):
    """Kernel family of a launch label: what ops.py names (the convolutions share the label prefix ``conv_cl[`` but run on
    four kernels: conv_wino, conv_wino1d, conv_igemm, conv_rows), else the label's name in front of the shape."""
    return family if family is not None else label.split("[", 1)[0]


class KernelTimer:
    def __init__(self, only=None):
        self.only = only            # filter: None = every launch, a label, or a callable (label, family) -> bool
        self.events = {}            # label -> list of (start, stop)
        self.sequence = []          # labels in launch order
        self.family = {}            # label -> kernel family

    @contextlib.contextmanager
    def span(self, label, family=None):
        fam = family_of(label, family)
        if self.only is not None and not (self.only(label, fam) if callable(self.only) else label == self.only):
            yield
            return
        self.family[label] = fam
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


_NULL = contextlib.nullcontext()


def enabled():
    """True while a KernelTimer is active or a replay is being recorded: the hot ops build their launch labels only then (a label is
    a string format of ~10 numbers: ~1.5 us per launch on the host for nothing otherwise)."""
    return _active is not None or _replay_label is not None


def span(label, family=None):
    if _active is None or torch.cuda.is_current_stream_capturing():
        return _NULL
    return _active.span(label, family)


def span_f(fmt, args, family=None):
    """span(fmt % args, family) with the formatting left out while nothing is being timed (the per-step launches)."""
    if _active is None:
        return _NULL
    return span(fmt % args, family)


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
