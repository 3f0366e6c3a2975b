"""Optional per-kernel HIP-event timing of the C-ABI launches (used by bench.py; off by default, zero cost when off)."""
import contextlib

import torch

_records = None     # None = off; else list of (name, start_event, end_event)
_pool = []          # pre-created events (creating them inside a timed region costs host time)
_every, _step = 1, 0
_launches = {}      # name -> text of mf_last_launch() for the sampled launches since the last start()


def start(capacity=4096, every=1):
    """Record HIP events around the C-ABI launches from now on.  `every` > 1 samples: only the launches of every `every`-th
    step (the caller marks steps with `next_step()`) are bracketed -- an event record is a packet of its own on the stream, and
    eight of them around the four kernels of a 0.55 ms train step cost 33 us (6 %) of it."""
    global _records, _every, _step
    while len(_pool) < 2 * capacity:
        _pool.append(torch.cuda.Event(enable_timing=True))
    _records = []
    _every, _step = max(int(every), 1), -1
    _launches.clear()


def set_every(every):
    """Change the sampling period of a running recording (the records so far are kept)."""
    global _every, _step
    _every, _step = max(int(every), 1), -1


def next_step():
    global _step
    _step += 1


def sampled():
    """True while the launches of the current step are being bracketed with events (callers that replay a captured graph run
    such a step launch by launch instead)."""
    return _records is not None and (_every <= 1 or _step % _every == 0)


def stop():
    """Stop recording and return {name: [ms, ...]} (synchronises)."""
    global _records
    recs, _records = _records or [], None
    torch.cuda.synchronize()
    out = {}
    for name, a, b in recs:
        out.setdefault(name, []).append(a.elapsed_time(b))
    return out


@contextlib.contextmanager
def timed(name, device):
    if _records is None or (_every > 1 and _step % _every != 0):
        yield
        return
    i = 2 * len(_records)
    if i + 1 < len(_pool):
        a, b = _pool[i], _pool[i + 1]
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream(device))     # the stream the kernel is launched on
    yield
    b.record(torch.cuda.current_stream(device))
    _records.append((name, a, b))


def note_launch(name):
    """Called right after a sampled C-ABI rollout launch, on the launching thread (mf_last_launch is thread-local and the backward
    runs on autograd's thread): remember which kernel template the library's dispatcher picked."""
    if _records is None or (_every > 1 and _step % _every != 0):
        return
    from . import _lib
    _launches[name] = _lib.lib().mf_last_launch().decode()


def launches():
    """{name: 'kernel<template parameters> grid=.. block=.. launches=..'} of the sampled launches since the last start()."""
    return dict(_launches)
