"""Stream capture with the Python garbage collector held off.

`torch.cuda.graph.__enter__` no longer collects garbage before a capture (torch 2.10: only under `torch.compiler.config.force_cudagraph_gc`),
and nothing keeps a collection from running INSIDE one: the cyclic collector is triggered by allocation counts, on whichever thread happens to
allocate -- the autograd thread that replays this package's backward included.  What such a collection frees is whatever cycles were lying
around: an earlier hipGraph of a discarded cache entry, events, tensors of another pool.  Destroying a graph (hipGraphExecDestroy) while a
stream is capturing aborts the process in the HIP runtime -- no exception, no message: `pytest tests -m gpu -q` died in the seventh test
(tests/test_api_cache_gpu.py, the first re-capture after a discarded entry) while `-x -q` passed, by the collector's timing alone.

`capture(graph, **kw)` = `torch.cuda.graph(graph, **kw)` with one full collection BEFORE the capture begins and the collector disabled until it
has ended.  Every capture of the package (api_cache, train, graphed) and of bench.py goes through it."""
import contextlib
import gc

import torch

__all__ = ['capture']


@contextlib.contextmanager
def capture(graph, **kw):
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield graph
    finally:
        if was_enabled:
            gc.enable()
