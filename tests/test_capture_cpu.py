"""monoforce_amd/capture.py on the host alone (the capture itself replaced by a stand-in): one full collection BEFORE the capture begins, the
cyclic collector off inside, its state put back as found -- also when the body raises.  The hazard proper (a hipGraph destroyed by a collection
inside another capture aborts the process) needs a GPU: tests/test_api_cache_gpu.py::test_capture_holds_the_garbage_collector_off,
tools/debug_gc_capture.py."""
import contextlib
import gc
import weakref

import pytest
import torch

from monoforce_amd import capture as cap


def test_collector_is_off_inside_and_restored(monkeypatch):
    seen = {}

    @contextlib.contextmanager
    def fake_graph(graph, **kw):
        seen['begin'] = (gc.isenabled(), wr() is None, kw)
        yield
        seen['end'] = gc.isenabled()
    monkeypatch.setattr(torch.cuda, 'graph', fake_graph)

    class Holder:
        pass
    h = Holder(); h.me = h
    wr = weakref.ref(h)
    del h
    assert gc.isenabled() and wr() is not None
    with cap.capture('g', stream='s', capture_error_mode='thread_local') as g:
        assert g == 'g' and not gc.isenabled()
    assert seen['begin'] == (False, True, dict(stream='s', capture_error_mode='thread_local')) and seen['end'] is False      # collected BEFORE, off throughout
    assert gc.isenabled()
    with pytest.raises(KeyError):
        with cap.capture('g'):
            raise KeyError('body')
    assert gc.isenabled()
    gc.disable()
    try:
        with cap.capture('g'):
            pass
        assert not gc.isenabled()      # a caller who had it off keeps it off
    finally:
        gc.enable()
