import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')
    # The CPU oracle's tensors are small ([B, N, 3] at a handful of rollouts): with all 128 intra-op threads of a GPU box's host every ATen
    # call is mostly thread hand-off -- the oracle calls of the `-m gpu` tier took 5 x longer than with 8 threads (round 6, same box: 91 s ->
    # 18 s for 32 multi-wave cases; the whole tier 805 s -> see DESIGN 2).  The referee's arithmetic does not depend on the thread count.
    # (the environment variable too: child processes of the tests -- the old-kernel / other-route children that must see the SAME CPU-made
    #  inputs bit for bit, e.g. a softmax whose last bit depends on how many threads split it -- inherit it)
    os.environ.setdefault('OMP_NUM_THREADS', '8')
    try:
        import torch
        torch.set_num_threads(min(int(os.environ['OMP_NUM_THREADS']), torch.get_num_threads()))
    except Exception:
        pass
    # MF_TEST_GC_STRESS=1: the cyclic collector runs every few allocations -- finalizers land in the middle of everything (stream captures, the
    # autograd thread's backward): the run that would have shown the abort monoforce_amd/capture.py fixes without waiting for the collector's luck
    if os.environ.get('MF_TEST_GC_STRESS'):
        import gc
        gc.set_threshold(20, 2, 2)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(REPO, 'tests', 'golden')
