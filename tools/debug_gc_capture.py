"""The hazard monoforce_amd/capture.py guards against, reproduced in isolation: a hipGraph that sits in a reference cycle is destroyed by a garbage
collection that runs while ANOTHER capture is under way.  `python tools/debug_gc_capture.py` (raw torch.cuda.graph: expected to abort) and
`python tools/debug_gc_capture.py guarded` (through capture(): the cycle is collected before the capture begins)."""
import gc, sys, os, weakref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoforce_amd.capture import capture

x = torch.zeros(1024, device='cuda')
s = torch.cuda.Stream()


def make_cycle():
    class Holder:
        pass
    h = Holder(); h.me = h
    h.g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(h.g, stream=s, capture_error_mode='thread_local'):
        h.y = x * 2
    h.g.replay()
    torch.cuda.synchronize()
    return weakref.ref(h)


gc.collect()
wr = make_cycle()
print('old graph alive:', wr() is not None, flush=True)
g = torch.cuda.CUDAGraph()
if len(sys.argv) > 1 and sys.argv[1] == 'guarded':
    with capture(g, stream=s, capture_error_mode='thread_local'):
        print('inside: old graph alive:', wr() is not None, 'gc enabled:', gc.isenabled(), flush=True)
        z = x + 1
else:
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        z = x + 1
        print('collecting inside the capture ...', flush=True)
        gc.collect()
        print('survived; old graph alive:', wr() is not None, flush=True)
        z = z + 1
g.replay(); torch.cuda.synchronize()
print('done', float(z[0]), flush=True)
