"""One-off soak of the BEV pooling kernels: tests/test_splat_gpu.py::_check_random_pool over many seeds.
python tools/soak_splat.py [first_seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_splat_gpu import _check_random_pool
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
bad = 0
for seed in range(first, first + count):
    try:
        _check_random_pool(seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:200], flush=True)
print(f'{count} seeds from {first}: {bad} failures')
