"""One-off soak of the component-parallel kernels: tests/test_random_shapes_gpu.py::_cp_vs_lanes over many seeds.
python tools/soak_cp.py [first_seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_random_shapes_gpu import _cp_vs_lanes, _cp_case
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad, worst = 0, 0.0
for seed in range(first, first + count):
    try:
        worst = max(worst, _cp_vs_lanes(seed))
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
print(f'{count} seeds from {first}: {bad} failures, worst relative difference {worst:.2e}')
