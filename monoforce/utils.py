"""The handful of `monoforce.utils` helpers the hot-path callers use (`/root/reference/monoforce/src/monoforce/utils.py`:
`timing` :32-40, `read_yaml` :68-71, `write_to_yaml` :74-76, `str2bool` :78-79)."""
import functools
import time

import yaml


def read_yaml(path):
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def write_to_yaml(cfg: dict, path):
    with open(path, 'w') as f:
        yaml.dump(cfg, f, default_flow_style=False)


def str2bool(v):
    return v.lower() in ('1', 'yes', 'true', 't', 'y')


def timing(f):
    @functools.wraps(f)
    def wrapper(*args, **kwargs):
        t0 = time.time()
        out = f(*args, **kwargs)
        print(f'{f.__name__} took {time.time() - t0:.3f} s')
        return out
    return wrapper
