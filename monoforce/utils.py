"""The `monoforce.utils` helpers the hot-path callers import (`/root/reference/monoforce/src/monoforce/utils.py`:
`timing` :32-40, `read_yaml` :68-71, `write_to_yaml` :74-76, `str2bool` :78-79, `load_calib` :98-121).  `compile_data`
(:124-188) builds the ROUGH datasets, which are out of scope (no data): the name imports, calling it raises."""
import functools
import os
import time

import numpy as np
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


def load_calib(calib_path):
    """Camera calibration of a ROUGH sequence: `<calib_path>/cameras/*.yaml` (one dict per camera, keyed by file stem) plus
    `<calib_path>/transformations.yaml` under 'transformations', and the robot clearance |T_base_link__base_footprint[2, 3]|.
    Returns None (after a message) when there is no `cameras` directory."""
    cams = os.path.join(calib_path, 'cameras')
    if not os.path.exists(cams):
        print('No cameras calibration found in path {}'.format(cams))
        return None
    calib = {name[:-len('.yaml')]: read_yaml(os.path.join(cams, name)) for name in os.listdir(cams) if name.endswith('.yaml')}
    calib['transformations'] = read_yaml(os.path.join(calib_path, 'transformations.yaml'))
    T = np.asarray(calib['transformations']['T_base_link__base_footprint']['data'], dtype=np.float32).reshape(4, 4)
    calib['clearance'] = np.abs(T[2, 3])
    return calib


def compile_data(*args, **kwargs):
    raise NotImplementedError('compile_data builds the ROUGH datasets (monoforce.datasets), which are outside the scope of '
                              'monoforce_amd (DESIGN.md 7): feed LiftSplatShoot / DPhysics your own samples')
