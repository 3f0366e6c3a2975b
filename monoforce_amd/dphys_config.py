"""DPhysConfig: robot geometry, terrain grid and simulation constants consumed by the rollout.

Mirror of `/root/reference/monoforce/src/monoforce/models/traj_predictor/dphys_config.py:77-153`: same constructor
signature `DPhysConfig(robot='marv', grid_res=0.1)`, same attribute names and defaults, so scripts that mutate
`traj_sim_time`, `dt`, `grid_res`, `d_max` after construction keep working.  Differences, all on the mesh side:
the reference loads `config/meshes/<robot>.obj` through open3d (`:8-35`); open3d is not a dependency here, so the
OBJ vertices are read and voxel-averaged by `points_from_obj`, and when no mesh file can be found (marv.obj is not
even in the reference checkout) a documented box-shaped stand-in body is generated instead.  Pass `robot_points=` /
`driving_parts=` to supply an exact body.
"""
import math
import os

import numpy as np
import torch
import yaml

ROBOT_MASS = {'tradr': 40.0, 'marv': 60.0, 'husky': 50.0}                       # dphys_config.py:84,98,112
JOINT_XYZ = {'tradr': (0.250, 0.272, 0.019), 'marv': (0.250, 0.272, 0.019), 'husky': (0.256, 0.285, 0.033)}


def _robot_family(robot):
    for fam in ('tradr', 'marv', 'husky'):
        if fam in robot:
            return fam
    raise ValueError(f'Robot {robot} not supported. Available robots: tradr, marv, husky')


def points_from_obj(path, voxel_size=0.1):
    """Vertices of a Wavefront OBJ, averaged per `voxel_size` cube (what open3d's voxel_down_sample returns,
    up to point order; cf. dphys_config.py:26-31)."""
    verts = []
    with open(path) as f:
        for line in f:
            if line.startswith('v '):
                verts.append([float(s) for s in line.split()[1:4]])
    v = np.asarray(verts, np.float64)
    if voxel_size:
        origin = v.min(0) - 0.5 * voxel_size
        key = np.floor((v - origin) / voxel_size).astype(np.int64)
        _, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        cnt = np.bincount(inv).astype(np.float64)
        v = np.stack([np.bincount(inv, weights=v[:, k]) / cnt for k in range(3)], 1)
    return torch.as_tensor(v, dtype=torch.float32)


def standin_points(robot):
    """Box-shaped stand-in body used when no mesh is available: a 0.1 m lattice over the chassis plus two (tradr) or
    four (marv/husky) rows of low track/flipper points at y = +-joint_y."""
    fam = _robot_family(robot)
    jx, jy, jz = JOINT_XYZ[fam]
    xs = np.arange(-0.4, 0.4001, 0.1)
    pts = [[x, y, 0.12] for x in xs for y in (-0.15, 0.0, 0.15)]                  # chassis top
    pts += [[x, s * jy, jz - 0.12] for x in xs for s in (-1, 1)]                  # tracks / flippers
    return torch.as_tensor(np.asarray(pts), dtype=torch.float32)


def find_mesh(robot):
    fam = _robot_family(robot)
    roots = [os.environ.get('MONOFORCE_MESH_DIR', ''), os.path.join(os.path.dirname(__file__), 'config', 'meshes')]
    for r in roots:
        p = os.path.join(r, f'{fam}.obj')
        if r and os.path.exists(p):
            return p
    return None


def get_points_from_robot_mesh(robot, voxel_size=0.1):
    """Body points of `robot` (dphys_config.py:8-35), from its mesh if one is found, else the stand-in body."""
    path = find_mesh(robot)
    if path:
        return points_from_obj(path, voxel_size)
    import warnings
    warnings.warn(f"no mesh '{_robot_family(robot)}.obj' found (searched $MONOFORCE_MESH_DIR and monoforce_amd/config/meshes): "
                  f"DPhysConfig('{robot}') simulates a box-shaped STAND-IN body -- different point count, inertia and driving "
                  'masks than the reference robot.  Point MONOFORCE_MESH_DIR at the reference\'s config/meshes, or pass '
                  'robot_points= / driving_parts=.', stacklevel=3)
    return standin_points(robot)


def robot_geometry(robot, x_points=None):
    """(points[N,3], driving_parts [bool[N]...], robot_size (s_x, s_y)) with the split rules of dphys_config.py:38-74."""
    if x_points is None:
        x_points = get_points_from_robot_mesh(robot)
    ext = x_points.max(0).values - x_points.min(0).values
    s_x, s_y = ext[0], ext[1]
    cog = x_points.mean(0)
    px, py, pz = x_points[:, 0], x_points[:, 1], x_points[:, 2]
    if robot in ('tradr', 'tradr2'):
        low = pz < cog[2]
        parts = [(py > cog[1] + s_y / 4.) & low, (py < cog[1] - s_y / 4.) & low]          # left, right track
    elif robot in ('marv', 'husky', 'husky_oru'):
        front, rear = px > cog[0] + s_x / 8., px < cog[0] - s_x / 8.
        left, right = py > cog[1] + s_y / 3., py < cog[1] - s_y / 3.
        parts = [front & left, front & right, rear & left, rear & right]                # fl, fr, rl, rr
    else:
        raise ValueError(f'Robot {robot} not supported. Available robots: tradr, marv, husky')
    return x_points, parts, (s_x, s_y)


class DPhysConfig:
    def __init__(self, robot='marv', grid_res=0.1, robot_points=None, driving_parts=None):
        fam = _robot_family(robot)
        self.robot = robot
        self.vel_max = 1.0       # m/s
        self.omega_max = 2.0     # rad/s
        self.robot_mass = ROBOT_MASS[fam]
        jx, jy, jz = JOINT_XYZ[fam]
        self.joint_positions = {'fl': [jx, jy, jz], 'fr': [jx, -jy, jz], 'rl': [-jx, jy, jz], 'rr': [-jx, -jy, jz]}
        self.joint_angles = {'fl': 0.0, 'fr': 0.0, 'rl': 0.0, 'rr': 0.0}
        if robot_points is not None and driving_parts is not None:
            pts = torch.as_tensor(robot_points, dtype=torch.float32)
            ext = pts.max(0).values - pts.min(0).values
            self.robot_points, self.robot_size = pts, (ext[0], ext[1])
            self.driving_parts = [torch.as_tensor(m, dtype=torch.bool) for m in driving_parts]
        else:
            self.robot_points, self.driving_parts, self.robot_size = robot_geometry(
                robot, None if robot_points is None else torch.as_tensor(robot_points, dtype=torch.float32))

        self.gravity = 9.81
        self.gravity_direction = torch.tensor([0., 0., -1.])

        # height map
        self.grid_res = grid_res
        self.r_min = 0.6
        self.d_max = 6.4
        self.h_max = 2.0
        ax = torch.arange(-self.d_max, self.d_max, self.grid_res)
        self.x_grid, self.y_grid = torch.meshgrid(ax, ax, indexing='ij')
        self.z_grid = torch.zeros_like(self.x_grid)
        self.friction = 1.0 * torch.ones_like(self.z_grid)
        self.stiffness = 50_000.
        self.damping = math.sqrt(4 * self.robot_mass * self.stiffness)     # critical damping
        self.hm_interp_method = None

        # trajectory shooting
        self.traj_sim_time = 5.0
        self.dt = 0.01
        self.n_sim_trajs = 64
        self.integration_mode = 'euler'
        self.use_odeint = True       # reference default: torchdiffeq fixed-grid euler semantics

    def __str__(self):
        return str(self.__dict__)

    def to_yaml(self, path):
        # like the reference (dphys_config.py:173-181) array-valued attributes are replaced by lists IN PLACE
        for k, v in list(self.__dict__.items()):
            if isinstance(v, (np.ndarray, torch.Tensor)):
                setattr(self, k, v.tolist())
        with open(path, 'w') as f:
            yaml.safe_dump(_plain(self.__dict__), f)

    def from_yaml(self, path):
        with open(path) as f:
            for k, v in yaml.load(f, Loader=yaml.FullLoader).items():
                setattr(self, k, v)

    def to_rosparam(self):
        import rospy
        for k, v in self.__dict__.items():
            rospy.set_param('~' + k, v.tolist() if isinstance(v, np.ndarray) else v)

    def from_rosparams(self, node_name):
        import rospy
        for k in rospy.get_param_names():
            if k.startswith('/' + node_name):
                setattr(self, k.split('/')[-1], rospy.get_param(k))


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, (np.ndarray, torch.Tensor)):
        return o.tolist()
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    return o
