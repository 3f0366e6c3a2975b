"""`monoforce.models.traj_predictor.dphys_config` -> monoforce_amd.dphys_config."""
from monoforce_amd.dphys_config import DPhysConfig, get_points_from_robot_mesh, robot_geometry  # noqa: F401
