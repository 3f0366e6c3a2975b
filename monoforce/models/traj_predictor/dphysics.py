"""`monoforce.models.traj_predictor.dphysics` -> monoforce_amd.dphysics (HIP rollout)."""
from monoforce_amd.dphysics import (DPhysics, generate_controls, inertia_tensor, normalized, skew_symmetric,  # noqa: F401
                                    vw_to_track_vels)
