"""`monoforce.losses` -> monoforce_amd.losses."""
from monoforce_amd.losses import hm_loss, physics_loss, rotation_difference, total_variation  # noqa: F401
