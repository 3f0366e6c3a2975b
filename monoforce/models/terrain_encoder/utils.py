"""`monoforce.models.terrain_encoder.utils` -> monoforce_amd.lss_utils."""
from monoforce_amd.lss_utils import *  # noqa: F401,F403
