"""`monoforce.models.terrain_encoder.utils` -> monoforce_amd.lss_utils (pooling helpers) + monoforce_amd.img_utils (images)."""
from monoforce_amd.lss_utils import *  # noqa: F401,F403
from monoforce_amd.img_utils import *  # noqa: F401,F403
