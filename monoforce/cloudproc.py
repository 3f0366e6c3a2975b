"""`monoforce.cloudproc` -> monoforce_amd.cloudproc (estimate_heightmap on the HIP kernel)."""
from monoforce_amd.cloudproc import *  # noqa: F401,F403
