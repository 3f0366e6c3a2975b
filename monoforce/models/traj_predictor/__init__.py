from .dphys_config import DPhysConfig  # noqa: F401
from .dphysics import DPhysics  # noqa: F401
