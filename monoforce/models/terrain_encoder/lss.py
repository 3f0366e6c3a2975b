"""`monoforce.models.terrain_encoder.lss` -> monoforce_amd.terrain_encoder (HIP BEV splat)."""
from monoforce_amd.terrain_encoder import BevEncode, CamEncode, LiftSplatShoot, ScaledTanh, Up  # noqa: F401
