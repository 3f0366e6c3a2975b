"""monoforce_amd: MI355X-native (gfx950) hot path of MonoForce -- DPhysics rollout + LSS BEV splat.

Importing the package is cheap and GPU-free; the HIP library (`monoforce_amd/csrc/libmonoforce_hip.so`)
is loaded on first use by `monoforce_amd._lib.lib()` and its absence is a hard error (no CPU fallback).
"""
__version__ = '0.1.0'


def __getattr__(name):
    """Lazy top-level names (keeps `import monoforce_amd` free of torch and of the HIP library)."""
    if name in ('DPhysics', 'generate_controls'):
        from . import dphysics
        return getattr(dphysics, name)
    if name == 'DPhysConfig':
        from .dphys_config import DPhysConfig
        return DPhysConfig
    if name == 'LiftSplatShoot':
        from .terrain_encoder import LiftSplatShoot
        return LiftSplatShoot
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
