"""monoforce_amd: MI355X-native (gfx950) hot path of MonoForce -- DPhysics rollout + LSS BEV splat.

Importing the package is cheap and GPU-free; the HIP library (`monoforce_amd/csrc/libmonoforce_hip.so`)
is loaded on first use by `monoforce_amd._lib.lib()` and its absence is a hard error (no CPU fallback).
"""
__version__ = '0.1.0'
