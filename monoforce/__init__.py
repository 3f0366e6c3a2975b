"""Import-surface shim: the reference's module paths (`monoforce.models.traj_predictor.dphysics`, ...) re-exporting the
MI355X-native implementation in `monoforce_amd`, so the reference's `scripts/run.py` / `train.py` import unchanged.
Only the hot path (DPhysics rollout, LSS terrain encoder, losses) and the few helpers those scripts need are provided;
datasets / ROS / visualisation are out of scope."""
