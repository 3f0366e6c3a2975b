"""Train-step drivers for the hot path: the call pattern of `scripts/fit_terrain.py:53-62` (optimise a terrain through
the physics) and `scripts/train.py:377-410` (physics loss on predicted terrain), sharded over GPUs.
"""
import torch

from . import dist as mfdist
from .losses import physics_loss


class TerrainFitProblem:
    """B rollouts on ONE shared terrain; loss = `physics_loss` against ground-truth poses at 10 Hz (50 of 500 steps,
    `datasets/rough.py:217,238`); gradient flows to the terrain height and friction grids.

    The ground truth is a rollout of the same controls on a "true" terrain, generated once with the HIP forward.
    """

    def __init__(self, dphysics, z_true, mu_true, controls, gt_every=10):
        self.dp = dphysics
        self.controls = controls
        B, T = controls.shape[:2]
        cfg = dphysics.dphys_cfg
        with torch.no_grad():
            (Xs, Xds, Rs, Om), _ = dphysics(z_true.unsqueeze(0), controls, friction=mu_true.unsqueeze(0))
        full_ts = torch.linspace(0, cfg.traj_sim_time, int(cfg.traj_sim_time / cfg.dt), device=controls.device)[:T]
        self.pred_ts = full_ts.unsqueeze(0).expand(B, -1)
        sel = torch.arange(gt_every - 1, T, gt_every, device=controls.device)
        self.gt_ts = full_ts[sel].unsqueeze(0).expand(B, -1).contiguous()
        self.states_gt = [Xs[:, sel].contiguous(), Xds[:, sel].contiguous(), Rs[:, sel].contiguous(), Om[:, sel].contiguous()]
        self.bucket = None

    def step(self, z, mu):
        """One forward + backward: returns the loss; leaves d loss / d z, d mu (summed over ALL ranks' rollouts) in .grad."""
        z.grad = None
        mu.grad = None
        states, _ = self.dp(z.unsqueeze(0), self.controls, friction=mu.unsqueeze(0))
        loss = physics_loss(states_pred=states, states_gt=self.states_gt, pred_ts=self.pred_ts, gt_ts=self.gt_ts)
        loss.backward()
        # the one exchange step of the backward: 2 x H x W floats over RCCL
        self.bucket = mfdist.allreduce_sum_([z.grad, mu.grad], self.bucket)
        return loss
