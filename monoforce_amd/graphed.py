"""The perception -> planning loop of the reference's ROS node as ONE hipGraph launch.

`monoforce_ros/nodes/monoforce_node.py:proc` does, per camera frame: terrain encoder forward (`lss.py:282-291`), 64 sampled
control sequences rolled out on the predicted terrain (`predict_paths`, :54-96), path costs and `argmin` (:91,126).  Eagerly
that is ~1100 kernel launches driven from Python (the encoder alone is launch-bound: 7.7 ms wall for 6.2 ms of GPU work);
with a fixed camera calibration everything after the image upload is a static sequence of kernels, so it is captured once
into a HIP graph and replayed per frame:

    camera features (MIOpen)  ->  BEV splat with the prepared plan (mf_bev_splat_fwd)  ->  BEV heads (MIOpen)
      -> terrain / friction pooled to the physics grid  ->  path-cost rollout of the sampled controls (mf_rollout_fwd, COST)
      -> costs, argmin  (all on the device; one host read of the winner at the end, if the caller wants it)

The frustum geometry (`get_geometry` uses `torch.inverse`, which synchronises) and the splat plan depend on the calibration
only and are built once, outside the graph.
"""
import torch

from .planner import costs_from_rows, sample_controls
from .capture import capture
from .splat import SplatPlan, _LiftPool

__all__ = ['GraphedTerrainPlanner']


class GraphedTerrainPlanner:
    def __init__(self, encoder, dphysics, calib, img_shape, n_trajs=None, cost='force', pose_stride=None, generator=None,
                 controls=None):
        """encoder: LiftSplatShoot (eval mode is set); dphysics: DPhysics on the same GPU (float32 fast math);
        calib = (rots, trans, intrins, post_rots, post_trans) of ONE sample [1, n_cams, ...]; img_shape = (n_cams, 3, H, W)."""
        assert cost in ('force', 'inclination')
        self.enc, self.dp, self.cost = encoder.eval(), dphysics, cost
        dev = next(encoder.parameters()).device
        self.device = dev
        cfg = dphysics.dphys_cfg
        self.n_trajs = n_trajs or cfg.n_sim_trajs
        self.controls = controls if controls is not None else sample_controls(self.n_trajs, cfg, dev, generator)
        self.pose_stride = pose_stride
        k = max(int(round(cfg.grid_res / float(encoder.dx[0]))), 1)       # scripts/train.py:93-99: encoder grid -> physics grid
        self.pool = torch.nn.AvgPool2d(kernel_size=k, stride=k) if k > 1 else torch.nn.Identity()
        with torch.no_grad():
            self.plan = encoder.splat_plan(*[t.to(dev) for t in calib])
        self.imgs = torch.zeros((1,) + tuple(img_shape), device=dev)
        self.graph = None
        self.out = None
        self._capture()

    def _pipeline(self):
        enc = self.enc
        B, N, C, imH, imW = self.imgs.shape
        depth, context = enc.camencode.get_depth_and_context(self.imgs.view(B * N, C, imH, imW))
        bev = enc.bevencode(_LiftPool.apply(depth, context, self.plan))
        z = self.pool(bev['terrain']).squeeze(1)                 # [1, H, W]: one terrain shared by all sampled rollouts
        mu = self.pool(bev['friction']).squeeze(1)
        r = self.dp.rollout_costs(z, self.controls, friction=mu, pose_stride=self.pose_stride, project=self.cost == 'inclination')
        costs = r['force_cost'] if self.cost == 'force' else costs_from_rows(r['cost_rows'], self.cost)
        best = torch.argmin(costs)
        return dict(terrain=bev['terrain'], friction=bev['friction'], costs=costs, best=best, Xs=r['Xs'], Rs=r['Rs'],
                    pose_steps=r['pose_steps'], best_controls=self.controls.index_select(0, best.view(1)))

    @torch.no_grad()
    def _capture(self):
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                            # warm-up off the capture: MIOpen picks its kernels here
            for _ in range(3):
                self._pipeline()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph):
            self.out = self._pipeline()

    @torch.no_grad()
    def eager(self, imgs):
        """The same pipeline launched op by op (reference for tests and for timing the graph against)."""
        self.imgs.copy_(imgs.view_as(self.imgs))
        return self._pipeline()

    @torch.no_grad()
    def __call__(self, imgs):
        """imgs [n_cams, 3, H, W] (or [1, n_cams, 3, H, W]): one graph launch; returns the graph's static output tensors
        (overwritten by the next call -- clone what must survive)."""
        self.imgs.copy_(imgs.view_as(self.imgs), non_blocking=True)
        self.graph.replay()
        return self.out
