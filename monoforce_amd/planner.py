"""Trajectory shooting on top of the rollout kernels: sample control sequences, roll them out on one terrain, score the
paths, pick the cheapest -- the device-side part of the reference's planning nodes (no ROS here):

  * controls:     `MonoForce.init_controls`  (/root/reference/monoforce_ros/nodes/monoforce_node.py:41-52): half the
                  trajectories forward (v in [v_max/2, v_max]), half backward, w in [-omega_max, omega_max], constant in time
                  (`generate_controls`, dphysics.py:42-72) -- sampled on the GPU here
  * path costs:   force based   `norm(F_springs).std(points).std(time)`   (monoforce_node.py:91)
                  inclination   `mean|roll| + mean|pitch|`                 (monoforce_ros/nodes/diff_physics.py:263-266;
                                 roll/pitch = scipy `as_euler('xyz')` of the predicted rotations)
  * selection:    `argmin(path_costs)` (monoforce_node.py:126)

All rollouts share ONE height/friction map (the rollout kernels' shared-map path), so thousands of samples cost one
512 KiB map read.  By default the rollout runs in the kernel's path-cost mode (`DPhysics.rollout_costs`): per step it writes
one 16-byte cost row (the last row of R and the std over the contact points of |F_spring|) and keeps every
`pose_stride`-th pose instead of the full 180-byte output row; `fused=False` uses the full outputs.
"""
import torch

__all__ = ['sample_controls', 'force_path_cost', 'inclination_path_cost', 'nearest_rotation_row2', 'costs_from_rows', 'TrajectoryShooter']


def sample_controls(n_trajs, cfg, device, generator=None):
    """[n_trajs, T, 2] constant-in-time (v, w): first half forward, second half backward (monoforce_node.py:41-52).  Returned
    as the [n_trajs, 1, 2] samples EXPANDED over time (stride 0): the rollout kernels read one (v, w) per trajectory instead of
    a 2 T-float row; `.contiguous()` gives the reference's materialised tensor."""
    T = int(cfg.traj_sim_time / cfg.dt)
    n_f = n_trajs // 2
    u = torch.rand(n_trajs, 2, device=device, generator=generator)
    v = torch.empty(n_trajs, device=device)
    v[:n_f] = cfg.vel_max / 2 + u[:n_f, 0] * (cfg.vel_max / 2)
    v[n_f:] = -cfg.vel_max + u[n_f:, 0] * (cfg.vel_max / 2)
    w = -cfg.omega_max + u[:, 1] * (2 * cfg.omega_max)
    return torch.stack([v, w], -1).unsqueeze(1).expand(-1, T, -1)


def force_path_cost(F_springs):
    """[B,T,N,3] -> [B]: std over time of the std over contact points of |F_spring| (monoforce_node.py:91)."""
    return torch.norm(F_springs, dim=-1).std(dim=-1).std(dim=-1)


def costs_from_rows(cost_rows, kind):
    """[B,T,4] cost rows of the path-cost kernel -> [B] path costs (same formulas as the two functions above)."""
    if kind == 'force':
        return cost_rows[..., 3].std(dim=-1)
    pitch = torch.asin(torch.clamp(-cost_rows[..., 0], -1.0, 1.0))
    roll = torch.atan2(cost_rows[..., 1], cost_rows[..., 2])
    return roll.abs().mean(dim=-1) + pitch.abs().mean(dim=-1)


def inclination_path_cost(Rs):
    """[B,T,3,3] -> [B]: mean |roll| + mean |pitch| with roll/pitch of the extrinsic xyz Euler decomposition
    R = Rz(yaw) Ry(pitch) Rx(roll) (what scipy's `Rotation.as_euler('xyz')` returns; diff_physics.py:263-266)."""
    # scipy's `from_matrix` projects a non-orthonormal matrix onto the nearest rotation first (U V^T of its SVD); the default
    # integrator's R drifts off SO(3), so the same projection is applied here (the path-cost kernel does it in registers)
    r2 = nearest_rotation_row2(Rs)
    pitch = torch.asin(torch.clamp(-r2[..., 0], -1.0, 1.0))
    roll = torch.atan2(r2[..., 1], r2[..., 2])
    return roll.abs().mean(dim=-1) + pitch.abs().mean(dim=-1)


def nearest_rotation_row2(Rs, iters=3):
    """Third row of the polar factor U V^T of [..., 3, 3] near-rotations by Newton's iteration X <- (X + X^-T) / 2 with the
    closed-form 3x3 inverse (rows of X^-T = cross products of the rows of X over det); a batched SVD of 3e7 matrices is not
    an option.  Quadratic convergence: 3 steps from |R R^T - I| = 0.06 are exact to float32."""
    X = Rs
    for _ in range(iters):
        r0, r1, r2 = X[..., 0, :], X[..., 1, :], X[..., 2, :]
        c0, c1, c2 = torch.linalg.cross(r1, r2), torch.linalg.cross(r2, r0), torch.linalg.cross(r0, r1)
        det = (r0 * c0).sum(-1, keepdim=True)
        X = 0.5 * (X + torch.stack([c0, c1, c2], dim=-2) / det.unsqueeze(-1))
    return X[..., 2, :]


class TrajectoryShooter:
    def __init__(self, dphysics, n_trajs=None, cost='inclination', fused=True, pose_stride=None):
        assert cost in ('inclination', 'force')
        self.dp = dphysics
        self.cfg = dphysics.dphys_cfg
        self.n_trajs = n_trajs or self.cfg.n_sim_trajs
        self.cost = cost
        self.fused = fused and not dphysics.precise      # path-cost kernel: float32 fast math
        self.pose_stride = pose_stride

    @torch.no_grad()
    def shoot(self, z_grid, friction=None, pose0=None, controls=None, generator=None):
        """z_grid [H,W] (or [1,H,W]); pose0 optional 4x4 start pose shared by all samples.
        Returns dict(controls, Xs, Rs, costs, best) -- `best` is the index of the cheapest path; with the fused path-cost
        kernel Xs / Rs hold every `pose_stride`-th pose (+ the final one) and `pose_steps` their step indices."""
        dev = z_grid.device
        if controls is None:
            controls = sample_controls(self.n_trajs, self.cfg, dev, generator)
        B = controls.shape[0]
        z = z_grid if z_grid.dim() == 3 else z_grid.unsqueeze(0)
        mu = None if friction is None else (friction if friction.dim() == 3 else friction.unsqueeze(0))
        state = None
        if pose0 is not None:
            x = pose0[:3, 3].to(dev).repeat(B, 1)
            state = (x, torch.zeros_like(x), pose0[:3, :3].to(dev).repeat(B, 1, 1).contiguous(), torch.zeros_like(x))   # monoforce_node.py:67-72
        if self.fused and z.dtype == torch.float32:
            out = self.dp.rollout_costs(z, controls, state=state, friction=mu, pose_stride=self.pose_stride,
                                        project=self.cost == 'inclination')
            costs = out['force_cost'] if self.cost == 'force' else costs_from_rows(out['cost_rows'], self.cost)
            return dict(controls=controls, Xs=out['Xs'], Rs=out['Rs'], pose_steps=out['pose_steps'], costs=costs,
                        best=int(torch.argmin(costs)))
        need_forces = self.cost == 'force'
        old = self.dp.return_forces
        self.dp.return_forces = need_forces          # inclination cost only needs the states: states-only kernel
        try:
            (Xs, Xds, Rs, Om), (Fs, Ff) = self.dp(z, controls, state=state, friction=mu)
        finally:
            self.dp.return_forces = old
        costs = force_path_cost(Fs) if need_forces else inclination_path_cost(Rs)
        return dict(controls=controls, Xs=Xs, Rs=Rs, costs=costs, best=int(torch.argmin(costs)))
