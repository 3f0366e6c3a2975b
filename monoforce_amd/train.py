"""Train-step drivers for the hot path: the call pattern of `scripts/fit_terrain.py:53-62` (optimise a terrain through
the physics) and `scripts/train.py:377-410` (physics loss on predicted terrain), sharded over GPUs.
"""
import os

import torch

from . import dist as mfdist
from .capture import capture
from .losses import nearest_steps, physics_loss, physics_loss_fused


class TerrainFitProblem:
    """B rollouts on ONE shared terrain; loss = `physics_loss` against ground-truth poses at 10 Hz (50 of 500 steps,
    `datasets/rough.py:217,238`); gradient flows to the terrain height and friction grids.

    The ground truth is a rollout of the same controls on a "true" terrain, generated once with the HIP forward.
    """

    def __init__(self, dphysics, z_true, mu_true, controls, gt_every=10, fused_loss=True, graph=False, loss_in_kernel=True):
        self.dp = dphysics
        self.fused_loss = fused_loss      # mf_physics_loss_* instead of ~25 small ATen kernels (same value and gradient)
        # physics_loss inside the rollout's own two launches (DPhysics.physics_loss_rollout): a step is four
        # launches instead of six and the [B,T,3] gradient rows are never built; where the library cannot fuse, the route above
        self.loss_in_kernel = bool(loss_in_kernel) and fused_loss and controls.is_cuda
        self.loss_value_in_backward = os.environ.get('MF_LOSS_VALUE_IN_BACKWARD', '1') != '0'      # (A/B switch; DPhysics.physics_loss_rollout)
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
        self.nearest = nearest_steps(self.pred_ts, self.gt_ts).to(torch.int32)      # time stamps are fixed: computed once
        self.spec = dphysics.loss_spec(full_ts[sel], gamma=0.9, n_steps=T) if self.loss_in_kernel else None
        self.bucket = None
        self._one = None
        self.fast_exchange = None         # True once a step exchanged gradients and loss in place (one collective, no copies)
        # graph = True: forward + loss + backward are captured once and replayed as ONE hipGraph launch per step -- the host side
        # of a step is ~0.3 ms of Python and launch calls against ~0.55 ms of kernels at the BASELINE shape (a loaded or slower
        # host makes the step host-bound) and against ~0.15 ms at 256 rollouts x 100 steps (0.38 -> 0.15 ms per step replayed).
        self.graph = bool(graph) and fused_loss and controls.is_cuda
        self._captured = None

    def _seed(self, loss):
        if self._one is None or self._one.dtype != loss.dtype or self._one.device != loss.device:
            self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        return self._one

    def _exchange(self, z, mu, loss):
        """The one exchange step of the backward: 2 x H x W floats (+ the loss scalar) over RCCL.  Every rank's loss is the MEAN
        over its own rollouts (equal shares), so the mean over ranks is the gradient of the global-mean loss: the step does
        not depend on the number of GPUs."""
        if not mfdist.active():
            return loss.detach()
        # The shared-map backward hands both gradients out as views of one buffer with a spare scalar behind them: the loss goes
        # there and the buffer is averaged in place by one collective -- no pack, divide or unpack launches around it (eight
        # ~10 us launches on a 0.56 ms step).
        flat = mfdist.shared_flat_buffer([z.grad, mu.grad])
        self.fast_exchange = flat is not None
        if flat is not None:
            flat[-1:].copy_(loss.detach().reshape(1))
            mfdist.allreduce_mean_inplace_(flat)
            return flat[-1]
        lbuf = loss.detach().reshape(1).clone()
        self.bucket = mfdist.allreduce_sum_([z.grad, mu.grad, lbuf], self.bucket, average=True)
        return lbuf[0]

    def step(self, z, mu, eager=False):
        """One forward + backward: returns the loss averaged over ALL ranks' rollouts; leaves its gradient w.r.t. z, mu in .grad.
        `eager=True` runs this one step launch by launch even in graph mode (bench.py brackets the kernels of such steps with
        HIP events)."""
        if self.graph and not eager:
            return self._step_graph(z, mu)
        z.grad = None
        mu.grad = None
        loss = self._loss(z, mu)
        loss.backward(self._seed(loss))   # (the default seed is a fresh ones_like: one more launch in front of the backward)
        return self._exchange(z, mu, loss)

    def _loss(self, z, mu):
        if self.loss_in_kernel:
            # (the step always runs the backward next and hands the value out afterwards: the backward launch forms it too)
            return self.dp.physics_loss_rollout(z.unsqueeze(0), self.controls, self.states_gt[0], self.spec, friction=mu.unsqueeze(0),
                                                value_in_backward=self.loss_value_in_backward)[0]
        # (the fit discards the forces, scripts/fit_terrain.py:56: the forward writes the states only -- for the reference's 223-point
        #  body that is 36 of 241 MB per launch; `return_forces` is restored for the module's other callers)
        keep, self.dp.return_forces = self.dp.return_forces, False
        try:
            states, _ = self.dp(z.unsqueeze(0), self.controls, friction=mu.unsqueeze(0))
        finally:
            self.dp.return_forces = keep
        loss_fn = physics_loss_fused if self.fused_loss else physics_loss
        return loss_fn(states, self.states_gt, self.pred_ts, self.gt_ts, nearest=self.nearest if self.fused_loss else self.nearest.long())

    def _step_graph(self, z, mu):
        cap = self._captured
        if cap is None or cap['z'] is not z or cap['mu'] is not mu:
            try:
                cap = self._captured = self._capture(z, mu)
            except RuntimeError as e:       # a capture the runtime refuses (e.g. another thread's call landed in it): launch by launch
                import warnings
                warnings.warn(f'TerrainFitProblem: hipGraph capture failed ({str(e).splitlines()[0][:120]}); running launch by launch')
                self.graph, self._captured = False, None
                torch.cuda.synchronize(z.device)
                return self.step(z, mu, eager=True)
        cap['graph'].replay()
        z.grad, mu.grad = cap['gz'], cap['gmu']
        return self._exchange(z, mu, cap['loss'])

    def _capture(self, z, mu):
        dev = z.device

        def fwd_bwd():
            loss = self._loss(z, mu)
            gz, gmu = torch.autograd.grad(loss, [z, mu], grad_outputs=self._seed(loss))
            return loss.detach(), gz, gmu

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):          # warm-up on the capture stream: per-stream workspaces (tickets, gradient pool) exist
            for _ in range(2):
                fwd_bwd()
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        # thread_local: only this thread's calls are policed during the capture (a process-group watchdog polling its events
        # from another thread must not abort it); the backward's launches come from the autograd thread and are captured with
        # the stream they run on either way
        with capture(g, stream=s, capture_error_mode='thread_local'):
            loss, gz, gmu = fwd_bwd()
        return dict(graph=g, z=z, mu=mu, loss=loss, gz=gz, gmu=gmu)


class EncoderTrainStep:
    """The reference's end-to-end training step (`scripts/train.py:377-410, 231-246, 142-185`) on one GPU / one rank:

        terrain = encoder(imgs, rots, trans, intrins, post_rots, post_trans)        # LiftSplatShoot, HIP BEV splat inside
        loss = w_g * hm_loss(geom) + w_t * hm_loss(terrain) + w_p * physics_loss(dphysics(terrain, friction), gt)
        loss.backward(); [all-reduce encoder grads over RCCL]; clip_grad_norm_(1.0); Adam(lr, betas=(0.8, 0.999), wd=1e-7)

    The predicted terrain / friction of each sample is shared by `rollouts_per_sample` control sequences (BASELINE config 4:
    one 4-camera sample, 1024 rollouts), which maps onto the rollout kernels' shared-map path.
    """

    def __init__(self, encoder, dphysics, lr=1e-3, geom_weight=1.0, terrain_weight=1.0, phys_weight=1.0, graph=False):
        self.enc, self.dp = encoder, dphysics
        # graph = True: the whole step -- encoder forward, lift-splat, heads, staging, rollout + loss, backward, clip, Adam -- is
        # captured once per batch object and replayed as ONE hipGraph launch: ~4000 launches of ~17 ms of kernels otherwise spend a
        # quarter of the 23 ms step in launch gaps.  Needs the batch tensors to stay the same objects (a fixed-rig loop would copy
        # each sample into them: stamps included, the captured step rebuilds its stamp tables on the device).  Several ranks: two
        # graphs around the live exchange (`_step_graph`).
        self.graph = bool(graph)
        self._cap = None
        # coarser grid for the physics than for the encoder: average pooling (scripts/train.py:93-99, 233-235)
        k = max(int(round(dphysics.dphys_cfg.grid_res / float(encoder.dx[0]))), 1)
        self.terrain_preproc = torch.nn.AvgPool2d(kernel_size=k, stride=k) if k > 1 else torch.nn.Identity()
        self.pool_k = k
        self.fused_stage = True        # terrain = geom - diff, both poolings and the (z, mu) interleave as one kernel
        self.loss_in_kernel = True     # physics_loss inside the rollout launches where the stamps are shared by the rollouts
        self._specs = {}        # per batch STRUCTURE (shape, shared stamps or not): the fused-loss tables, refreshed from each batch's stamps
        self._nearest = None    # the unfused route's nearest-step table of the last batch (same tensor objects and versions only)
        self.w = (geom_weight, terrain_weight, phys_weight)
        # train.py:374-375; the fused (single multi-tensor kernel) implementation where the parameters live on the GPU
        on_gpu = all(p.is_cuda for p in encoder.parameters())
        self.opt = torch.optim.Adam(encoder.parameters(), lr=lr, betas=(0.8, 0.999), weight_decay=1e-7, fused=on_gpu,
                                    capturable=bool(graph) and on_gpu)
        self.params = [p for p in encoder.parameters() if p.requires_grad]
        # gradients live in a few flat buckets whose all-reduce (RCCL) starts from autograd hooks while the backward is still
        # running; one zero fill per bucket replaces the per-parameter zero_grad
        self.buckets = mfdist.GradBuckets(self.params, bucket_mb=32.0, average=True)

    def losses(self, batch):
        from .losses import hm_loss
        (inputs, hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, nearest) = batch
        if self.fused_stage and inputs[0].is_cuda:
            # heads -> (terrain, pooled z, pooled mu, interleaved (z, mu)) in one kernel (terrain_stage.py); same values
            terrain = self.enc(*inputs, stage_k=self.pool_k)
            z, mu = terrain['terrain_phys'], terrain['friction_phys']
        else:
            terrain = self.enc(*inputs)
            # one predicted map per sample; B controls per sample share it ([1,H,W] map + [B,T,2] controls)
            z = self.terrain_preproc(terrain['terrain']).squeeze(1)
            mu = self.terrain_preproc(terrain['friction']).squeeze(1)
        l_geom = hm_loss(terrain['geom'], hm_geom[:, 0:1], hm_geom[:, 1:2])                  # train.py:388-392
        l_terr = hm_loss(terrain['terrain'], hm_terrain[:, 0:1], hm_terrain[:, 1:2])         # train.py:395-398
        x0 = pose0[:, :3, 3].clone()
        state0 = (x0, torch.zeros_like(x0), pose0[:, :3, :3].contiguous(), torch.zeros_like(x0))   # train.py:237-241
        spec = self._loss_spec(gt_ts, controls.shape[1])
        if spec is not None:      # losses.py:102-127 inside the rollout's own launches
            l_phys = self.dp.physics_loss_rollout(z, controls, states_gt[0], spec, state=state0, friction=mu)[0] + spec.poison
        else:
            states, _ = self.dp(z_grid=z, controls=controls, state=state0, friction=mu)
            l_phys = physics_loss_fused(states, states_gt, pred_ts, gt_ts, nearest=nearest)       # losses.py:102-127 on mf_physics_loss_*
        return l_geom, l_terr, l_phys

    def compute_losses(self, batch):
        """`TrainerLSS.compute_losses` (scripts/train.py:377-410) on the reference's ROUGH sample tuple, in ITS order
        (datasets/rough.py:651-663, as the DataLoader collates it):
            (imgs, rots, trans, intrins, post_rots, post_trans, hm_geom, hm_terrain, control_ts, controls, pose0, traj_ts, Xs, Xds, Rs, Omegas)
        -> (loss_geom, loss_terrain, loss_phys).  One predicted map per sample; `controls` may hold one trajectory per sample
        ([Bs,T,2], the reference's case: per-rollout maps) or, with a single sample, many ([B,T,2]: they share its map)."""
        (imgs, rots, trans, intrins, post_rots, post_trans, hm_geom, hm_terrain, control_ts, controls, pose0,
         traj_ts, Xs, Xds, Rs, Omegas) = batch
        return self.losses(((imgs, rots, trans, intrins, post_rots, post_trans), hm_geom, hm_terrain, controls, pose0,
                            [Xs, Xds, Rs, Omegas], control_ts, traj_ts, self._nearest_for(control_ts, traj_ts)))

    @staticmethod
    def _same_tensor(entry, *tensors):
        """True when `entry` was built from exactly these tensor OBJECTS at their current versions.  The entry holds the tensors, so
        their memory cannot have been recycled for another batch; an in-place refill (`copy_`) bumps `_version`."""
        return (entry is not None and len(entry['src']) == len(tensors)
                and all(a is b and v == b._version for (a, v), b in zip(entry['src'], tensors)))

    def _nearest_for(self, control_ts, traj_ts):
        """`nearest_steps(control_ts, traj_ts)` ([N,T2,T1] argmin, losses.py:116) for THIS batch's stamps.  The reference's
        `traj_ts` are measured per sample (datasets/rough.py:261-296), so nothing is reused across batches by address: the
        result is kept only for the same tensor objects at the same versions, and always rebuilt (on the device, no host round
        trip) inside a graph capture, so that a replay follows stamps copied into the batch tensors."""
        e = self._nearest
        if torch.cuda.is_current_stream_capturing() or not self._same_tensor(e, control_ts, traj_ts):
            if control_ts.stride(0) == 0 and traj_ts.stride(0) == 0 and control_ts.shape[0] > 1:      # one row expanded: one argmin
                near = nearest_steps(control_ts[:1], traj_ts[:1]).to(torch.int32).expand(traj_ts.shape[0], -1)
            else:
                near = nearest_steps(control_ts, traj_ts).to(torch.int32)
            e = self._nearest = dict(src=[(control_ts, control_ts._version), (traj_ts, traj_ts._version)], near=near)
        return e['near']

    def _loss_spec(self, gt_ts, T):
        """LossSpec of a batch whose ground-truth stamps are the same for every rollout (one row expanded, one rollout, or rows found
        equal when this batch structure was first seen); None = per-rollout stamps: the unfused physics loss.  WHICH route a
        batch structure takes is decided once (one host round trip); the tables themselves are rebuilt on the device from every
        new batch's stamps (`LossSpec.refresh_`: same tensor objects at the same versions are the only thing reused; inside a graph
        capture always), and a later batch the fused route cannot carry (rows that differ, two stamps on one row) poisons the
        loss with NaN instead of being scored against another batch's stamps."""
        if not (self.loss_in_kernel and gt_ts.is_cuda):
            return None
        structural = gt_ts.stride(0) == 0 or gt_ts.shape[0] == 1
        key = (tuple(gt_ts.shape), structural, T)
        e = self._specs.get(key)
        if e is None:
            shared = structural or bool((gt_ts == gt_ts[:1]).all())
            spec = self.dp.loss_spec(gt_ts[0], gamma=0.9, n_steps=T) if shared else None
            if spec is not None and not spec.fusable:
                spec = None
            while len(self._specs) >= 8:      # batch structures are few; bounded all the same
                self._specs.pop(next(iter(self._specs)))
            e = self._specs[key] = dict(spec=spec, src=[(gt_ts, gt_ts._version)])
            if spec is None or not torch.cuda.is_current_stream_capturing():
                return spec
        spec = e['spec']
        if spec is None:
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing or not self._same_tensor(e, gt_ts):
            if not capturing and not structural and not bool((gt_ts == gt_ts[:1]).all()):
                # a later batch of this shape whose rollouts carry DIFFERENT stamps (the reference's per-sample stamps,
                # datasets/rough.py:261-296): the unfused loss, like the per-tensor check of round 3 -- not a NaN.  (One host round
                # trip per new batch of a non-expanded stamp tensor; inside a capture the decision is frozen and the poison stands.)
                return None
            spec.refresh_(gt_ts)
            e['src'] = [(gt_ts, gt_ts._version)]
        return spec

    def exchange_only(self):
        """The step's collectives alone, on the buckets as they stand (bench.py: `comm_ms`)."""
        self.buckets.exchange()

    def step(self, batch, eager=False):
        """One training step; `eager=True` runs this one launch by launch even in graph mode.  Several ranks: the step replays
        as TWO graphs around its one exchange (`_step_graph`)."""
        if self.graph and not eager:
            return self._step_graph(batch)
        return self._step_eager(batch)

    # -- graph mode ------------------------------------------------------------------------------------------------------------
    def _snapshot(self):
        """Everything a step changes: parameters and buffers (batch-norm statistics) of the encoder, the optimizer's state."""
        import copy
        return ([t.detach().clone() for t in self.enc.state_dict().values()], copy.deepcopy(self.opt.state_dict()['state']))

    def _restore(self, snap):
        """Back to `_snapshot()` IN PLACE (captured graphs and Adam's capturable state keep their addresses).  Optimizer state that
        did not exist at the snapshot (the first step creates it) is reset to what a fresh optimizer holds: zeros."""
        tensors, opt_state = snap
        with torch.no_grad():
            for t, old in zip(self.enc.state_dict().values(), tensors):
                t.copy_(old)
            ids = {id(p): i for i, p in enumerate(p for g in self.opt.param_groups for p in g['params'])}
            for p, st in self.opt.state.items():
                old = opt_state.get(ids[id(p)])
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(old[k]) if old is not None else v.zero_()

    def _step_graph(self, batch):
        """The step as hipGraph replays, captured once per batch object.  One rank: ONE graph.  Several ranks (collectives are not
        captured): graph A = zero the buckets, forward, backward, pack the buckets | the bucket all-reduces, launched live |
        graph B = average, clip, Adam.  The exchange then starts after the backward instead of from its hooks -- ~55 MB over xGMI
        against the ~3 ms of launch gaps a 1000-launch step pays launch by launch.
        The warm-up steps in front of the capture (MIOpen picks its solvers, workspaces and Adam state come into being) run on
        real data but leave no trace: parameters, batch-norm statistics and optimizer state are restored afterwards, so the first
        `step()` applies exactly ONE update, like the launch-by-launch step."""
        cap = self._cap
        split = mfdist.active()
        if cap is None or cap['batch'] is not batch or cap['split'] != split:
            dev = next(self.enc.parameters()).device
            snap = self._snapshot()
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            self.buckets.defer = split
            try:
                with torch.cuda.stream(s):
                    for _ in range(3):
                        self._step_eager(batch)
                    self._restore(snap)
                torch.cuda.current_stream(dev).wait_stream(s)
                g, g2 = torch.cuda.CUDAGraph(), None
                with capture(g, stream=s, capture_error_mode='thread_local'):
                    out = self._forward_backward(batch) if split else self._step_eager(batch)
                if split:
                    g2 = torch.cuda.CUDAGraph()
                    with capture(g2, stream=s, pool=g.pool(), capture_error_mode='thread_local'):
                        self._apply()
            except RuntimeError as e:
                import warnings
                if os.environ.get('MF_DEBUG_CAPTURE'):
                    import traceback
                    traceback.print_exc()
                warnings.warn(f'EncoderTrainStep: hipGraph capture failed ({str(e).splitlines()[0][:160]}); running launch by launch')
                # (several ranks: the others may have captured and will call `buckets.exchange()` -- bucket-INDEX order -- between
                #  their two replays; this rank keeps the deferred exchange too, so every rank issues the same collectives in the same
                #  order.  The hook-launched exchange orders them by gradient readiness: mixed with the above it can pair different
                #  buckets across ranks.  ADVICE r4.)
                self.graph, self.buckets.defer = False, split
                torch.cuda.synchronize(dev)
                self._restore(snap)
                return self._step_eager(batch)
            # (the captured launches read the device tables of the LossSpecs in use by address: pinned here, so that the eviction
            #  in `_loss_spec` cannot free them under a live graph)
            cap = self._cap = dict(graph=g, apply=g2, batch=batch, out=out, split=split,
                                   pinned_specs=[e['spec'] for e in self._specs.values() if e['spec'] is not None])
        cap['graph'].replay()
        if cap['split']:
            self.buckets.exchange()
            cap['apply'].replay()
        return cap['out']

    # -- the step itself ---------------------------------------------------------------------------------------------------
    def _forward_backward(self, batch):
        """Zero the buckets, forward, backward: afterwards every bucket holds this rank's packed gradients; their all-reduces are
        running (launched from the hooks) unless the buckets are in deferred mode."""
        self.buckets.zero()
        l_geom, l_terr, l_phys = self.losses(batch)
        loss = self.w[0] * l_geom + self.w[1] * l_terr + self.w[2] * l_phys
        loss.backward()                      # bucket all-reduces are launched from the hooks as their gradients complete
        self.buckets.pack()                  # (whatever the hooks have not packed)
        return loss.detach(), (l_geom.detach(), l_terr.detach(), l_phys.detach())

    def _apply(self):
        """Average the exchanged buckets, clip, Adam (scripts/train.py:165-168)."""
        self.buckets.finish()                # wait + average (no-op for one process)
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=1.0)                   # train.py:167
        self.opt.step()

    def _step_eager(self, batch):
        out = self._forward_backward(batch)
        if self.buckets.defer:
            self.buckets.exchange()
        self._apply()
        return out


def synthetic_rough_batch(encoder, dphysics, n_rollouts, device, seed=0, img_hw=(256, 512)):
    """The same synthetic sample as `synthetic_encoder_batch`, as the reference's 16-tuple (datasets/rough.py:651-663 order) --
    what `EncoderTrainStep.compute_losses` / scripts/train.py:377-410 consume."""
    (inputs, hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, _near) = synthetic_encoder_batch(
        encoder, dphysics, n_rollouts, device, seed=seed, img_hw=img_hw)
    return (*inputs, hm_geom, hm_terrain, pred_ts, controls, pose0, gt_ts, *states_gt)


def synthetic_encoder_batch(encoder, dphysics, n_rollouts, device, seed=0, img_hw=(256, 512)):
    """Synthetic stand-in for a ROUGH sample (`datasets/rough.py:651-663`): 4 cameras, random images, a "true" bump
    terrain as height-map labels, ground-truth poses from rolling the controls out on that terrain (HIP forward)."""
    from . import synthetic as syn
    from .losses import nearest_steps
    g = torch.Generator().manual_seed(seed)
    H, W = img_hw
    imgs = torch.randn(1, 4, 3, H, W, generator=g).to(device)
    rig = [t.to(device) for t in syn.lss_camera_rig(1, 4, H, W, 300.0)]
    cfg = dphysics.dphys_cfg
    nxy = int(encoder.nx[0])
    res = float(encoder.dx[0])
    z_true = syn.bump_terrain(syn.bump_params(100 + seed), cfg.d_max, res).to(device)
    mu_true = syn.wave_friction(cfg.d_max, res).to(device)
    assert z_true.shape == (nxy, nxy)
    ones = torch.ones_like(z_true)
    hm_geom = torch.stack([z_true.clamp(-1, 1), ones]).unsqueeze(0)          # [1, 2, H, W]: value, weight (train.py:390)
    hm_terrain = torch.stack([z_true.clamp(-1, 1) * 0.9, ones]).unsqueeze(0)
    T = int(cfg.traj_sim_time / cfg.dt)
    controls = syn.const_controls(n_rollouts, T, seed=seed).to(device)
    pose0 = torch.eye(4, device=device).repeat(n_rollouts, 1, 1)
    with torch.no_grad():
        st = (pose0[:, :3, 3].clone(), torch.zeros(n_rollouts, 3, device=device), pose0[:, :3, :3].contiguous(),
              torch.zeros(n_rollouts, 3, device=device))
        (Xs, Xds, Rs, Om), _ = dphysics(z_true.unsqueeze(0), controls, state=st, friction=mu_true.unsqueeze(0))
    ts = torch.linspace(0, cfg.traj_sim_time, T, device=device)
    sel = torch.arange(9, T, 10, device=device)                               # 10 Hz poses (rough.py:217,238)
    pred_ts = ts.unsqueeze(0).expand(n_rollouts, -1)
    gt_ts = ts[sel].unsqueeze(0).expand(n_rollouts, -1).contiguous()
    states_gt = [Xs[:, sel].contiguous(), Xds[:, sel].contiguous(), Rs[:, sel].contiguous(), Om[:, sel].contiguous()]
    return ((imgs, *rig), hm_geom, hm_terrain, controls, pose0, states_gt, pred_ts, gt_ts, nearest_steps(pred_ts, gt_ts))
