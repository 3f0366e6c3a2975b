"""The unchanged-API route at the speed of the fused step (north_star: "drops into scripts/run.py and train.py unchanged").

What a user of the reference writes (`scripts/fit_terrain.py:53-62`, `scripts/train.py:399-406`):

    states, forces = dphysics(z_grid=z, controls=u, friction=mu)
    loss = physics_loss(states_pred=states, states_gt=gt, pred_ts=ts, gt_ts=ts_gt, gamma=0.9)
    loss.backward()

is three calls with ~0.4 ms of Python, ctypes and autograd between them (2 ms on a slow host) around ~0.4 ms of kernels, and it cannot use the
fused loss: `forward` does not know a `physics_loss` follows.  After `ARM_AFTER` identical cycles -- same buffers (address, shape, strides,
dtype, requires_grad) into `forward`, `physics_loss` on exactly the states it returned with the same ground-truth / stamp tensors -- this
module captures ONE hipGraph of the whole step (rollout with force rows + `physics_loss` inside the launches + backward + reduction of the
gradient copies: what `TerrainFitProblem` replays) and, from then on,

  * `forward` replays it and hands out its outputs (fresh aliases of the graph's buffers; two alternating buffer sets, so that the states of
    the previous call stay intact while the user still holds them; a set somebody still references is never overwritten -- that call runs
    launch by launch instead);
  * `physics_loss` on those states with the same arguments returns the graph's loss, whose backward hands the graph's gradients to `z_grid`
    / `friction` (scaled by the upstream gradient) -- nothing is launched but that product;
  * anything else falls back, correctly: another loss on the states sends their gradient into `_CachedStepFn.backward`, which re-runs the
    rollout launch by launch and differentiates it (the inputs' version counters are checked like autograd checks saved tensors); a
    different `physics_loss` call runs the ordinary kernels on the handed-out states; a different `forward` call runs launch by launch.

The graph reads `z_grid`, `friction`, `controls`, the ground truth and the stamps where they lie, so in-place updates (an optimizer step)
are seen; the stamp TABLES are rebuilt only when `gt_ts` / `pred_ts` change version (then the cache re-arms).  No user-visible API;
MF_API_GRAPH=0 switches it off.  Speculation costs a backward's worth of kernels when a forward under grad is NOT followed by the loss: after
`MAX_MISSES` such forwards the cache disarms itself."""
import os
import warnings

import torch

from .capture import capture

ARM_AFTER = 3          # identical (forward, physics_loss) cycles before the step is captured
MAX_MISSES = 4         # replayed forwards in a row that no matching physics_loss followed: disarm
ENABLED = os.environ.get('MF_API_GRAPH', '1') != '0'


def _tkey(t):
    return None if t is None else (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, bool(t.requires_grad), t.device.index)


def _use_count(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


class _BufferSet:
    """One captured step: the graph and what it writes (`outs`: the six rollout outputs exactly as `forward` returns them)."""
    __slots__ = ('graph', 'loss', 'outs', 'grads', 'bufs', 'base_counts', 'consumed')


class _CachedStepFn(torch.autograd.Function):
    """forward = one hipGraph replay (rollout + fused loss + backward); outputs = the six rollout outputs.  backward is reached only when
    something OTHER than the matching physics_loss differentiates the states: it re-runs the rollout launch by launch and differentiates that."""

    @staticmethod
    def forward(ctx, cache, st, z, mu, controls):
        st.graph.replay()
        ctx.cache, ctx.args = cache, (z, mu, controls)
        ctx.versions = tuple(t._version if t is not None else None for t in (z, mu, controls))
        return tuple(o.detach() for o in st.outs)      # fresh aliases (same sizes / strides / offset) of the graph's buffers

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return None, None, None, None, None
        z, mu, controls = ctx.args
        for t, v in zip((z, mu, controls), ctx.versions):
            if t is not None and t._version != v:
                raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace operation '
                                   '(DPhysics cached step: z_grid / friction / controls changed between forward and backward)')
        mod = ctx.cache.mod
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(bool(t.requires_grad)) if t is not None else None for t in (z, mu, controls)]
            ctx.cache.bypass += 1
            try:
                states, forces = mod.dphysics(ins[0], ins[2], friction=ins[1])
            finally:
                ctx.cache.bypass -= 1
            outs = list(states) + list(forces)
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o is not None]
            wrt = [t for t in ins if t is not None and t.requires_grad]
            got = torch.autograd.grad([o for o, _ in pairs], wrt, grad_outputs=[g for _, g in pairs], allow_unused=True)
        it = iter(got)
        res = [next(it) if (t is not None and t.requires_grad) else None for t in ins]
        return None, None, res[0], res[1], res[2]


class _CachedLossFn(torch.autograd.Function):
    """The graph's loss; backward = the graph's gradients times the upstream gradient (one small launch per map)."""

    @staticmethod
    def forward(ctx, st, z, mu):
        ctx.st, ctx.has_mu = st, mu is not None
        return st.loss.detach()

    @staticmethod
    def backward(ctx, gloss):
        st = ctx.st
        if gloss is None:
            return None, None, None
        it = iter(st.grads)
        gz = next(it) * gloss if ctx.needs_input_grad[1] else None
        gmu = next(it) * gloss if (ctx.has_mu and ctx.needs_input_grad[2]) else None
        return None, gz, gmu


class ApiStepCache:
    def __init__(self, mod):
        self.mod = mod
        self.bypass = 0
        self.last_cycle, self.cycles = None, 0      # observed (forward key, loss key) and how many times in a row
        self.pending = None                         # (forward key, loss args) to capture at the next matching forward
        self.entry = None                           # dict(key_f, key_l, sets, next, spec, loss_args)
        self.misses = 0
        self.failed = False
        self.replays = 0                            # (for tests / diagnostics)

    # -- eligibility ------------------------------------------------------------------------------------------
    def _config_key(self):
        m, c = self.mod, self.mod.dphys_cfg
        return (bool(c.use_odeint), float(c.dt), float(c.traj_sim_time), float(c.grid_res), float(c.d_max), float(m.stiffness), float(m.damping),
                len(m.ts), m.contiguous_outputs, m.precise, m.return_forces, m.points_per_lane, m.snap_to_terrain, m.block, c.integration_mode)

    def forward_key(self, z_grid, controls, friction, joint_angles, state):
        """None = this call is not a candidate (then nothing is observed, nothing is replayed)."""
        if (not ENABLED or self.failed or self.bypass or joint_angles is not None or state is not None or not torch.is_grad_enabled()
                or torch.cuda.is_current_stream_capturing()):
            return None
        m = self.mod
        if m.contiguous_outputs or m.precise or not m.return_forces:
            return None
        if not (torch.is_tensor(z_grid) and z_grid.is_cuda and z_grid.dtype == torch.float32 and z_grid.dim() == 3 and torch.is_tensor(controls)
                and controls.is_cuda and controls.device == z_grid.device and controls.dtype == z_grid.dtype and not controls.requires_grad):
            return None
        if friction is not None and not (friction.is_cuda and friction.dtype == z_grid.dtype and friction.device == z_grid.device):
            return None
        if not (z_grid.requires_grad or (friction is not None and friction.requires_grad)):
            return None
        # (ONE map expanded over the batch with stride 0 gets its gradient as a stride-0 expand too: scaling that would materialise [B,H,W])
        if any(t is not None and t.shape[0] > 1 and t.stride(0) == 0 for t in (z_grid, friction)):
            return None
        # The graph reads its inputs WHERE THEY LIE: anything the launch-by-launch route would first copy -- rows that are not contiguous,
        # one shared map beside per-rollout ones (expanded for real by `_make_desc`), tensors on another device than the module's -- would
        # freeze that copy into the graph.  Such calls are not candidates.
        if not (controls.is_contiguous() and z_grid.is_contiguous() and (friction is None or friction.is_contiguous())):
            return None
        if friction is not None and friction.shape[0] != z_grid.shape[0]:
            return None
        mdev = torch.device(m.device)
        if mdev.type != 'cuda' or (mdev.index is not None and mdev.index != z_grid.device.index) or (mdev.index is None and z_grid.device.index != torch.cuda.current_device()):
            return None
        return (_tkey(z_grid), _tkey(controls), _tkey(friction), self._config_key())

    @staticmethod
    def loss_key(states_gt, pred_ts, gt_ts, gamma, rotation_loss, nearest):
        if rotation_loss or nearest is not None or not (torch.is_tensor(pred_ts) and torch.is_tensor(gt_ts)):
            return None
        X_gt = states_gt[0]
        if not (X_gt.is_cuda and gt_ts.is_cuda and pred_ts.is_cuda) or X_gt.requires_grad or gt_ts.requires_grad or pred_ts.requires_grad:
            return None
        if not (X_gt.is_contiguous() and X_gt.dtype == torch.float32 and X_gt.dim() == 3):      # (read where it lies: no converted / compacted copy)
            return None
        return (_tkey(X_gt), _tkey(gt_ts), gt_ts._version, _tkey(pred_ts), pred_ts._version, float(gamma))

    # -- observation (launch-by-launch calls) ------------------------------------------------------------------
    def tag_eager_outputs(self, Xs, key_f, inputs):
        Xs._mf_obs = (self, key_f, inputs)

    def observe_loss(self, tag, states_pred, states_gt, pred_ts, gt_ts, gamma, rotation_loss, nearest):
        _, key_f, inputs = tag
        key_l = self.loss_key(states_gt, pred_ts, gt_ts, gamma, rotation_loss, nearest)
        if key_l is None or key_f is None:
            self.last_cycle, self.cycles = None, 0
            return
        cyc = (key_f, key_l)
        self.cycles = self.cycles + 1 if cyc == self.last_cycle else 1
        self.last_cycle = cyc
        if self.cycles >= ARM_AFTER and (self.entry is None or (self.entry['key_f'], self.entry['key_l']) != cyc):
            self.pending = dict(key_f=key_f, key_l=key_l, X_gt=states_gt[0], pred_ts=pred_ts, gt_ts=gt_ts, gamma=float(gamma))

    # -- the cached route ------------------------------------------------------------------------------------------
    def try_forward(self, key_f, z_grid, controls, friction):
        """The cached step for this call, or None (run launch by launch)."""
        if self.pending is not None and self.pending['key_f'] == key_f:
            pend, self.pending = self.pending, None
            try:
                self.entry = self._capture(pend, z_grid, controls, friction)
            except Exception as e:      # a capture the runtime refuses, a shape the fused route cannot carry: stay launch by launch, for good
                warnings.warn(f'DPhysics: the cached step could not be captured ({type(e).__name__}: {str(e).splitlines()[0][:160]}); '
                              'forward / physics_loss keep running launch by launch')
                self.entry, self.failed = None, True
                torch.cuda.synchronize(z_grid.device)
                return None
        e = self.entry
        if e is None or e['key_f'] != key_f:
            return None
        if self.misses >= MAX_MISSES:      # forwards nobody followed with the loss: the speculative backward is wasted work
            self.entry, self.misses, self.last_cycle, self.cycles = None, 0, None, 0
            return None
        st = e['sets'][e['next']]
        if [_use_count(b) for b in st.bufs] != st.base_counts:      # somebody still holds this set's outputs: do not overwrite them
            return None
        if not st.consumed:
            self.misses += 1
        e['next'] ^= 1
        st.consumed = False
        outs = _CachedStepFn.apply(self, st, z_grid, friction, controls)
        outs[0]._mf_step = (self, st, outs[0]._version, z_grid, friction)
        self.replays += 1
        m = self.mod
        m.z_grid = z_grid.detach()
        m.friction = friction.detach() if friction is not None else m.dphys_cfg.friction
        m.controls = controls.detach()
        m.joint_angles = None
        return tuple(outs[:4]), tuple(outs[4:])

    def cached_loss(self, tag, states_pred, states_gt, pred_ts, gt_ts, gamma, rotation_loss, nearest):
        """The graph's loss for `physics_loss(states_pred, ...)` when this IS the call the step was captured with; else None."""
        cache, st, version, z, mu = tag
        e = self.entry
        X_pred = states_pred[0]
        if e is None or st not in e['sets'] or X_pred._version != version:
            return None
        if self.loss_key(states_gt, pred_ts, gt_ts, gamma, rotation_loss, nearest) != e['key_l']:
            # another ground truth / other stamps: the ordinary kernels take this call (its gradient reaches the states and the step is
            # re-run launch by launch in `_CachedStepFn.backward`); the cache starts observing again
            self.entry, self.last_cycle, self.cycles, self.misses = None, None, 0, 0
            return None
        st.consumed = True
        self.misses = 0
        return _CachedLossFn.apply(st, z, mu)

    # -- capture ---------------------------------------------------------------------------------------------------
    def _capture(self, pend, z_grid, controls, friction):
        m = self.mod
        dev = z_grid.device
        X_gt, gt_ts, pred_ts, gamma = pend['X_gt'], pend['gt_ts'], pend['pred_ts'], pend['gamma']
        B, T = controls.shape[0], min(int(m.dphys_cfg.traj_sim_time / m.dphys_cfg.dt), controls.shape[1])
        # what the fused loss needs, checked ONCE on the host: the same stamps for every rollout, the module's own time grid as `pred_ts`
        rows = gt_ts if gt_ts.dim() == 2 else gt_ts.unsqueeze(0)
        prow = pred_ts if pred_ts.dim() == 2 else pred_ts.unsqueeze(0)
        grid = m._time_grid(T, z_grid.dtype, dev)
        if X_gt.device != dev or rows.shape[0] not in (1, B) or prow.shape[0] not in (1, B) or prow.shape[1] != T or X_gt.shape[:2] != (B, rows.shape[1]) or X_gt.shape[-1] != 3:
            raise ValueError('shapes the cached step does not carry')
        if not (bool((rows == rows[:1]).all()) and bool((prow == prow[:1]).all()) and bool(torch.allclose(prow[0].to(grid.dtype), grid, rtol=0, atol=1e-7))):
            raise ValueError('stamps differ between the rollouts, or pred_ts is not the module\'s time grid')
        spec = m.loss_spec(rows[0].detach(), gamma=gamma, n_steps=T, dtype=z_grid.dtype)
        if not spec.fusable:
            raise ValueError('two stamps on one output row')
        zc = z_grid.detach().requires_grad_(bool(z_grid.requires_grad))
        muc = friction.detach().requires_grad_(bool(friction.requires_grad)) if friction is not None else None
        cc = controls.detach()
        Xg = X_gt.detach()
        one = torch.ones((), dtype=z_grid.dtype, device=dev)
        wrt = [t for t in (zc, muc) if t is not None and t.requires_grad]

        def step():
            loss, states, forces = m.physics_loss_rollout(zc, cc, Xg, spec, friction=muc, value_in_backward=True, want_forces=True)
            grads = torch.autograd.grad(loss, wrt, grad_outputs=one)
            return loss.detach(), states, forces, grads

        self.bypass += 1
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):
                    step()
            torch.cuda.current_stream(dev).wait_stream(s)
            sets = []
            for _ in range(2):
                g = torch.cuda.CUDAGraph()
                with capture(g, stream=s, capture_error_mode='thread_local'):
                    loss, states, forces, grads = step()
                st = _BufferSet()
                st.graph, st.loss = g, loss
                st.outs = tuple(o.detach() for o in tuple(states) + tuple(forces))
                st.grads = tuple(grads)
                st.consumed = True
                sets.append(st)
                del loss, states, forces, grads
        finally:
            self.bypass -= 1
        for st in sets:
            st.bufs = list(st.outs)
            st.base_counts = [_use_count(b) for b in st.bufs]
        return dict(key_f=pend['key_f'], key_l=pend['key_l'], sets=sets, next=0, spec=spec, keep=(Xg, gt_ts, pred_ts, cc, zc, muc, one))      # (everything the graph reads by address lives as long as the graph: the seed too)
