"""No-GPU checks of the C ABI: the library builds, loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from tests.conftest import REPO


@pytest.fixture(scope='module')
def built_lib():
    import __graft_entry__ as g
    g.build()
    from monoforce_amd import _lib
    return _lib.lib()


def header_functions():
    src = open(os.path.join(REPO, 'include', 'monoforce_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return re.findall(r'\b(mf_[a-z0-9_]+)\s*\(', src)


def test_library_exports_every_declared_symbol(built_lib):
    from monoforce_amd import _lib
    names = header_functions()
    assert 'mf_rollout_fwd_f32' in names and 'mf_last_error' in names
    for n in names:
        assert hasattr(built_lib, n), f'{n} declared in include/monoforce_hip.h but not exported'
    assert set(names) == set(_lib.SYMBOLS), 'monoforce_amd._lib.SYMBOLS is out of sync with the header'
    assert built_lib.mf_version().decode().startswith('monoforce_hip')


def test_struct_layout_matches_header(built_lib):
    """ctypes mirrors must have the C struct sizes (checked against sizes the library reports)."""
    from monoforce_amd import _lib
    built_lib.mf_sizeof.restype = ctypes.c_int
    built_lib.mf_sizeof.argtypes = [ctypes.c_char_p]
    for name in ('MfRolloutDesc', 'MfRolloutFwdBufs', 'MfRolloutBwdBufs', 'MfSplatDesc', 'MfLossDesc', 'MfHeightmapDesc', 'MfStageDesc', 'MfInterpDesc', 'MfRolloutLoss'):
        assert built_lib.mf_sizeof(name.encode()) == ctypes.sizeof(getattr(_lib, name)), name


def test_argument_validation_without_gpu(built_lib):
    """Descriptor validation happens before any launch, so it can be exercised on a CPU-only box."""
    from monoforce_amd import _lib
    d = _lib.MfRolloutDesc(B=0, T=1, N=4, H=8, W=8, n_tracks=2)
    b = _lib.MfRolloutFwdBufs()
    rc = built_lib.mf_rollout_fwd_f32(ctypes.byref(d), ctypes.byref(b), None)
    assert rc == 1 and b'positive' in built_lib.mf_last_error()
    d = _lib.MfRolloutDesc(B=1, T=1, N=4, H=8, W=8, n_tracks=3)
    rc = built_lib.mf_rollout_fwd_f32(ctypes.byref(d), ctypes.byref(b), None)
    assert rc == 1 and b'n_tracks' in built_lib.mf_last_error()
    d = _lib.MfRolloutDesc(B=1, T=1, N=4, H=8, W=8, n_tracks=2)
    rc = built_lib.mf_rollout_fwd_f64(ctypes.byref(d), ctypes.byref(b), None)
    assert rc == 1 and b'null input' in built_lib.mf_last_error()


def test_force_stride_query(built_lib):
    from monoforce_amd import _lib
    q = lambda **kw: built_lib.mf_rollout_force_stride(ctypes.byref(_lib.MfRolloutDesc(**kw)))  # noqa: E731
    assert q(B=1024, N=4) == 4 and q(B=10 ** 6, N=4) == 4
    assert q(B=8, N=223) == 256 and q(B=8, N=33) == 64 and q(B=8, N=33, points_per_lane=4) == 64
    assert q(B=8, N=300) == 512 and q(B=0, N=4) == -1 and q(B=1, N=513) == -1


def test_record_bytes_query(built_lib):
    """`mf_rollout_record_bytes` is a host-side policy: 256 B per rollout and step where both directions run component-parallel
    (dynamics(): only while its backward streams it, 256 waves) with at most one wave per SIMD (1024 waves = 4096 rollouts); 0 everywhere else (the caller then passes
    rec = NULL).  VERDICT r2 item 1: <= 131 MB at the BASELINE shape (round 2: 524 MB, and only up to 1024 rollouts)."""
    import ctypes as C
    from monoforce_amd import _lib
    built_lib.mf_rollout_record_bytes.restype = C.c_longlong

    def q(**kw):
        d = dict(B=1024, T=500, N=4, H=256, W=256, integrator=1, math_mode=_lib.MF_MATH_FAST, force_stride=4, map_shared=1,
                 layout=_lib.MF_LAYOUT_TIME_MAJOR)
        d.update(kw)
        return int(built_lib.mf_rollout_record_bytes(C.byref(_lib.MfRolloutDesc(**d))))
    assert q() == 1024 * 500 * 256 <= 131.1e6
    assert q(B=1, T=200) == 200 * 256
    assert q(B=3, T=7, N=2) == 3 * 7 * 256                   # per-lane slabs: absent contact points included
    assert q(B=2048) == 2048 * 500 * 256 and q(B=4096) == 4096 * 500 * 256
    assert q(B=4097) == 0 and q(B=8192) == 0                 # more than one wave per SIMD: the recomputing kernels
    assert q(integrator=0) == q() and q(integrator=0, B=2048) == 0      # dynamics(): while its backward streams the record (one workgroup per CU)
    assert q(math_mode=_lib.MF_MATH_EXACT) == 0 and q(has_joints=1) == 0
    # bodies of 5..512 points, one point per lane, either integrator, up to two waves per SIMD: the 16-byte record of
    # rollout_bwd_mw_kernel.h (contact count + unclamped angular acceleration per rollout-step)
    assert q(N=5, force_stride=8) == 1024 * 500 * 16 and q(N=32, force_stride=32) == 1024 * 500 * 16 and q(B=64, T=600, N=223, force_stride=256) == 64 * 600 * 16
    assert q(N=32, force_stride=32, B=2048) == 2048 * 500 * 16 and q(N=32, force_stride=32, B=8192) == 0 and q(N=32, force_stride=32, integrator=0) == 1024 * 500 * 16 and q(N=223, force_stride=256, B=4096) == 0
    assert q(points_per_lane=1) == 0                         # an explicit other lane mapping
    assert int(built_lib.mf_rollout_record_bytes(None)) == 0


def test_loss_fusable_query(built_lib):
    """`mf_rollout_loss_fusable` over the batch range of a 4-point body (a host-side policy, answered without a device): 1 = both directions on
    the streaming component-parallel kernels; 0 = the record-reading and late-recompute one-wave forms (the fused kernel gains nothing there:
    profiles/r6_ab_one_wave_loss.txt); 3 = the early-recompute form; 2 = the saturated one-point-per-lane kernels."""
    import ctypes as C
    from monoforce_amd import _lib

    def q(**kw):
        d = dict(B=1024, T=500, N=4, H=256, W=256, integrator=1, math_mode=_lib.MF_MATH_FAST, force_stride=4, map_shared=1,
                 layout=_lib.MF_LAYOUT_TIME_MAJOR)
        d.update(kw)
        return int(built_lib.mf_rollout_loss_fusable(C.byref(_lib.MfRolloutDesc(**d))))
    assert [q(B=b) for b in (1, 1024, 2048)] == [1, 1, 1]
    assert [q(B=b) for b in (2049, 2304, 4096)] == [0, 0, 0]
    assert [q(B=b) for b in (4097, 6144, 8192)] == [3, 3, 3]
    assert [q(B=b) for b in (8193, 16384, 65536)] == [2, 2, 2]
    assert [q(B=b, integrator=0) for b in (1024, 1025, 4096, 4097, 8192, 16384)] == [1, 0, 0, 3, 3, 2]
    assert q(has_joints=1) == 0 and q(B=16384, has_joints=1) == 0 and q(math_mode=_lib.MF_MATH_EXACT) == 0
    assert int(built_lib.mf_rollout_loss_fusable(None)) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from monoforce_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_control_tensor_classification_on_the_host():
    """Host logic of the controls hand-over: a [B,1,2] sample expanded over time is recognised (and read in place on the GPU);
    everything else -- and every CPU tensor -- is handed over as a contiguous [B,T,2] copy."""
    import torch
    from monoforce_amd.dphys_config import DPhysConfig
    from monoforce_amd.dphysics import _kernel_controls, _time_constant
    from monoforce_amd.planner import sample_controls
    from monoforce_amd import synthetic as syn
    pts, masks = syn.robot_points_4()
    cfg = DPhysConfig(robot='tradr', grid_res=0.1, robot_points=pts, driving_parts=masks)
    c = sample_controls(6, cfg, 'cpu', torch.Generator().manual_seed(0))
    T = int(cfg.traj_sim_time / cfg.dt)
    assert c.shape == (6, T, 2) and c.stride(1) == 0 and _time_constant(c)
    assert float(c[:3, 0, 0].min()) > 0 > float(c[3:, 0, 0].max())            # first half forward, second half backward
    assert torch.equal(c[:, 0], c[:, T - 1])
    dense = c.contiguous()
    assert not _time_constant(dense) and not _time_constant(c[:, :1]) and not _time_constant(dense.transpose(0, 1))
    t, sb, st = _kernel_controls(c, allow_view=True)                          # CPU tensor: never handed over as a view
    assert t.is_contiguous() and (sb, st) == (0, 0) and torch.equal(t, dense)
    t, sb, st = _kernel_controls(c, allow_view=False)
    assert t.is_contiguous() and (sb, st) == (0, 0)


def test_forward_reports_where_it_stages_the_interleaved_maps(built_lib):
    """mf_rollout_fwd_stages_zmu: the component-parallel forward fills `zmu_scratch` from half a wave per SIMD up, for a SHARED float32 map
    pair of a rigid body of <= 4 points -- the buffer the record-reading backward may then be handed as `zmu` (no GPU needed: the query
    uses the MI355X's geometry when no device is present)."""
    from monoforce_amd import _lib
    def stages(**kw):  # noqa: E306
        base = dict(B=4096, T=500, N=4, H=256, W=256, n_tracks=2, integrator=1, math_mode=_lib.MF_MATH_FAST, force_stride=4, map_shared=1,
                    layout=_lib.MF_LAYOUT_TIME_MAJOR)
        base.update(kw)
        return built_lib.mf_rollout_fwd_stages_zmu(ctypes.byref(_lib.MfRolloutDesc(**base)))
    assert stages() == 1 and stages(B=2048) == 1
    assert stages(B=1024) == 0                      # below half a wave per SIMD the pass costs what it brings
    assert stages(map_shared=0) == 0 and stages(math_mode=0) == 0 and stages(N=32, force_stride=32) == 0 and stages(has_joints=1) == 0
    assert stages(B=100000) == 0                    # beyond the component-parallel kernels' range
    assert built_lib.mf_rollout_fwd_stages_zmu(None) == 0


def test_last_error_and_last_launch_are_thread_local(built_lib):
    """VERDICT r4 / housekeeping: two host threads (two streams) each read the message of THEIR failed call; a thread that never
    failed reads "".  mf_last_launch: "" for a thread that launched nothing."""
    import threading
    from monoforce_amd import _lib
    seen = {}

    def worker(name, n_tracks, barrier):
        d = _lib.MfRolloutDesc(B=1 if n_tracks else 0, T=1, N=4, H=8, W=8, n_tracks=n_tracks or 2)
        b = _lib.MfRolloutFwdBufs()
        rc = built_lib.mf_rollout_fwd_f32(ctypes.byref(d), ctypes.byref(b), None)
        barrier.wait()                      # both calls have failed before either thread reads its text
        seen[name] = (rc, built_lib.mf_last_error().decode(), built_lib.mf_last_launch().decode())

    bar = threading.Barrier(2)
    ts = [threading.Thread(target=worker, args=('a', 3, bar)), threading.Thread(target=worker, args=('b', 0, bar))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen['a'][0] == 1 and 'n_tracks' in seen['a'][1] and 'positive' not in seen['a'][1]
    assert seen['b'][0] == 1 and 'positive' in seen['b'][1] and 'n_tracks' not in seen['b'][1]
    assert seen['a'][2] == '' and seen['b'][2] == ''
    quiet = {}
    t = threading.Thread(target=lambda: quiet.setdefault('e', built_lib.mf_last_error().decode()))
    t.start(); t.join()
    assert quiet['e'] == ''
