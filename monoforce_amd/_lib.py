"""ctypes binding of libmonoforce_hip.so (the C ABI declared in include/monoforce_hip.h).

There is no CPU fallback: if the library has not been built (`python -c "import __graft_entry__ as g; g.build()"`
or `make -C monoforce_amd/csrc`) every entry point raises.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmonoforce_hip.so')

MF_INTEG_DYNAMICS, MF_INTEG_ODEINT_EULER = 0, 1
MF_LAYOUT_BATCH_MAJOR, MF_LAYOUT_TIME_MAJOR = 0, 1
MF_LOSS_VALUE_IN_BACKWARD = 1
MF_MATH_EXACT, MF_MATH_FAST = 0, 1
MF_LANES_COMPONENT = 16


class MfRolloutDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('n_tracks', C.c_int32), ('integrator', C.c_int32), ('layout', C.c_int32), ('map_shared', C.c_int32),
                ('block', C.c_int32), ('skip_snap', C.c_int32), ('points_per_lane', C.c_int32), ('math_mode', C.c_int32), ('force_stride', C.c_int32), ('grad_copies', C.c_int32), ('has_joints', C.c_int32), ('pose_stride', C.c_int32), ('cost_project', C.c_int32), ('default_state', C.c_int32), ('controls_stride_b', C.c_int32), ('controls_stride_t', C.c_int32),
                ('mass', C.c_double), ('gravity', C.c_double), ('stiffness', C.c_double), ('damping', C.c_double),
                ('omega_max', C.c_double), ('grid_res', C.c_double), ('d_max', C.c_double), ('dt', C.c_double),
                ('robot_size_y', C.c_double), ('Iinv', C.c_double * 9), ('joint_xyz', C.c_double * 12)]


class MfRolloutLoss(C.Structure):
    _fields_ = [('T2', C.c_int32), ('flags', C.c_int32)] + [(n, C.c_void_p) for n in ('gt', 'near', 'w', 'row_stamp', 'row_w', 'partial', 'ticket', 'loss', 'gloss', 'Xs')]


class MfRolloutFwdBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('z', 'mu', 'controls', 'ts', 'points', 'part', 'x0', 'xd0', 'R0', 'w0',
                                          'Xs', 'Xds', 'Rs', 'Omegas', 'Fs', 'Ff', 'Xraw', 'joint_angles', 'cost_rows', 'path_cost',
                                          'zmu_scratch', 'zmu', 'rec', 'loss')]


class MfRolloutBwdBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('z', 'mu', 'controls', 'ts', 'points', 'part', 'x_init', 'xd0', 'R0', 'w0',
                                          'Xraw', 'Xds', 'Rs', 'Omegas',
                                          'gXs', 'gXds', 'gRs', 'gOmegas', 'gFs', 'gFf', 'zeros',
                                          'gz', 'gmu', 'gcontrols', 'gx0', 'gxd0', 'gR0', 'gw0', 'joint_angles', 'gjoint_angles', 'rec', 'loss',
                                          'zmu_scratch', 'zmu')]


class MfLossDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('T1', C.c_int32), ('T2', C.c_int32), ('reserved', C.c_int32),
                ('x_stride_b', C.c_int64), ('x_stride_t', C.c_int64), ('gamma', C.c_double)]


class MfSplatDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('n_per_sample', C.c_int32), ('C', C.c_int32),
                ('nx', C.c_int32), ('ny', C.c_int32), ('nz', C.c_int32),
                ('off', C.c_float * 3), ('dx', C.c_float * 3), ('lift_D', C.c_int32), ('lift_hw', C.c_int32)]


class MfStageDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('k', C.c_int32)]


class MfInterpDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('map_shared', C.c_int32), ('math_mode', C.c_int32),
                ('grid_res', C.c_double), ('d_max', C.c_double)]


class MfHeightmapDesc(C.Structure):
    _fields_ = [('n_points', C.c_int32), ('nx', C.c_int32), ('ny', C.c_int32), ('d_max', C.c_float), ('h_min', C.c_float),
                ('h_max', C.c_float), ('r_min', C.c_float), ('inv_res', C.c_float)]


# every symbol include/monoforce_hip.h declares; tests check the library exports all of them
SYMBOLS = ['mf_rollout_force_stride', 'mf_rollout_fwd_f32', 'mf_rollout_fwd_f64', 'mf_rollout_default_state_f32', 'mf_rollout_default_state_f64', 'mf_rollout_bwd_f32', 'mf_rollout_bwd_f64', 'mf_rollout_bwd_wants_gcontrols', 'mf_rollout_record_bytes', 'mf_rollout_record_bytes_f64', 'mf_rollout_fwd_stages_zmu', 'mf_rollout_loss_fusable', 'mf_rollout_bwd_window', 'mf_bev_splat_workspace_bytes', 'mf_bev_splat_prepare', 'mf_bev_splat_prepare_cameras', 'mf_bev_splat_prepare_rig',
           'mf_bev_splat_fwd_f32', 'mf_bev_splat_fwd_f64', 'mf_bev_splat_bwd_f32', 'mf_bev_splat_bwd_f64',
           'mf_bev_lift_splat_fwd_f32', 'mf_bev_lift_splat_fwd_f64', 'mf_bev_lift_splat_bwd_f32', 'mf_bev_lift_splat_bwd_f64',
           'mf_physics_loss_fwd_f32', 'mf_physics_loss_fwd_f64', 'mf_physics_loss_bwd_f32', 'mf_physics_loss_bwd_f64', 'mf_physics_loss_value_f32', 'mf_physics_loss_value_f64', 'mf_nearest_steps_f32', 'mf_nearest_steps_f64', 'mf_reduce_grad_copies_f32', 'mf_reduce_grad_copies_f64', 'mf_estimate_heightmap_f32', 'mf_interpolate_grid_f32', 'mf_interpolate_grid_f64', 'mf_terrain_stage_fwd_f32', 'mf_terrain_stage_bwd_f32', 'mf_last_error', 'mf_last_launch', 'mf_version', 'mf_sizeof']

_lib = None
_lock = threading.Lock()


def lib():
    """Load (once) and return the HIP library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f'{LIB_PATH} is missing: build the HIP extension first (make -C monoforce_amd/csrc). '
                        'monoforce_amd has no CPU fallback.')
                # torch bundles its own libamdhip64.so.7; import it FIRST so this library binds to the same HIP runtime
                # instance (loading /opt/rocm's copy first leaves the process with two runtimes and no device).
                import torch  # noqa: F401
                L = C.CDLL(os.environ.get('MONOFORCE_HIP_LIB', LIB_PATH))   # override: A/B builds of the kernels (tools/)
                L.mf_last_error.restype = C.c_char_p
                L.mf_version.restype = C.c_char_p
                L.mf_last_launch.restype = C.c_char_p
                for name in SYMBOLS:
                    fn = getattr(L, name)   # AttributeError if the build is stale
                    if name.startswith(('mf_rollout', 'mf_bev', 'mf_physics', 'mf_estimate', 'mf_terrain', 'mf_reduce', 'mf_interpolate', 'mf_nearest')):
                        fn.restype = C.c_int
                L.mf_bev_splat_workspace_bytes.restype = C.c_size_t
                L.mf_rollout_record_bytes.restype = C.c_longlong
                L.mf_rollout_record_bytes_f64.restype = C.c_longlong
                _lib = L
    return _lib


ON_ERROR = []      # callables run when an entry point reports an error (state that assumed the launch completed is dropped)


def check(rc, what):
    if rc != 0:
        for fn in list(ON_ERROR):
            fn()
        raise RuntimeError(f'{what} failed (code {rc}): {lib().mf_last_error().decode()}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def require_hip_tensor(t, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name} is on {t.device}: monoforce_amd computes on the MI355X HIP path only '
                           '(no CPU fallback); move inputs to "cuda".')
