/*
 * monoforce_hip.h -- C ABI of libmonoforce_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the MonoForce hot path.  Every entry point takes plain device pointers, sizes and a
 * HIP stream (as void*); nothing here knows about torch.  All launches are asynchronous on the given stream; no
 * entry point synchronises or allocates.  Return value: MF_OK or an MF_ERR_* code (text via mf_last_error()).
 *
 * Reference interfaces replaced (paths under /root/reference/monoforce/src/monoforce/models/):
 *   mf_rollout_fwd_*   traj_predictor/dphysics.py:530-594  DPhysics.dphysics()  = forward_kinematics (:172-272)
 *                      + dynamics (:467-497) / dynamics_odeint (:499-528) + interpolate_grid (:385-455)
 *                      + update_joints (:326-358); in path-cost mode also the reductions the planning nodes apply to its
 *                      outputs (monoforce_ros/nodes/monoforce_node.py:91, diff_physics.py:263-266)
 *   mf_rollout_bwd_*   the autograd graph the reference builds through the same functions (loss.backward(),
 *                      scripts/fit_terrain.py:53-62, scripts/train.py:399-406)
 *   mf_bev_splat_*     terrain_encoder/lss.py:238-280 LiftSplatShoot.voxel_pooling() + terrain_encoder/utils.py:144-181
 *                      (cumsum_trick / QuickCumsum forward and backward)
 *   mf_bev_lift_splat_*  the same with the lift of lss.py:63-71 (depth distribution x context features) fused in
 *   mf_physics_loss_*  losses.py:102-127 physics_loss (position term) and its gradient, on the nearest-time-stamp
 *                      subset of the predicted poses
 *   mf_bev_splat_prepare_cameras  the voxel plan straight from the camera models: LiftSplatShoot.get_geometry()
 *                      (lss.py:204-224) evaluated inside the key pass
 */
#ifndef MONOFORCE_HIP_H
#define MONOFORCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MF_OK = 0,
  MF_ERR_INVALID = 1,     /* bad argument (shape / null pointer / enum) */
  MF_ERR_UNSUPPORTED = 2, /* valid request outside what the kernels cover (e.g. > 512 contact points) */
  MF_ERR_LAUNCH = 3       /* HIP reported an error at launch */
};

/* Integrators of DPhysics (dphys_config.py:150-153). */
enum {
  MF_INTEG_DYNAMICS = 0,    /* use_odeint=False: dynamics() -- semi-implicit Euler + Rodrigues, records state AFTER each step */
  MF_INTEG_ODEINT_EULER = 1 /* use_odeint=True (reference default): torchdiffeq fixed-grid explicit Euler on the
                               extended state; output 0 = initial state; "forces" are running impulses */
};

/* Output layout.  The reference's default path returns permuted views of time-major tensors (dphysics.py:515-526),
 * its dynamics() path returns contiguous [B,T,...] stacks (dphysics.py:490-495).  Both are supported; time-major lets a
 * wavefront's per-step stores land in one contiguous segment. */
enum {
  MF_LAYOUT_BATCH_MAJOR = 0, /* X[b][t][...] */
  MF_LAYOUT_TIME_MAJOR = 1   /* X[t][b][...] */
};

/* Arithmetic of the float32 kernels (float64 is always exact). */
enum {
  MF_MATH_EXACT = 0, /* IEEE divide/sqrt, libm exp/sincos, no FMA contraction: the reference's eager op sequence, bit-for-bit
                        for the first ~100 steps of a rollout */
  MF_MATH_FAST = 1   /* hardware rcp/rsq/exp2 (1 ulp) + FMA: ~1.5x faster per step; same parity tolerances hold */
};

/* MfRolloutDesc.points_per_lane, besides 0 / 1 / 4: the component-parallel mapping -- a rollout over a 16-lane row, four lanes
 * per contact point (lane = vector component / footprint cell).  float32 MF_MATH_FAST rigid bodies of <= 4 points, full or
 * states-only outputs, BOTH integrators (forward and backward); with points_per_lane = 0 it is chosen by itself for small
 * launches (forward: up to 1024 waves = B <= 4096, backward: up to 2048 waves = B <= 8192), where a step costs ~2x fewer
 * instructions per wave than one point per lane.  Other configurations fall back to the automatic choice. */
enum { MF_LANES_COMPONENT = 16 };

/* Shapes and physical constants of one rollout launch.  Scalars are double here and are rounded ONCE to the
 * kernel's arithmetic type, like the Python floats of DPhysConfig are when they meet a tensor. */
typedef struct MfRolloutDesc {
  int32_t B;          /* rollouts */
  int32_t T;          /* grid points N_ts = outputs per rollout (dphysics.py:573) */
  int32_t N;          /* contact points (cfg.robot_points.shape[0]) */
  int32_t H, W;       /* grid size; flat index = iy + H*ix, first grid axis = x (dphysics.py:427-430) */
  int32_t n_tracks;   /* len(cfg.driving_parts): 2 or 4 (dphysics.py:75-104) */
  int32_t integrator; /* MF_INTEG_* */
  int32_t layout;     /* MF_LAYOUT_* of all outputs */
  int32_t map_shared; /* 1: z/mu are one [H,W] map shared by all rollouts; 0: [B,H,W] */
  int32_t block;      /* threads per workgroup: 0 = default (64); otherwise 64, 128 or 256.  Ignored where one rollout
                         spans a whole workgroup (bodies of 65..512 points at small batch sizes: 128, 256 or 512 threads) */
  int32_t skip_snap;  /* 1: do NOT move x0.z onto the terrain first (dphysics.py:567-571) -- for continuing a rollout
                         from a mid-trajectory state (chunked horizons, teacher-forced single steps) */
  int32_t points_per_lane; /* lane mapping: 0 = choose from B and N; 1 = one contact point per lane (fewest instructions per
                         wave, best when the launch is latency-bound); 4 = four points per lane (least redundant work,
                         best when the chip is full).  Results differ only in float summation order.  With 0 or 1, bodies of
                         65..512 points are spread over 2, 4 or 8 waves per rollout while the launch has <= 2048 waves. */
  int32_t math_mode;    /* MF_MATH_* */
  int32_t force_stride; /* point slots per row of the Fs / Ff buffers, >= mf_rollout_force_stride(desc); 0 means N
                           (only valid when N is a multiple of the lane tile, e.g. N = 4).  Padding slots get zeros. */
  int32_t grad_copies;  /* backward, shared map only: number of private copies of the gz / gmu maps (rollout b scatters into
                           copy b % grad_copies -- the LDS-window kernels: a workgroup's window into copy workgroup % grad_copies; the caller sums the copies).  Thousands of rollouts of one batch cross the
                           same cells, and same-address float atomics serialise in L2 at ~20 ns each; 0 or 1 = one copy.
                           Measured: 32 copies for bodies of up to 4 points (~64 rollouts per copy beyond 2048 rollouts), at least
                           64 for larger bodies, whose points sit on nearly the same cells in every rollout of a batch
                           (1024 rollouts x 32 points: backward 0.98 ms at 16 copies, 0.81 ms at 64). */
  int32_t has_joints;   /* 1: MfRolloutFwdBufs.joint_angles will be given (selects the articulated kernels / force stride) */
  int32_t pose_stride;  /* path-cost mode (MfRolloutFwdBufs.cost_rows): Xs / Rs keep every pose_stride-th output row; else 0 */
  int32_t cost_project; /* path-cost mode, MF_INTEG_ODEINT_EULER: 1 = the cost rows carry the third row of the NEAREST ROTATION to
                           the (drifting) R -- what scipy's Rotation.from_matrix(R).as_euler() reads roll / pitch from
                           (diff_physics.py:263-266); 0 = the raw third row (enough for the force cost; ~20 % faster) */
  int32_t default_state; /* forward only: 1 = start from the reference's default state (dphysics.py:554-559: x = 0,
                           xd = (v_0, 0, 0), R = I, omega = (0, 0, w_0) with (v_0, w_0) = controls[b][0]); the kernel computes it
                           and WRITES x0 / xd0 / R0 / w0 (for the caller and the backward pass) instead of reading them --
                           saves the separate mf_rollout_default_state_* launch */
  int32_t controls_stride_b, controls_stride_t; /* forward only: element strides of `controls` over rollouts and over time
                           ((v, w) stay adjacent); both 0 = contiguous [B][T][2].  controls_stride_t = 0 reads ONE (v, w) per
                           rollout for the whole horizon -- the constant-in-time samples of trajectory shooting
                           (generate_controls, dphysics.py:42-72; monoforce_node.py:41-52) as the [B,1,2] tensor they are,
                           instead of a materialised [B,T,2] copy that every step of every rollout would fetch from HBM */
  double mass, gravity, stiffness, damping, omega_max;
  double grid_res, d_max;
  double dt;           /* cfg.dt: step of MF_INTEG_DYNAMICS (ODEINT takes its steps from ts[]) */
  double robot_size_y; /* Ly = cfg.robot_size[1] */
  double Iinv[9];      /* inverse body inertia, row-major (dphysics.py:152-153) */
  double joint_xyz[12]; /* joint positions of the 4 driving parts [fl, fr, rl, rr][xyz] (cfg.joint_positions); has_joints only */
} MfRolloutDesc;

/* physics_loss (losses.py:102-127: time-discounted position MSE at the output rows nearest to the ground-truth stamps) INSIDE the
 * rollout kernels: the forward accumulates  mean_{b,j,c} (Xs[b, near[j], c] w_j - gt[b, j, c] w_j)^2  at the
 * <= T2 stamped rows while it writes them, and the backward synthesises dL/dXs at those rows from Xs and gt instead of reading a
 * [T][B][3] gradient tensor -- a train step loses two launches and 6 MB of rows.  Float32, component-parallel kernels with the
 * streaming backward only: ask mf_rollout_loss_fusable(desc).  The ground-truth stamps must be the SAME for every rollout
 * (near / w / row_stamp are per stamp, not per rollout) and `near` strictly increasing. */
/* MfRolloutLoss.flags.  MF_LOSS_VALUE_IN_BACKWARD: a caller that ALWAYS runs the backward after the forward (a train step) lets the
 * backward form the loss VALUE as well -- its fetching waves hold Xs, the ground truth and the weight of every stamped row anyway and
 * have issue slots to spare, the forward (one wave per SIMD, issue-bound) has none: the forward then only marks loss[0] as not yet
 * known (NaN; needs `loss` alone), the backward fills it (needs partial, ticket, loss besides gloss / Xs / the tables).  One launch
 * and ~10 us less per step than forming the value in a launch of its own. */
#define MF_LOSS_VALUE_IN_BACKWARD 1
typedef struct MfRolloutLoss {
  int32_t T2;               /* ground-truth stamps per rollout */
  int32_t flags;            /* 0, or MF_LOSS_VALUE_IN_BACKWARD */
  const void* gt;           /* S[B][T2][3] ground-truth positions */
  const int32_t* near;      /* int32[T2]: output row nearest in time to stamp j (losses.py:116), strictly increasing, < T */
  const void* w;            /* S[T2]: time weights 1 / (1 + gamma t_j) (losses.py:122) */
  const int32_t* row_stamp; /* int32[T]: stamp index j of output row t, -1 where the row carries none (the inverse of `near`) */
  const void* row_w;        /* S[T]: w[row_stamp[t]], 0 where the row carries no stamp (the kernels read the stamps through the two
                               row tables; near / w document them and serve mf_physics_loss_* callers) */
  void* partial;            /* scratch of the direction that forms the value: S[ceil(B / 4)] per-workgroup partial sums */
  uint32_t* ticket;         /* the same direction: ONE zero-initialised counter; the launch leaves it zero again */
  void* loss;               /* out: S[1], the mean over B x T2 x 3 (written by the forward, or by the backward: flags) */
  const void* gloss;        /* backward: S[1] upstream gradient of the loss (device scalar) */
  const void* Xs;           /* backward: the forward's Xs rows (shifted positions, layout of the launch) */
} MfRolloutLoss;

/* Device buffers of the forward rollout; S = float for _f32, double for _f64.  All contiguous. */
typedef struct MfRolloutFwdBufs {
  const void* z;        /* S[map_shared ? 1 : B][H][W] height map */
  const void* mu;       /* same shape friction map, or NULL = all ones (cfg.friction, dphysics.py:562) */
  const void* controls; /* S[B][T][2] = (v, w) */
  const void* ts;       /* S[T] time grid (dphysics.py:167,581); only its differences are used (ODEINT) */
  const void* points;   /* S[N][3] body-frame contact points */
  const int32_t* part;  /* int32[N]: index of the LAST driving mask holding the point, -1 = not driving (:242-246) */
  void* x0;             /* S[B][3] IN/OUT: z component is overwritten with the terrain height under the robot (:567-571) */
  const void* xd0;      /* S[B][3] */
  const void* R0;       /* S[B][3][3] */
  const void* w0;       /* S[B][3] */
  void* Xs;             /* S[..][3]   positions, already shifted by R[:,2]*m*g/(k+1e-6) (:587-589) */
  void* Xds;            /* S[..][3]   */
  void* Rs;             /* S[..][3][3]*/
  void* Omegas;         /* S[..][3]   */
  void* Fs;             /* S[..][force_stride][3] spring forces (DYNAMICS) or their running impulses (ODEINT).  Fs and Ff
                           may both be NULL for float32 MF_MATH_FAST rigid-body rollouts: states-only kernels that skip
                           24 N of the 80 + 56 N bytes per step (training consumes only the states, scripts/train.py:243) */
  void* Ff;             /* S[..][force_stride][3] friction forces / impulses */
  void* Xraw;           /* optional S[..][3]: unshifted positions saved for the backward pass; may be NULL */
  const void* joint_angles; /* optional S[B][T][4] flipper angles: each driving part is rotated about the y-axis through
                           its joint and the body inertia recomputed EVERY step (update_joints, dphysics.py:192-197,
                           326-358; the reference does this for robot == 'marv' and non-zero angles).  All six
                           outputs must be given.  NULL = rigid body. */
  void* cost_rows;      /* optional S[T][B][4], float32 MF_MATH_FAST + MF_LAYOUT_TIME_MAJOR rigid-body rollouts: PATH-COST mode
                           for trajectory shooting.  Per output row the kernel writes (R[2][0], R[2][1], R[2][2], s) (see
                           desc->cost_project) with
                           s = unbiased std over the N contact points of |F_spring row| -- the inputs of the reference's
                           path costs (monoforce_node.py:91 `norm(F_springs).std(points).std(time)`; diff_physics.py:263-266
                           roll / pitch) -- instead of the full rows: Xds, Omegas, Fs, Ff, Xraw must be NULL, and Xs / Rs are
                           DECIMATED to S[1 + ceil((T-1) / pose_stride)][B][3 | 3x3]: pose row r holds output row
                           min(r * pose_stride, T - 1) (what the nodes publish: poses[::pose_step], plus the final pose). */
  void* path_cost;      /* optional S[B], with cost_rows: the force path cost itself, std over the T rows of s (unbiased, Welford
                           in registers) = norm(F_springs, dim=-1).std(dim=-1).std(dim=-1) of monoforce_node.py:91 */
  void* zmu_scratch;    /* optional scratch, 2*H*W floats (8-byte aligned): with a SHARED map (desc->map_shared), float32
                           MF_MATH_FAST, a rigid body of <= 64 contact points and >= 512 waves of rollouts the entry point first interleaves z and mu
                           into it, cell by cell, and the rollout kernel reads a point's footprint in both maps with two
                           16-byte loads instead of eight 4-byte ones (the L1 looks up one line per lane and cycle: at
                           >= 16 k rollouts that bounds the kernel).  Same results, bit for bit.  Contents are undefined
                           afterwards; NULL (or any other configuration) = the maps are read where they are. */
  const void* zmu;      /* optional, float32: the SHARED maps already interleaved, S[H][W][2] = (z, mu) per cell, holding the same
                           values as z / mu (mf_terrain_stage_fwd_f32 writes all three in one pass).  Kernels that gather from an
                           interleaved pair (float32 MF_MATH_FAST, rigid body of <= 64 points, one point per lane or component-
                           parallel) read it whatever the batch size and skip their own interleave pass; the others read z / mu. */
  void* rec;            /* optional, float32: mf_rollout_record_bytes(desc) bytes (16-byte aligned) that receive the forward's per-step
                           record for a backward that then reads it instead of recomputing (autograd saves every intermediate of
                           dphysics.py:172-272).  Two forms, chosen by the launch shape:
                           component-parallel kernels (rigid body of <= 4 points): [T][B * 16 lanes] quads (u, c, omega_d raw, A) --
                           cell coordinate, contact weight, unclamped angular acceleration component, A = k dh + d v_n -- 256 B per
                           rollout-step;
                           one point per lane, bodies of 5..512 points, either integrator, up to two waves per SIMD: [T][B] quads
                           (sum of the contact weights, omega_d before its clamp) -- 16 B per rollout-step, what the record-reading
                           backward (rollout_bwd_mw_kernel.h) cannot rebuild without a reduction over the contact points.
                           NULL, or a launch the record does not apply to: nothing is written and the backward recomputes. */
  const MfRolloutLoss* loss; /* optional: fuse physics_loss into the launch (mf_rollout_loss_fusable(desc) must be 1; Fs = Ff = NULL,
                           MF_LAYOUT_TIME_MAJOR): fills loss->loss.  NULL = plain rollout. */
} MfRolloutFwdBufs;

/* Can this launch carry the fused physics loss (MfRolloutLoss)?
 *   1  BOTH directions, on the component-parallel kernels with the streaming backward: float32 MF_MATH_FAST, rigid body of <= 4 points,
 *      time-major outputs, few enough rollouts (<= 2048; dynamics(): <= 1024).  The value: from the forward launch (default integrator),
 *      from the backward launch (MF_LOSS_VALUE_IN_BACKWARD), or from mf_physics_loss_value_* on the written rows.
 *   2  the BACKWARD of a saturated launch (round 6): the positions-only one-point-per-lane kernels (float32 MF_MATH_FAST, rigid body of <= 64
 *      points, time-major, beyond the component-parallel / record-reading ranges: > 8192 rollouts of a 4-point body) form dL/dXs at the
 *      stamped rows themselves -- pass MfRolloutBwdBufs.loss and NO MfRolloutFwdBufs.loss; the value comes from mf_physics_loss_value_* on
 *      the forward's rows or, with MF_LOSS_VALUE_IN_BACKWARD, from the backward launch (partial: S[B], near and w required).  The step loses the loss-gradient launch and the dense [T][B][3] gradient (98 MB at
 *      16 384 rollouts, a tenth of its rows non-zero).
 *   3  the BACKWARD of a component-parallel launch in its early-recompute form (no record, more than one wave per SIMD: 4097 .. 8192
 *      rollouts of a <= 4-point body, either integrator; round 6): as 2 -- MfRolloutBwdBufs.loss with near and w, no MfRolloutFwdBufs.loss --
 *      without MF_LOSS_VALUE_IN_BACKWARD (the value: mf_physics_loss_value_*).  The other one-wave launches (2049 .. 4096 rollouts;
 *      dynamics() from 1025) answer 0: there the fused kernel loses what the two loss launches cost (profiles/r6_ab_one_wave_loss.txt).
 *   0  neither: run mf_physics_loss_* on the outputs. */
int mf_rollout_loss_fusable(const MfRolloutDesc* desc);

/* 1 where mf_rollout_bwd_f32 with a positions-only upstream (gXs or a fused loss; the other five NULL) sends the cell gradients of this
 * launch through per-workgroup LDS windows (saturated launches of a <= 4-point body on ONE shared power-of-two map pair): a workgroup
 * adds its window to gradient copy blockIdx % grad_copies once, at its end, so desc->grad_copies may stay small (64) -- the caller's
 * reduction over the copies (mf_reduce_grad_copies_*) is what grows with them: 0.125 ms at 256 copies of a 256 x 256 pair. */
int mf_rollout_bwd_window(const MfRolloutDesc* desc);

/* Bytes of MfRolloutFwdBufs.rec / MfRolloutBwdBufs.rec for this launch shape; 0 where the kernels chosen for it keep no record
 * (then pass NULL).  The record pays while the launch is bound by the instruction stream of its waves: few rollouts of a small
 * body (forward +5 %, backward -17 % at 1024 rollouts x 4 points), and bodies of 5..512 points up to two waves per SIMD
 * (16 bytes per rollout-step; backward 2.05 -> 1.04 ms at 64 rollouts x 223 points, 1.38 -> 0.97 ms at 1024 x 32). */
long long mf_rollout_record_bytes(const MfRolloutDesc* desc);
/* 1 where mf_rollout_fwd_f32, given `zmu_scratch` and no `zmu`, fills the scratch with the interleaved (z, mu) pair (few-point bodies on the
 * component-parallel kernels from half a wave per SIMD up): the caller may hand that buffer to mf_rollout_bwd_f32 as `zmu` for the same
 * step and maps -- the backward then runs no interleave pass of its own. */
int mf_rollout_fwd_stages_zmu(const MfRolloutDesc* desc);
/* The same for the _f64 entry points: non-zero only for the float64 VALIDATION build of the component-parallel kernels
 * (points_per_lane = MF_LANES_COMPONENT: the float32 kernels' source instantiated on double, 32-byte quads), which exists so that the
 * code the BASELINE configurations run can be held to the float64 oracle (dphysics.py:172-272, 467-528 under float64) over the full
 * horizon, where float32 trajectories are chaotic. */
long long mf_rollout_record_bytes_f64(const MfRolloutDesc* desc);

/* Point slots per Fs/Ff row the kernels chosen for (B, N, points_per_lane) need (>= N; -1 on a bad descriptor). */
int mf_rollout_force_stride(const MfRolloutDesc* desc);
int mf_rollout_fwd_f32(const MfRolloutDesc* desc, const MfRolloutFwdBufs* bufs, void* hip_stream);
int mf_rollout_fwd_f64(const MfRolloutDesc* desc, const MfRolloutFwdBufs* bufs, void* hip_stream);

/* The start state DPhysics uses when the caller gives none (dphysics.py:554-559): x = 0, xd = (v_0, 0, 0), R = I,
 * omega = (0, 0, w_0) with (v_0, w_0) = controls[b][0]; one launch.  controls S[B][T][2]; outputs S[B][3], [3], [3][3], [3]. */
int mf_rollout_default_state_f32(int32_t B, int32_t T, const float* controls, float* x0, float* xd0, float* R0, float* w0, void* hip_stream);
int mf_rollout_default_state_f64(int32_t B, int32_t T, const double* controls, double* x0, double* xd0, double* R0, double* w0, void* hip_stream);

/* Device buffers of the backward rollout (reverse-time adjoint of the same scan; per-step intermediates are
 * recomputed from the saved per-step states, which are the forward's own outputs).  `desc` must be the forward's.
 * Upstream gradients use the outputs' layout and may each be NULL (= zeros).  Gradient outputs: gz/gmu are
 * ACCUMULATED with atomic adds (zero them first; S[1 or B][H][W] like z/mu); the others are overwritten. */
typedef struct MfRolloutBwdBufs {
  const void* z;        /* as forward */
  const void* mu;       /* as forward (NULL = ones) */
  const void* controls; /* S[B][T][2] */
  const void* ts;       /* S[T] */
  const void* points;   /* S[N][3] */
  const int32_t* part;  /* int32[N] */
  const void* x_init;   /* S[B][3] x0 AFTER the forward's terrain snap (the forward's in/out x0 buffer) */
  const void* xd0;      /* S[B][3] */
  const void* R0;       /* S[B][3][3] */
  const void* w0;       /* S[B][3] */
  const void* Xraw;     /* saved forward outputs: unshifted positions, */
  const void* Xds;      /*   velocities, */
  const void* Rs;       /*   rotations, */
  const void* Omegas;   /*   angular velocities */
  const void* gXs;      /* upstream dL/dXs (the SHIFTED positions the API returned) ... */
  const void* gXds;
  const void* gRs;
  const void* gOmegas;
  const void* gFs;
  const void* gFf;
  const void* zeros;    /* >= 9 zero scalars of type S; required when any of the six upstream pointers is NULL */
  void* gz;             /* out (atomic accumulate): dL/dz, S[map_shared ? max(grad_copies,1) : B][H][W] */
  void* gmu;            /* out (atomic accumulate): dL/dmu, same shape; NULL to skip */
  void* gcontrols;      /* out: S[B][T][2]; may be NULL = gradient not wanted (round 6: every kernel; before, only where
                           mf_rollout_bwd_wants_gcontrols(desc) returned 0) -- 8 B per rollout-step of stores less */
  void* gx0;            /* out: S[B][3] (z component is 0 unless skip_snap); NULL to skip */
  void* gxd0;           /* out: S[B][3] */
  void* gR0;            /* out: S[B][3][3] */
  void* gw0;            /* out: S[B][3] */
  const void* joint_angles; /* the forward's S[B][T][4] flipper angles (desc->has_joints), else NULL */
  void* gjoint_angles;      /* out: dL/d(joint_angles), S[B][T][4], through update_joints AND the per-step inertia
                               (dphysics.py:191-197, 326-358); NULL to skip.  Rows the scheme never reads (the last one of the
                               default integrator) are not written: hand in zeros. */
  const void* rec;          /* the record the forward wrote (MfRolloutFwdBufs.rec of the same desc), or NULL: recompute */
  const MfRolloutLoss* loss; /* the forward's fused physics loss (then all six upstream pointers are NULL: the kernel forms dL/dXs
                               itself from loss->Xs, gt, w and gloss), or NULL */
  void* zmu_scratch;        /* optional scratch, 2*H*W floats (8-byte aligned), as MfRolloutFwdBufs.zmu_scratch: with a SHARED float32 map
                               pair the record-reading backward (two to four waves of rollouts per CU) re-gathers a cell's (z, mu)
                               with ONE 8-byte load from an interleaved copy it stages here (B = 4096: 0.45 -> 0.40 ms) */
  const void* zmu;          /* optional, float32: the SHARED maps already interleaved (MfRolloutFwdBufs.zmu of the same step) */
} MfRolloutBwdBufs;

/* 1 if the backward kernels chosen for this descriptor always write the control gradient (gcontrols must then be a buffer), 0 if
 * they can skip it (gcontrols may be NULL).  Round 6: 0 for every descriptor -- the component-parallel kernels compile the gradient out
 * (its dot product, sums and stores), the others keep their instruction stream and drop the stores' traffic. */
int mf_rollout_bwd_wants_gcontrols(const MfRolloutDesc* desc);
int mf_rollout_bwd_f32(const MfRolloutDesc* desc, const MfRolloutBwdBufs* bufs, void* hip_stream);
int mf_rollout_bwd_f64(const MfRolloutDesc* desc, const MfRolloutBwdBufs* bufs, void* hip_stream);

/* ---- LSS BEV voxel pooling (lss.py:238-280) ------------------------------------------------------------------
 * Points are the B * n_per_sample frustum points of a batch (n_per_sample = cams * D * fH * fW), in the reference's
 * flattening order; features are row-major [point][C]; the output is the reference's [B][nz*C][nx][ny] grid
 * (channel index = iz*C + c, lss.py:274-278).  Voxel index = trunc((geom - off) / dx) in float32 with
 * off = bx - dx/2 (lss.py:246); points outside [0,n) on any axis are dropped (lss.py:253-255).
 * Usage: workspace = mf_bev_splat_workspace_bytes(desc) bytes of device memory; mf_bev_splat_prepare() once per
 * geometry; then any number of _fwd / _bwd calls with that workspace. */
typedef struct MfSplatDesc {
  int32_t B;            /* samples */
  int32_t n_per_sample; /* frustum points per sample */
  int32_t C;            /* feature channels per point */
  int32_t nx, ny, nz;   /* BEV grid (gen_dx_bx, terrain_encoder/utils.py:136-141) */
  float off[3];         /* bx - dx/2, computed in float32 like the reference */
  float dx[3];          /* voxel size */
  int32_t lift_D;       /* mf_bev_lift_splat_*: depth bins per pixel (D) ... */
  int32_t lift_hw;      /* ... and pixels per camera feature map (fH * fW); n_per_sample = cameras * lift_D * lift_hw.  0 otherwise */
} MfSplatDesc;

size_t mf_bev_splat_workspace_bytes(const MfSplatDesc* desc); /* 0 on a bad descriptor */
int mf_bev_splat_prepare(const MfSplatDesc* desc, const float* geom /* [B*n_per_sample][3] */, void* workspace, void* hip_stream);
/* The same plan from the camera models instead of a geometry tensor: get_geometry (lss.py:204-224) evaluated per point
 * inside the key pass, in the reference's order of float32 operations.  frustum[pts_per_cam][3] = (u, v, d) of
 * create_frustum (lss.py:191-202), pts_per_cam = D*fH*fW, n_per_sample = cameras * pts_per_cam;
 * cams[B*cameras][24] = post_trans[3], inverse(post_rots)[9], (rots x inverse(intrins))[9], trans[3], matrices row-major. */
int mf_bev_splat_prepare_cameras(const MfSplatDesc* desc, const float* frustum, int32_t pts_per_cam, const float* cams,
                                 void* workspace, void* hip_stream);
/* The same plan straight from the calibration tensors LiftSplatShoot.forward receives (lss.py:282-296): rots, intrins, post_rots
 * [B*cameras][3][3] and trans, post_trans [B*cameras][3], float32 row-major -- BOTH 3 x 3 inversions of get_geometry (lss.py:212, 218:
 * torch.inverse(post_rots), torch.inverse(intrins)) and the product rots x inverse(intrins) are formed inside the key kernel (float64
 * adjugate rounded once to float32), so a data loader with per-sample augmentation (terrain_encoder/utils.py:110-133) pays no
 * torch.inverse -- two LU launches and a host synchronisation -- per step.  For diagonal-plus-translation intrinsics and scale / flip /
 * crop augmentations the inverse is exact up to one rounding per entry and the keys equal mf_bev_splat_prepare_cameras' bit for bit;
 * an in-plane rotation (rot_lim) makes the two differ in the last bit of some entries, which moves a point only if it lies within
 * ~1e-6 voxel of a voxel face (tests/test_splat_gpu.py). */
int mf_bev_splat_prepare_rig(const MfSplatDesc* desc, const float* frustum, int32_t pts_per_cam, const float* rots, const float* trans,
                             const float* intrins, const float* post_rots, const float* post_trans, void* workspace, void* hip_stream);
/* out[B][nz*C][nx][ny] = per-voxel sums of x[B*n_per_sample][C]; every output element is written (zeros where empty) */
int mf_bev_splat_fwd_f32(const MfSplatDesc* desc, const float* x, const void* workspace, float* out, void* hip_stream);
int mf_bev_splat_fwd_f64(const MfSplatDesc* desc, const double* x, const void* workspace, double* out, void* hip_stream);
/* gx[B*n_per_sample][C] = gout at the point's voxel, 0 for dropped points (QuickCumsum.backward, utils.py:174-181) */
int mf_bev_splat_bwd_f32(const MfSplatDesc* desc, const float* gout, const void* workspace, float* gx, void* hip_stream);
int mf_bev_splat_bwd_f64(const MfSplatDesc* desc, const double* gout, const void* workspace, double* gx, void* hip_stream);

/* The lift fused into the splat (lss.py:63-71 + :238-280): point p = ((cam * D + d) * fHW + pixel) carries
 * depth[p] * ctx[cam * fHW + pixel][0..C) without the [points][C] tensor ever being materialised.
 *   depth  S[B*cameras][D][fH][fW]   softmax depth distribution (= point order)
 *   ctx    S[B*cameras][fH][fW][C]   context features, PIXEL-major
 * fwd: out[B][nz*C][nx][ny] as mf_bev_splat_fwd.  bwd: g_depth (like depth), g_ctx (like ctx); rows_scratch = S[B*nz*nx*ny][C]
 * (the BEV gradient re-laid voxel-major for the occupied tiles).  Same prepared workspace as the plain splat. */
int mf_bev_lift_splat_fwd_f32(const MfSplatDesc* desc, const float* depth, const float* ctx, const void* workspace, float* out, void* hip_stream);
int mf_bev_lift_splat_fwd_f64(const MfSplatDesc* desc, const double* depth, const double* ctx, const void* workspace, double* out, void* hip_stream);
int mf_bev_lift_splat_bwd_f32(const MfSplatDesc* desc, const float* depth, const float* ctx, const void* workspace, const float* gout,
                              float* rows_scratch, float* g_depth, float* g_ctx, void* hip_stream);
int mf_bev_lift_splat_bwd_f64(const MfSplatDesc* desc, const double* depth, const double* ctx, const void* workspace, const double* gout,
                              double* rows_scratch, double* g_depth, double* g_ctx, void* hip_stream);

/* ---- fused physics loss (losses.py:102-127) -------------------------------------------------------------------
 * loss = mean_{b,j,c} ((Xs[b, nearest[b,j], c] - Xgt[b,j,c]) * w[b,j])^2,  w = 1 / (1 + gamma * gt_ts[b,j]).
 * Xs is addressed as Xs[b*x_stride_b + t*x_stride_t + c] (elements), so both rollout output layouts work in place.
 * _fwd writes ceil(B*T2/256) per-workgroup partial sums (loss = sum(partial) / (B*T2*3)); _bwd writes d loss/d Xs into a gXs that is ZERO on entry (same
 * strides as Xs; stamps of one rollout that share a step add up, a step one stamp has to itself is stored) given the upstream scalar gradient gloss[0]. */
typedef struct MfLossDesc {
  int32_t B, T1, T2;            /* rollouts, predicted steps, ground-truth stamps */
  int32_t reserved;
  int64_t x_stride_b, x_stride_t;
  double gamma;
} MfLossDesc;
int mf_physics_loss_fwd_f32(const MfLossDesc* desc, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest, float* partial, void* hip_stream);
int mf_physics_loss_fwd_f64(const MfLossDesc* desc, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest, double* partial, void* hip_stream);
int mf_physics_loss_bwd_f32(const MfLossDesc* desc, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest, const float* gloss, float* gXs, void* hip_stream);
int mf_physics_loss_bwd_f64(const MfLossDesc* desc, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest, const double* gloss, double* gXs, void* hip_stream);
/* nearest[b][j] = argmin_t |pred_ts[b][t] - gt_ts[b][j]|, the first minimum (losses.py:116 `torch.argmin(torch.abs(pred_ts.unsqueeze(1) -
 * gt_ts.unsqueeze(2)), dim=2)`): the index table the entry points above take, without the two [B,T2,T1] temporaries of the reference's
 * form.  pred_ts rows of T1, gt_ts rows of T2 stamps, row strides in elements (0 = one row shared by all rollouts); nearest int32[B][T2]. */
int mf_nearest_steps_f32(int32_t B, int32_t T1, int32_t T2, const float* pred_ts, long long pred_stride_b, const float* gt_ts, long long gt_stride_b, int32_t* nearest, void* hip_stream);
int mf_nearest_steps_f64(int32_t B, int32_t T1, int32_t T2, const double* pred_ts, long long pred_stride_b, const double* gt_ts, long long gt_stride_b, int32_t* nearest, void* hip_stream);
/* The same loss, finished inside the launch: `loss[0]` receives the mean (what `partial.sum() / (3 B T2)` gives, summed in block
 * order).  `partial` holds ceil(B*T2/256) scalars of scratch; `ticket` is ONE zero-initialised uint32 the library resets itself --
 * reusable launch after launch by calls ordered on one stream (two streams need two tickets).  `zero_fill` (may be NULL): a buffer
 * of `zero_count` scalars cleared by the same launch -- the gXs a later mf_physics_loss_bwd_* scatters into. */
int mf_physics_loss_value_f32(const MfLossDesc* desc, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest, float* partial, uint32_t* ticket, float* loss, float* zero_fill, long long zero_count, void* hip_stream);
int mf_physics_loss_value_f64(const MfLossDesc* desc, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest, double* partial, uint32_t* ticket, double* loss, double* zero_fill, long long zero_count, void* hip_stream);
/* The reduction that follows a shared-map mf_rollout_bwd_* (MfRolloutDesc.grad_copies private copies of each map gradient,
 * pool = [n_maps][copies][n]):  out[m][i] = sum_c pool[m][c][i], and pool is left ZEROED -- a caller that keeps the pool across
 * steps never fills it again (replaces a zero fill + `maps.sum(1)`, scripts/train.py's optimizer step reads `out`). */
int mf_reduce_grad_copies_f32(float* pool, int n_maps, int copies, long long n, float* out, void* hip_stream);
int mf_reduce_grad_copies_f64(double* pool, int n_maps, int copies, long long n, double* out, void* hip_stream);

/* ---- terrain staging between the BEV heads and the rollout (scripts/train.py:93-99, 233-235; lss.py:158) ------------------
 * One pass over the head outputs geom, diff, friction (each float32 [B][H][W]):
 *   terrain[B][H][W] = geom - diff (may be NULL);  z / mu [B][h][w] = k x k average pooling (stride k, h = H / k, w = W / k,
 *   overhanging cells dropped like torch.nn.AvgPool2d) of terrain / friction;  zmu[B][h][w][2] = (z, mu) interleaved -- what
 *   MfRolloutFwdBufs.zmu takes (may be NULL).
 * _bwd: g_geom = g_terrain + upsample(gz) / k^2, g_diff = -g_geom, g_friction = upsample(gmu) / k^2; g_terrain, gz, gmu may
 * each be NULL (= zeros); every element of the three outputs is written. */
typedef struct MfStageDesc {
  int32_t B, H, W, k;
} MfStageDesc;
int mf_terrain_stage_fwd_f32(const MfStageDesc* desc, const float* geom, const float* diff, const float* friction, float* terrain,
                             float* z, float* mu, float* zmu, void* hip_stream);
int mf_terrain_stage_bwd_f32(const MfStageDesc* desc, const float* g_terrain, const float* gz, const float* gmu, float* g_geom,
                             float* g_diff, float* g_friction, void* hip_stream);

/* ---- estimate_heightmap (cloudproc.py:88-148): per-cell maximum height of a point cloud + measurement mask ------------
 * points[n][3] float32 (rows with NaNs, points within r_min of the origin in xy, and points outside the open box
 * (-d_max, d_max)^2 x (h_min, h_max) are dropped); x_bins[nx] / y_bins[ny] are the bin edges the reference builds with
 * torch.arange(-d_max, d_max, grid_res) -- a point falls into bin torch.bucketize(x, bins) - 1; scratch = nx*ny int32;
 * hm[2][nx][ny]: hm[0][ix][iy] = max z of the cell (0 where empty), hm[1][ix][iy] = 1.0 where measured (the reference's
 * transposed output layout: first axis = x). */
typedef struct MfHeightmapDesc {
  int32_t n_points, nx, ny;
  float d_max, h_min, h_max;
  float r_min;   /* < 0: no inner radius filter */
  float inv_res; /* 1 / grid_res: first guess of the bin (the edges decide) */
} MfHeightmapDesc;
int mf_estimate_heightmap_f32(const MfHeightmapDesc* desc, const float* points, const float* x_bins, const float* y_bins,
                              int32_t* scratch, float* hm, void* hip_stream);

/* ---- interpolate_grid (dphysics.py:385-455) on its own: the reference's public sampling method, bug for bug --
 * grid[Bg][H][W] with Bg = B, or one map for all rows (map_shared); xq, yq [B][N] query positions; z_out [B][N];
 * n_out [B][N][3] unit normals (may be NULL: return_normals=False); cells_out [B][N][4] int32 (may be NULL) = the CLAMPED flat
 * indices (c, f, l, fl) the value was read from, and frac_out [B][N][2] (may be NULL) = (x_frac, y_frac): the integer half of
 * the function, exposed so a test can compare it with `((q + d_max) / grid_res).long()` directly.  The device code is the
 * rollout kernels' own (locate_m / gather / blend of rollout_fwd_kernel.h), in either arithmetic mode. */
typedef struct MfInterpDesc {
  int32_t B, N, H, W;
  int32_t map_shared;
  int32_t math_mode; /* MF_MATH_* (float32; float64 is always exact) */
  double grid_res, d_max;
} MfInterpDesc;
int mf_interpolate_grid_f32(const MfInterpDesc* desc, const float* grid, const float* xq, const float* yq, float* z_out, float* n_out,
                            int32_t* cells_out, float* frac_out, void* hip_stream);
int mf_interpolate_grid_f64(const MfInterpDesc* desc, const double* grid, const double* xq, const double* yq, double* z_out, double* n_out,
                            int32_t* cells_out, double* frac_out, void* hip_stream);

/* Text of the calling thread's last error ("" if none).  THREAD-LOCAL: host threads driving different streams each read the message
 * of their own failed call (the reference raises a Python exception in the calling thread, dphysics.py:575,579 asserts). */
const char* mf_last_error(void);
/* Which rollout kernel the calling thread's last mf_rollout_fwd_* / mf_rollout_bwd_* call launched: the demangled kernel template
 * with its parameters (lane mapping, integrator, arithmetic, record / streaming mode ...), grid, workgroup size and the thread's
 * launch count, e.g. "mf::rollout_bwd_cp_kernel<float, 1, true, false, 3, 12, 3, false> grid=256 block=192 launches=7"; "" before
 * the first launch.  Diagnostic only (bench.py reports it as config.launch; the reference has one code path, dphysics.py:530-594,
 * this library has a dispatcher).  Thread-local. */
const char* mf_last_launch(void);
/* Library version, e.g. "monoforce_hip 0.1 gfx950". */
const char* mf_version(void);
/* sizeof() of an ABI struct by name ("MfRolloutDesc", ...), -1 if unknown: lets bindings verify their mirrors. */
int mf_sizeof(const char* struct_name);

#ifdef __cplusplus
}
#endif
#endif /* MONOFORCE_HIP_H */
