// Fused DPhysics rollout, backward pass for bodies of 5..512 contact points in launches of up to two waves per SIMD (gfx950) --
// the reference's own operating point, 4..64 rollouts of its 175- / 223-point robots
// (/root/reference/monoforce/examples/diff_physics.ipynb:199-226, scripts/train.py --bsz 4, n_sim_trajs = 64), and everything between
// that and the component-parallel kernels of 4-point bodies.  One contact point per lane: G = 8 / 16 / 32 / 64 lanes per rollout
// inside a wave, or ONE rollout over 2 / 4 / 8 waves of a workgroup.  float32 MF_MATH_FAST, rigid body; the default integrator
// (torchdiffeq fixed-grid euler, dphysics.py:499-528) and dynamics() (:467-497).
//
// Same adjoint as rollout_bwd_kernel.h (reverse-time vector-Jacobian product of `forward_kinematics`, dphysics.py:172-272, from
// the saved state rows), re-organised around what one wave per SIMD pays for: instructions and group exchanges.  The general
// kernel spends, per step, four exchanges (contact count, torque, one adjoint scalar, 23 adjoint sums) and 168 DPP
// reductions of 1425 instructions (G = 256).  Here
//   * the forward keeps a 16-byte record per rollout-step -- the contact count and the unclamped angular acceleration it
//     evaluated (MfRolloutFwdBufs.rec for this mapping) -- so the recompute of a step needs no reduction at all, and the clamp
//     of omega_d is gated on the forward's own value;
//   * the adjoint of the body state (x, xd, R, w) is kept UN-SUMMED over the contact points: every lane carries the partial its
//     point contributed, pushed through the step's linear recurrence (the coefficients -- R, w, h -- are group-uniform),
//     and summed once after the loop.  Only what a step's per-point chain reads as a total is exchanged: the six components of
//     the velocity adjoints, the two control-gradient outputs and the scalar gS = dL/d(sum of contact weights) -- nine values;
//   * gS enters the step linearly (through dL/d(dh) = ... + gS kappa_j), and nothing that depends on it is read as a total
//     before the step after next: it rides in the SAME exchange as the velocity adjoints and is applied one step late as a
//     correction (position and rotation partials, the four cell accumulators of the point);
//   * over several waves (and for a whole-wave group) the exchange is TransposedExchange (mf_common.h).
// One exchange -- and over several waves one barrier -- per step (dynamics(): two); the state rows, controls, time grid and record
// are group-uniform loads.  632 instructions per step at G = 256.
// (Measured and dropped: the adjoint-independent half of step n - 1 -- pose, gathers, contact model, gates -- rebuilt in the shadow
//  of step n's exchange.  The ~75 values it carries across the loop edge push the kernel past 256 architectural VGPRs; the
//  accumulator-register copies that follow cost more than the covered latency: 1.04 -> 1.12 ms at 64 x 223 points, 0.97 -> 1.06 ms
//  at 1024 x 32.  The loop runs at ~6.6 cycles per instruction of ONE wave per SIMD: its time is its instruction count.)
#pragma once
#include "rollout_bwd_kernel.h"
#include <type_traits>

namespace mf {

// o += a x b with every product fused into the accumulation: two instructions per component (a cross product formed on its own
// and added afterwards takes three)
#define MF_CROSS_ACC(o0, o1, o2, a, b)                                 \
  do {                                                                 \
    (o0) = mf_fma(-(a)[2], (b)[1], mf_fma((a)[1], (b)[2], (o0)));          \
    (o1) = mf_fma(-(a)[0], (b)[2], mf_fma((a)[2], (b)[0], (o1)));          \
    (o2) = mf_fma(-(a)[1], (b)[0], mf_fma((a)[0], (b)[1], (o2)));          \
  } while (0)

constexpr int kMwRecFloats = 4;   // per rollout-step: (sum of contact weights, omega_d before its clamp [3])

// Map gradients.  A point adds to the four cells of its footprint per map and step; it moves <= 0.2 cell per step, so the
// contributions are summed in registers while the footprint stays (rollout_bwd_kernel.h) and written out when it changes, one
// iteration late, behind that step's loads.  TILE = 0 writes them with device-scope float atomics: they execute at the memory side
// (a fabric transaction each, ~20 ns apart on one address) and sit in the wave's in-order `vmcnt` queue in front of every later
// load -- 0.20-0.24 ms of a 1.0-1.1 ms launch (A/B build without them).  TILE > 0 sends them to an LDS tile of TILE x TILE cells
// (x 2 maps) that follows the robot: `ds_add_f32` from the lanes whose footprint just changed; the tile goes out to the gradient
// maps (one atomic per touched cell) when the body has moved a quarter of the window from its centre, and after the loop.
// Points off the map edge or outside the window (the reference folds and wraps their flat indices, dphysics.py:427-435) keep
// the direct route.  (Measured and dropped: EVERY step's contributions straight into the tile, no register accumulators --
// the LDS retires about one atomic lane per cycle and CU, so 8 instructions x 64 lanes x 4 waves cost 2000 cycles per step:
// 1.53 instead of 1.10 ms at 64 rollouts x 223 points.)  The host picks TILE > 0 where all workgroups fit the CUs' LDS at once.
// INTEG: the default integrator (torchdiffeq fixed-grid euler, dphysics.py:499-528) keeps ONE exchange per step; dynamics()
// (semi-implicit Euler + Rodrigues, :467-497, :274-324) reads the totals of the velocity adjoints AFTER its own step's rotation
// update has been pushed through the partials, so it takes a second, six-value exchange in the middle of the step.
// S: float (fast math: the kernels of the reference's own operating point) or double (the VALIDATION build of the same source, exact
// arithmetic: csrc/rollout_mw_f64.hip)
template <typename S, int G, bool XS_ONLY, int TILE, int INTEG = MF_INTEG_ODEINT_EULER>
__global__ void __launch_bounds__(G > 64 ? G : 64) rollout_bwd_mw_kernel(const RolloutBwdArgs<S> a) {
  constexpr bool kFast = std::is_same<S, float>::value;
  using M = Mth<S, kFast>;
  constexpr int NW = G > 64 ? G / 64 : 1;
  // G > 64: one rollout per workgroup of G / 64 waves.  G <= 64: 64 / G rollouts per one-wave workgroup, the exchange is a DPP sum
  const int tid = blockIdx.x * (G > 64 ? G : 64) + threadIdx.x;
  const int b = tid / G;
  const int gl = tid % G;          // one contact point per lane
  if (b >= a.B) return;            // whole groups leave together (G <= 64 only: no barrier in those kernels)
  const S one = (S)1, zero = (S)0;
  const int HW = a.H * a.W, last = HW - 1;
  __shared__ S gs_lds[G > 64 ? 2 * NW * kGroupSumMaxValues : 1];
  constexpr int NT = G > 64 ? 1 : 64 / G;                       // rollouts = tiles per workgroup
  // (rows of TILE cells at a pitch of TILE + 1 words: neighbours in x would otherwise share an LDS bank, and a wave's 64 points
  //  spread over a dozen columns -- measured at TILE = 64 without the pad: 1.54 instead of 1.10 ms at 64 x 223 points)
  constexpr int TS = TILE + 1;
  constexpr int T2 = TS * TILE;                                 // words per plane
  __shared__ S tile_lds[TILE > 0 ? NT * 2 * T2 : 1];
  S* const tile_z = tile_lds + (G > 64 ? 0 : (int)(threadIdx.x / G) * 2 * T2);      // this rollout's tile: z plane, then mu plane
  int ocx = 0, ocy = 0;                                        // cell under the tile's centre (group-uniform)
  GroupSum<G, S> gs;
  gs.lds = gs_lds;
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S* zmap = a.z;
  const bool has_mu = a.mu != nullptr;
  const S* mumap = has_mu ? a.mu : a.z;
  const unsigned goff = a.map_shared ? (unsigned)(b % a.grad_copies) * (unsigned)HW : (unsigned)b * (unsigned)HW;
  S* gzmap = a.gz;
  const bool want_gmu = a.gmu != nullptr && has_mu;
  S* gmumap = want_gmu ? a.gmu : a.gz;

  const bool act = gl < a.N;
  const int ii = act ? gl : 0;
  const S P[3] = {a.points[ii * 3 + 0], a.points[ii * 3 + 1], a.points[ii * 3 + 2]};
  const int part = act ? a.part[ii] : -1;
  const S tcw = part < 0 ? zero : ((part & 1) ? a.half_ly : -a.half_ly);   // d(track speed)/d(w command); d/d(v command) = drv
  const S drv = part >= 0 ? one : zero;
  const S m0 = gl == 0 ? one : zero;     // upstream gradients of the body state enter ONE lane's partial
  S Iv[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) Iv[c] = a.Iinv[c];

  const size_t row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)a.B : 1;
  const size_t row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)b : (size_t)b * a.T;
  const S* ctrl = a.controls + (size_t)b * a.T * 2;
  S* gctrl = a.gcontrols + (size_t)b * a.T * 2;
  constexpr bool DYN = INTEG == MF_INTEG_DYNAMICS;
  const int n_steps = DYN ? a.T : a.T - 1;

  // un-summed adjoint of the body state: the sum over the workgroup's lanes is the adjoint
  S lx[3] = {zero, zero, zero}, lxd[3] = {zero, zero, zero}, lw[3] = {zero, zero, zero}, lR[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) lR[c] = zero;
  S laFs[3] = {zero, zero, zero}, laFf[3] = {zero, zero, zero};   // adjoint of this point's impulse accumulators (ODEINT outputs)

  struct StepIn {   // workgroup-uniform inputs of step n: the state it started from, its controls, step size and the forward's record
    S x[3], xd[3], R[9], w[3], cv, cw, t0, t1, csum, wraw[3];
  };
  struct UpIn {     // upstream gradients of output row n + 1
    S gXs[3], gXds[3], gRs[9], gOm[3], gFs[3], gFf[3];
  };
  auto load_step = [&](int n, StepIn& s) {
    // the state step n started from: output row n (default integrator: rows are the grid points); dynamics() records the state AFTER
    // each step, so step n starts from row n - 1 and step 0 from the initial state (pointer selects, no branch)
    const size_t row = row0 + (size_t)(DYN ? max(n - 1, 0) : n) * row_stride;
    const bool init = DYN && n == 0;
    const S* px = init ? a.x_init + b * 3 : a.Xraw + row * 3; const S* pxd = init ? a.xd0 + b * 3 : a.Xds + row * 3;
    const S* pw = init ? a.w0 + b * 3 : a.Om + row * 3; const S* pR = init ? a.R0 + b * 9 : a.Rs + row * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) { s.x[c] = px[c]; s.xd[c] = pxd[c]; s.w[c] = pw[c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) s.R[c] = pR[c];
    s.cv = ctrl[n * 2 + 0]; s.cw = ctrl[n * 2 + 1];
    if constexpr (DYN) { s.t0 = zero; s.t1 = a.dt; }
    else { s.t0 = a.ts[n]; s.t1 = a.ts[min(n + 1, a.T - 1)]; }
    const S* pr = a.rec + ((size_t)n * a.B + b) * kMwRecFloats;
    s.csum = pr[0]; s.wraw[0] = pr[1]; s.wraw[1] = pr[2]; s.wraw[2] = pr[3];
  };
  auto load_up = [&](int m, UpIn& u) {   // output row m
    const size_t row = row0 + (size_t)m * row_stride;
    const S* g1 = a.gXs + row * a.sXs;
#pragma unroll
    for (int c = 0; c < 3; ++c) u.gXs[c] = g1[c];
    if constexpr (!XS_ONLY) {
      const S* g2 = a.gXds + row * a.sXds; const S* g3 = a.gOm + row * a.sOm; const S* g4 = a.gRs + row * a.sRs;
      const size_t pt = row * a.N + min(gl, a.N - 1);
      const S* f1 = a.gFs + pt * a.sFs; const S* f2 = a.gFf + pt * a.sFf;
#pragma unroll
      for (int c = 0; c < 3; ++c) { u.gXds[c] = g2[c]; u.gOm[c] = g3[c]; u.gFs[c] = f1[c]; u.gFf[c] = f2[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) u.gRs[c] = g4[c];
    }
  };
  // upstream gradient of a state row: into lane 0's partials (and, by the caller, into the totals the step reads)
  auto add_upstream_partials = [&](const UpIn& u) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lx[c] += m0 * u.gXs[c];
      lR[c * 3 + 2] += (m0 * a.sink) * u.gXs[c];     // Xs = x + R[:, 2] * sink
    }
    if constexpr (!XS_ONLY) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { lxd[c] += m0 * u.gXds[c]; lw[c] += m0 * u.gOm[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] += m0 * u.gRs[c];
    }
  };

  // map-gradient accumulators of the point's footprint (rollout_bwd_kernel.h, plain form): a robot moves <= 0.2 cell per step
  unsigned acc_idx[4] = {0u, 0u, 0u, 0u}, st_idx[4] = {0u, 0u, 0u, 0u};
  S acc_z[4] = {zero, zero, zero, zero}, acc_m[4] = {zero, zero, zero, zero}, st_z[4], st_m[4];
  bool st_pending = false;
  int acc_t = -1, st_t = -1;      // TILE: tile index of the accumulators' / the stash's first cell, -1 = direct route (acc_idx / st_idx)
#pragma unroll
  for (int q = 0; q < 4; ++q) st_z[q] = st_m[q] = zero;
  // four cells per map to the LDS tile (t >= 0: index of the footprint's first cell) or straight to the gradient maps
  auto emit_cells = [&](int t, const unsigned (&idx)[4], const S (&vz)[4], const S (&vm)[4]) {
    if (TILE > 0 && t >= 0) {
      S* tz = tile_z + t;
      __hip_atomic_fetch_add(tz, vz[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(tz + TS, vz[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(tz + 1, vz[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(tz + TS + 1, vz[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (want_gmu) {
        S* tm = tz + T2;
        __hip_atomic_fetch_add(tm, vm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(tm + TS, vm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(tm + 1, vm[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(tm + TS + 1, vm[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) atomic_add(at32(gzmap, goff + idx[q]), vz[q]);
      if (want_gmu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomic_add(at32(gmumap, goff + idx[q]), vm[q]);
      }
    }
  };
  auto flush_stash = [&]() {
    if (st_pending) emit_cells(st_t, st_idx, st_z, st_m);
    st_pending = false;
  };
  // what the deferred gS correction of the previous iteration's step needs: kappa = d(dh adjoint)/d(gS), the footprint's weights
  // and the height blend's derivatives over res
  S dk = zero, dzx = zero, dzy = zero, dw4[4] = {zero, zero, zero, zero};

  auto tile_sync = [&]() { if constexpr (G > 64) __syncthreads(); };
  auto tile_clear = [&]() {
    if constexpr (TILE > 0) {
      for (int i = gl; i < 2 * T2; i += G) tile_z[i] = zero;
    }
  };
  auto tile_flush = [&]() {   // the tile's non-zero cells to the gradient maps, the tile back to zero
    if constexpr (TILE > 0) {
      const int ox = ocx - TILE / 2, oy = ocy - TILE / 2;
      for (int k = gl; k < TILE * TILE; k += G) {
        const int i = (k % TILE) + TS * (k / TILE);
        const S vz = tile_z[i], vm = tile_z[T2 + i];
        tile_z[i] = zero; tile_z[T2 + i] = zero;
        if (vz != zero || vm != zero) {
          const unsigned flat = (unsigned)((oy + (k % TILE)) + a.H * (ox + (k / TILE)));
          atomic_add(at32(gzmap, goff + flat), vz);
          if (want_gmu) atomic_add(at32(gmumap, goff + flat), vm);
        }
      }
    }
  };
  auto centre_cell = [&](const S* xs, int& cx, int& cy) {
    const S lim = (S)262144.0;
    cx = (int)M::clamp((xs[0] + a.d_max) * a.inv_res, -lim, lim);
    cy = (int)M::clamp((xs[1] + a.d_max) * a.inv_res, -lim, lim);
  };
  if (!DYN && gl == 0 && a.gcontrols) { gctrl[(a.T - 1) * 2 + 0] = zero; gctrl[(a.T - 1) * 2 + 1] = zero; }   // the last control is never used by the explicit scheme

  S ex[9];      // the step's exchange: partials in, group totals out
  // The exchange itself.  Within a wave (G <= 64): nine DPP group sums.  Over several waves: TransposedExchange (mf_common.h) --
  // about half the instructions of nine plain workgroup sums (the loop: 762 -> 632 instructions per step at G = 256).
  // (a whole-wave group, G = 64, takes it too: 26 DPP adds and one LDS round trip instead of 54 DPP adds and nine readlanes)
  __shared__ __attribute__((aligned(4 * sizeof(S)))) S xch_lds[G >= 64 ? TransposedExchange<NW, S>::kWords : 4];
  TransposedExchange<NW, S> xch;
  xch.lds = xch_lds;
  auto post9 = [&](S (&v)[9]) {
    if constexpr (G < 64) {
      gs.template post<9>(v);
    } else {
      const S v8[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
      xch.post(v8, v[8]);
    }
  };
  auto wait9 = [&](S (&v)[9]) {
    if constexpr (G < 64) gs.template wait<9>(v);
    else xch.template wait<9>(v);
  };
  // One step of the reverse scan.  (cur, upn) = the inputs of step n, loaded an iteration ago; (nxt, up_nxt) receive those of step
  // n - 1.  The loop calls it twice per trip with the two buffer pairs swapped: no register copies at the back edge.
  auto one_step = [&](const int n, const StepIn& cur, const UpIn& upn, StepIn& nxt, UpIn& up_nxt) {
    S x[3], xd[3], R[9], w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = cur.x[c]; xd[c] = cur.xd[c]; w[c] = cur.w[c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = cur.R[c];
    const S cv = cur.cv, cw = cur.cw, h = cur.t1 - cur.t0;
    const S csum = cur.csum;
    const S wraw[3] = {cur.wraw[0], cur.wraw[1], cur.wraw[2]};

    // ---- geometry of the point under pose n, requests for its cells ----
    const S px = P[0] * R[0] + P[1] * R[1] + P[2] * R[2] + x[0];
    const S py = P[0] * R[3] + P[1] * R[4] + P[2] * R[5] + x[1];
    const S pz = P[0] * R[6] + P[1] * R[7] + P[2] * R[8] + x[2];
    const S r[3] = {px - x[0], py - x[1], pz - x[2]};
    const Cell<S> cell = locate_m<S, kFast>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
    // (ix, iy) of the footprint's first cell: locate_m's own arithmetic (trunc(u) == u - fraction exactly)
    const int ix = (int)(M::cell_coord(px, a.d_max, a.res, a.inv_res) - cell.fx), iy = (int)(M::cell_coord(py, a.d_max, a.res, a.inv_res) - cell.fy);
    S zc[4], mc[4];
#ifdef MF_MW_DBG_NOGATHER      // A/B hook (tools/build_variant.sh): the step without its map gathers
    zc[0] = zc[1] = zc[2] = zc[3] = P[2]; mc[0] = mc[1] = mc[2] = mc[3] = one;
#else
    gather4(zmap, moff, cell, last, zc);
    gather4(mumap, moff, cell, last, mc);
#endif
    // ---- deferred atomics of the previous iteration, prefetch of the next one's rows (younger than the gathers) ----
    flush_stash();
    load_step(max(n - 1, 0), nxt);
    load_up(DYN ? max(n - 1, 0) : n, up_nxt);      // the row step n - 1 produced

    // ---- the exchange posted by the previous iteration: totals of the velocity adjoints, the control gradient of step n + 1, gS ----
    // (`ex` lives across iterations: within a wave, G <= 64, post() leaves the group totals in it and wait() is empty)
#ifndef MF_MW_DBG_NOEXCHANGE
    wait9(ex);
#endif
    if (a.gcontrols && n + 1 < n_steps) { gctrl[(n + 1) * 2 + 0] = ex[6]; gctrl[(n + 1) * 2 + 1] = ex[7]; }
    {   // gS of step n + 1, one step late: dh adjoint += gS kappa  ->  height sample, position, rotation partials, cell accumulators
      const S dl = ex[8] * dk;
      const S g0 = -(dl * dzx), g1 = -(dl * dzy);
      lx[0] += g0; lx[1] += g1; lx[2] += dl;
#pragma unroll
      for (int c = 0; c < 3; ++c) { lR[0 * 3 + c] += g0 * P[c]; lR[1 * 3 + c] += g1 * P[c]; lR[2 * 3 + c] += dl * P[c]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc_z[q] -= dl * dw4[q];
    }
    if constexpr (TILE > 0) {
      // the window follows the body: once it has moved a quarter of the window, the accumulators go to the tile, the tile out to
      // the maps, and the window is centred under the body again (group-uniform; a workgroup-uniform branch for G > 64)
      int cx, cy;
      centre_cell(x, cx, cy);
      if (abs(cx - ocx) > TILE / 4 || abs(cy - ocy) > TILE / 4) {
        if (act) emit_cells(acc_t, acc_idx, acc_z, acc_m);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc_z[q] = acc_m[q] = zero; }
        acc_t = -1;      // (the zeroed accumulators keep their cells: the next change of footprint stashes zeros for them)
        tile_sync();
        tile_flush();
        ocx = cx; ocy = cy;
        tile_sync();
      }
    }
    // ---- upstream of the output row this step produced (default integrator: row n + 1; dynamics(): row n) ----
    S Sxd[3] = {ex[0], ex[1], ex[2]}, Sw[3] = {ex[3], ex[4], ex[5]};
    add_upstream_partials(upn);
    if constexpr (!XS_ONLY && !DYN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Sxd[c] += upn.gXds[c]; Sw[c] += upn.gOm[c];
        laFs[c] += act ? upn.gFs[c] : zero; laFf[c] += act ? upn.gFf[c] : zero;
      }
    }
    S gxdd[3], gwd[3];
    if constexpr (!DYN) {
      // ---- integrator backward: y' = y + h f(y) ----
#pragma unroll
      for (int c = 0; c < 3; ++c) { gxdd[c] = h * Sxd[c]; gwd[c] = h * Sw[c]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) lxd[c] += h * lx[c];                 // x' = x + h xd
#pragma unroll
      for (int c = 0; c < 3; ++c) {                                    // R' = R + h [w]x R, column by column
        const S gcol[3] = {h * lR[0 * 3 + c], h * lR[1 * 3 + c], h * lR[2 * 3 + c]};
        const S rc[3] = {R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]};
        MF_CROSS_ACC(lw[0], lw[1], lw[2], rc, gcol);                                // d/dw of (w x R_c) . g  =  R_c x g
        MF_CROSS_ACC(lR[0 * 3 + c], lR[1 * 3 + c], lR[2 * 3 + c], gcol, w);         // d/dR_c                  =  g x w
      }
    } else {
      // ---- update_state backward (dphysics.py:274-324), on the PARTIALS: every map below is linear in the adjoint with group-uniform
      //      coefficients.  R' = R M(w'),  w' = w + wd h,  M = I + K sin(th h) + K^2 (1 - cos(th h)),  K = [w']x / max(|w'|, eps) ----
      S wn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) wn[c] = w[c] + M::clamp(wraw[c], -a.omega_max, a.omega_max) * h;
      const S th = M::sqrt(wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]);
      const S den = mf_max(th, (S)1e-6);
      const S kv[3] = {M::div(wn[0], den), M::div(wn[1], den), M::div(wn[2], den)};
      S sn_, oc;
      M::sincos_small(th * h, &sn_, &oc);
      const S cs_ = one - oc;
      const S K[9] = {zero, -kv[2], kv[1], kv[2], zero, -kv[0], -kv[1], kv[0], zero};
      S K2[9], Mx[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          K2[i * 3 + j2] = K[i * 3 + 0] * K[0 * 3 + j2] + K[i * 3 + 1] * K[1 * 3 + j2] + K[i * 3 + 2] * K[2 * 3 + j2];
          Mx[i * 3 + j2] = ((i == j2 ? one : zero) + K[i * 3 + j2] * sn_) + K2[i * 3 + j2] * oc;
        }
      S gM[9], lRn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          gM[i * 3 + j2] = R[0 * 3 + i] * lR[0 * 3 + j2] + R[1 * 3 + i] * lR[1 * 3 + j2] + R[2 * 3 + i] * lR[2 * 3 + j2];      // R^T lR
          lRn[i * 3 + j2] = lR[i * 3 + 0] * Mx[j2 * 3 + 0] + lR[i * 3 + 1] * Mx[j2 * 3 + 1] + lR[i * 3 + 2] * Mx[j2 * 3 + 2];   // lR M^T
        }
      S ga = zero, gb = zero, gK[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) { ga += gM[c] * K[c]; gb += gM[c] * K2[c]; }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          S t = zero;      // d(K K) -> gM K^T + K^T gM
#pragma unroll
          for (int m = 0; m < 3; ++m) t += gM[i * 3 + m] * K[j2 * 3 + m] + K[m * 3 + i] * gM[m * 3 + j2];
          gK[i * 3 + j2] = sn_ * gM[i * 3 + j2] + oc * t;
        }
      const S gk[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
      S gth = ga * h * cs_ + gb * h * sn_;
      S gwn[3] = {M::div(gk[0], den), M::div(gk[1], den), M::div(gk[2], den)};
      if (th >= (S)1e-6) gth += M::div(-(gk[0] * wn[0] + gk[1] * wn[1] + gk[2] * wn[2]), den * den);
      if (th > zero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gwn[c] += M::div(gth * wn[c], th);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { lw[c] += gwn[c]; lxd[c] += h * lx[c]; }      // w' = w + wd h;  x' = x + xd' h
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] = lRn[c];
      // the totals this step's chain reads: a second exchange, after the rotation update has gone through the partials
      S tot6[6] = {lxd[0], lxd[1], lxd[2], lw[0], lw[1], lw[2]};
      if constexpr (G < 64) {
        gs.sum_n(tot6);
      } else {
        const S v8[8] = {tot6[0], tot6[1], tot6[2], tot6[3], tot6[4], tot6[5], zero, zero};
        xch.post(v8, zero);
        xch.template wait<6>(tot6);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { gxdd[c] = h * tot6[c]; gwd[c] = h * tot6[3 + c]; }      // xd' = xd + xdd h;  w' = w + wd h
    }

    // ---- forward recompute of the point's contact (rollout_fwd_kernel.h, PIPE path) ----
    const S vp[3] = {xd[0] + (w[1] * r[2] - w[2] * r[1]), xd[1] + (w[2] * r[0] - w[0] * r[2]), xd[2] + (w[0] * r[1] - w[1] * r[0])};
    const S len2 = R[0] * R[0] + R[3] * R[3] + R[6] * R[6];
    const S il = M::inv_len(len2);
    const S e[3] = {R[0] * il, R[3] * il, R[6] * il};
    const S tv_lo = cv - cw * a.half_ly, tv_hi = cv + cw * a.half_ly;
    const S tv = (part < 0) ? zero : ((part & 1) ? tv_hi : tv_lo);
    S zq, mub;
    blend2(cell, zc, mc, &zq, &mub);
    const S muq = has_mu ? mub : blend_ones(cell);
    const S gx = M::div(zc[1] - zc[0], a.res), gy = M::div(zc[2] - zc[0], a.res);
    const S inl = M::inv_len(gx * gx + gy * gy + one);
    const S nrm[3] = {-gx * inl, -gy * inl, inl};
    const S dh = pz - zq;
    const S cj = act ? M::sigmoid_m10(dh) : zero;
    const S vn = vp[0] * nrm[0] + vp[1] * nrm[1] + vp[2] * nrm[2];
    const S A = a.k * dh + a.damp * vn;
    const S F0[3] = {-(A * nrm[0]), -(A * nrm[1]), -(A * nrm[2])};
    const S inv_csum = M::div(one, csum);
    S F1[3], Fr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { F1[c] = F0[c] * cj * inv_csum; Fr[c] = M::clamp(F1[c], -a.mg, a.mg); }
    const S Nn = M::sqrt(Fr[0] * Fr[0] + Fr[1] * Fr[1] + Fr[2] * Fr[2]);
    const S cmdv[3] = {tv * e[0] - vp[0], tv * e[1] - vp[1], tv * e[2] - vp[2]};
    const S slip[3] = {muq * cmdv[0], muq * cmdv[1], muq * cmdv[2]};
    const S sn = slip[0] * nrm[0] + slip[1] * nrm[1] + slip[2] * nrm[2];
    S st[3], Gf[3], Ff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { st[c] = slip[c] - sn * nrm[c]; Gf[c] = Nn * st[c]; Ff[c] = M::clamp(Gf[c], -a.mg, a.mg); }
    if (!act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Fr[c] = Ff[c] = zero;
    }

    // ---- RHS backward ----
    S gtau[3];
    {
      const S m0_ = inside(wraw[0], -a.omega_max, a.omega_max) ? gwd[0] : zero;
      const S m1_ = inside(wraw[1], -a.omega_max, a.omega_max) ? gwd[1] : zero;
      const S m2_ = inside(wraw[2], -a.omega_max, a.omega_max) ? gwd[2] : zero;
#pragma unroll
      for (int c = 0; c < 3; ++c) gtau[c] = Iv[0 * 3 + c] * m0_ + Iv[1 * 3 + c] * m1_ + Iv[2 * 3 + c] * m2_;   // Iinv^T
    }
    const S f[3] = {Fr[0] + Ff[0], Fr[1] + Ff[1], Fr[2] + Ff[2]};
    S gf[3] = {gxdd[0] * a.inv_mass, gxdd[1] * a.inv_mass, gxdd[2] * a.inv_mass};      // xdd = sum F / m ...
    MF_CROSS_ACC(gf[0], gf[1], gf[2], gtau, r);      // ... and tau += r x f : df = gtau x r
    S gFr_[3], gG[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const S com = gf[c];
      // (default integrator: the force outputs are running impulses, their adjoints accumulate; dynamics(): the forces themselves)
      gFr_[c] = XS_ONLY ? com : (DYN ? (act ? upn.gFs[c] : zero) : h * laFs[c]) + com;
      const S gFf_ = XS_ONLY ? com : (DYN ? (act ? upn.gFf[c] : zero) : h * laFf[c]) + com;
      gG[c] = inside(Gf[c], -a.mg, a.mg) ? gFf_ : zero;
    }
    const S gNn = gG[0] * st[0] + gG[1] * st[1] + gG[2] * st[2];
    const S gst[3] = {Nn * gG[0], Nn * gG[1], Nn * gG[2]};
    const S gsn = -(gst[0] * nrm[0] + gst[1] * nrm[1] + gst[2] * nrm[2]);
    S gn[3], gslip[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { gn[c] = -sn * gst[c] + gsn * slip[c]; gslip[c] = gst[c] + gsn * nrm[c]; }
    const S gmuq = gslip[0] * cmdv[0] + gslip[1] * cmdv[1] + gslip[2] * cmdv[2];
    const S gcmd[3] = {muq * gslip[0], muq * gslip[1], muq * gslip[2]};
    S gvp[3] = {-gcmd[0], -gcmd[1], -gcmd[2]};
    const S gtv = drv * (gcmd[0] * e[0] + gcmd[1] * e[1] + gcmd[2] * e[2]);      // track speed of a driving point (tv = 0 elsewhere)
    const S ge[3] = {tv * gcmd[0], tv * gcmd[1], tv * gcmd[2]};
    {      // |F_n|: zero gradient at F_n = 0
      const S s_ = gNn * (Nn > zero ? M::div(one, Nn) : zero);
#pragma unroll
      for (int c = 0; c < 3; ++c) gFr_[c] += s_ * Fr[c];
    }
    S gF1[3], d = zero, gA = zero;
#pragma unroll
    for (int c = 0; c < 3; ++c) { gF1[c] = inside(F1[c], -a.mg, a.mg) ? gFr_[c] : zero; d += gF1[c] * F0[c]; }
    const S gc_p = d * inv_csum;
    const S gS_p = -(d * cj) * inv_csum * inv_csum;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const S gF0 = gF1[c] * cj * inv_csum;
      gA += -(gF0 * nrm[c]);
      gn[c] += -A * gF0;
    }
    const S gvn = a.damp * gA;
#pragma unroll
    for (int c = 0; c < 3; ++c) { gvp[c] += gvn * nrm[c]; gn[c] += gvn * vp[c]; }
    // dh adjoint WITHOUT the share of gS (applied by the next iteration: dl = gS kappa)
    const S kappa = (S)-10 * cj * (one - cj);
    const S gdh = a.k * gA + gc_p * kappa;
    const S gzq = -gdh;
    const S dotn = gn[0] * nrm[0] + gn[1] * nrm[1] + gn[2] * nrm[2];
    const S gu0 = (gn[0] - dotn * nrm[0]) * inl, gu1 = (gn[1] - dotn * nrm[1]) * inl;   // n = u / |u|, u = (-gx, -gy, 1)
    const S ggx = M::div(-gu0, a.res), ggy = M::div(-gu1, a.res);
    const BlendW<S> bw = blend_weights(cell);
    const S w4[4] = {bw.w00, bw.w01, bw.w10, bw.w11};
    const S nz[4] = {gzq * w4[0] - ggx - ggy, gzq * w4[1] + ggx, gzq * w4[2] + ggy, gzq * w4[3]};
    int new_t = -1;
    if constexpr (TILE > 0) {
      const int tix = ix - (ocx - TILE / 2), tiy = iy - (ocy - TILE / 2);
      const bool inmap = (ix >= 0) & (ix < a.W - 1) & (iy >= 0) & (iy < a.H - 1);       // no fold, no wrap of the flat indices
      const bool inwin = ((unsigned)tix < (unsigned)(TILE - 1)) & ((unsigned)tiy < (unsigned)(TILE - 1));
      new_t = (inmap & inwin) ? tiy + TS * tix : -1;
    }
    {   // cell accumulators: same footprint as the previous iteration's -> add; else stash the old ones for the next flush
      // (after a re-centring the same footprint has another tile index: new_t != acc_t writes the -- zeroed -- accumulators out once)
      const bool same = !act | (((unsigned)cell.ic == acc_idx[0]) & ((unsigned)cell.ifl == acc_idx[3]) & (new_t == acc_t));
      const unsigned ni[4] = {(unsigned)cell.ic, (unsigned)cell.i_f, (unsigned)cell.il, (unsigned)cell.ifl};
      st_pending = !same;
      st_t = acc_t; acc_t = new_t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        st_idx[q] = acc_idx[q]; st_z[q] = acc_z[q]; st_m[q] = acc_m[q];
        acc_idx[q] = ni[q];
        acc_z[q] = (same ? acc_z[q] : zero) + nz[q];
        acc_m[q] = (same ? acc_m[q] : zero) + gmuq * w4[q];
      }
    }
    S zfx, zfy, mfx, mfy;
    blend_grad(cell, zc[0], zc[1], zc[2], zc[3], &zfx, &zfy);
    blend_grad(cell, mc[0], mc[1], mc[2], mc[3], &mfx, &mfy);
    // (a map of ones without a friction map: its blend is the sum of the weights, whose derivative is zero up to rounding)
    mfx = has_mu ? mfx : zero; mfy = has_mu ? mfy : zero;
    dk = kappa; dzx = zfx * a.inv_res; dzy = zfy * a.inv_res;
#pragma unroll
    for (int q = 0; q < 4; ++q) dw4[q] = w4[q];
    const S gp[3] = {M::div(gzq * zfx + gmuq * mfx, a.res), M::div(gzq * zfy + gmuq * mfy, a.res), gdh};
    {   // v_p = xd + w x r;  p = R P + x, r = p - x
      S qa[3] = {gp[0], gp[1], gp[2]};
      MF_CROSS_ACC(qa[0], qa[1], qa[2], f, gtau);        // tau += r x f : dr = f x gtau
      MF_CROSS_ACC(qa[0], qa[1], qa[2], gvp, w);         // dr += gvp x w
      MF_CROSS_ACC(lw[0], lw[1], lw[2], r, gvp);         // dw += r x gvp
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        lxd[q] += gvp[q];
        lx[q] += gp[q];
        lR[q * 3 + 0] += qa[q] * P[0];
        lR[q * 3 + 1] += qa[q] * P[1];
        lR[q * 3 + 2] += qa[q] * P[2];
      }
    }
    {   // e = col0(R) / max(|col0|, eps): through |col0| only when it is >= eps
      const S dote = (len2 >= (S)1e-12) ? ge[0] * e[0] + ge[1] * e[1] + ge[2] * e[2] : zero;
      lR[0] += (ge[0] - dote * e[0]) * il;
      lR[3] += (ge[1] - dote * e[1]) * il;
      lR[6] += (ge[2] - dote * e[2]) * il;
    }
    // ---- what the next iteration reads as totals ----
    if constexpr (DYN) { ex[0] = ex[1] = ex[2] = ex[3] = ex[4] = ex[5] = zero; }      // (dynamics(): summed in the middle of the step)
    else { ex[0] = lxd[0]; ex[1] = lxd[1]; ex[2] = lxd[2]; ex[3] = lw[0]; ex[4] = lw[1]; ex[5] = lw[2]; }
    ex[6] = gtv; ex[7] = tcw * gtv; ex[8] = gS_p;
#ifndef MF_MW_DBG_NOEXCHANGE   // A/B hook: the step without its exchange (wrong gradients, the instruction stream minus the sums)
    post9(ex);
#endif
  };

  StepIn sA, sB;
  UpIn uA, uB;
  load_step(max(n_steps - 1, 0), sA);
  load_up(DYN ? max(n_steps - 1, 0) : min(n_steps, a.T - 1), uA);
  {   // the exchange the first iteration fetches: nothing yet
#pragma unroll
    for (int k = 0; k < 9; ++k) ex[k] = zero;
    post9(ex);
  }
  __builtin_amdgcn_s_waitcnt(0);
  if constexpr (TILE > 0) {      // the window starts centred under the pose the scan starts from
    tile_clear();
    centre_cell(sA.x, ocx, ocy);
    tile_sync();
  }
  int n = n_steps - 1;
  if (n >= 0 && (n_steps & 1)) {      // an odd number of steps: one ahead of the pairs, leaving the next step's inputs in (sA, uA) again
    one_step(n, sA, uA, sB, uB);
    sA = sB; uA = uB;
    --n;
  }
  for (; n >= 1; n -= 2) {
    one_step(n, sA, uA, sB, uB);
    one_step(n - 1, sB, uB, sA, uA);
  }
  UpIn up;
  {   // the last exchange: control gradient of step 0 and its gS
    wait9(ex);
    if (a.gcontrols && n_steps > 0) { gctrl[0] = ex[6]; gctrl[1] = ex[7]; }
    const S dl = ex[8] * dk;
    const S g0 = -(dl * dzx), g1 = -(dl * dzy);
    lx[0] += g0; lx[1] += g1; lx[2] += dl;
#pragma unroll
    for (int c = 0; c < 3; ++c) { lR[0 * 3 + c] += g0 * P[c]; lR[1 * 3 + c] += g1 * P[c]; lR[2 * 3 + c] += dl * P[c]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc_z[q] -= dl * dw4[q];
  }
  flush_stash();
  if (act) emit_cells(acc_t, acc_idx, acc_z, acc_m);      // what is still accumulated in registers
  if constexpr (TILE > 0) {
    tile_sync();
    tile_flush();
  }
  if constexpr (!DYN) {      // output row 0 is the initial state itself (its forces are constant zeros)
    load_up(0, up);
    add_upstream_partials(up);
  }
  S tot[18];
#pragma unroll
  for (int c = 0; c < 3; ++c) { tot[c] = lx[c]; tot[3 + c] = lxd[c]; tot[6 + c] = lw[c]; }
#pragma unroll
  for (int c = 0; c < 9; ++c) tot[9 + c] = lR[c];
  gs.sum_n(tot);

  // terrain snap of the initial height: x.z = mean_i blend(z; cell((R0 P_i + x0).xy))   (dphysics.py:567-571)
  S gx0[3] = {tot[0], tot[1], tot[2]};
  if (!a.skip_snap) {
    S R0[9], x0[2];
#pragma unroll
    for (int c = 0; c < 9; ++c) R0[c] = a.R0[b * 9 + c];
    x0[0] = a.x_init[b * 3 + 0]; x0[1] = a.x_init[b * 3 + 1];
    const S g = tot[2] / (S)a.N;
    S sn6[8] = {zero, zero, zero, zero, zero, zero, zero, zero};   // gpx, gpy, gpx P, gpy P
    if (act) {
      const S px = P[0] * R0[0] + P[1] * R0[1] + P[2] * R0[2] + x0[0];
      const S py = P[0] * R0[3] + P[1] * R0[4] + P[2] * R0[5] + x0[1];
      const Cell<S> c = locate_m<S, kFast>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
      const S v0 = ld32(zmap, moff + (unsigned)c.ic), v1 = ld32(zmap, moff + (unsigned)c.i_f), v2 = ld32(zmap, moff + (unsigned)c.il), v3 = ld32(zmap, moff + (unsigned)c.ifl);
      atomic_add(at32(gzmap, goff + (unsigned)c.ic), g * (one - c.fx) * (one - c.fy));
      atomic_add(at32(gzmap, goff + (unsigned)c.i_f), g * (one - c.fx) * c.fy);
      atomic_add(at32(gzmap, goff + (unsigned)c.il), g * c.fx * (one - c.fy));
      atomic_add(at32(gzmap, goff + (unsigned)c.ifl), g * c.fx * c.fy);
      S dfx, dfy;
      blend_grad(c, v0, v1, v2, v3, &dfx, &dfy);
      const S gpx = M::div(g * dfx, a.res), gpy = M::div(g * dfy, a.res);
      sn6[0] = gpx; sn6[1] = gpy;
#pragma unroll
      for (int q = 0; q < 3; ++q) { sn6[2 + q] = gpx * P[q]; sn6[5 + q] = gpy * P[q]; }
    }
    gs.sum_n(sn6);
    gx0[0] += sn6[0];
    gx0[1] += sn6[1];
    gx0[2] = zero;                          // the caller's x0.z is overwritten, so nothing flows to it
#pragma unroll
    for (int q = 0; q < 6; ++q) tot[9 + q] += sn6[2 + q];
  }
  if (gl == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (a.gx0) a.gx0[b * 3 + c] = gx0[c];
      a.gxd0[b * 3 + c] = tot[3 + c];
      a.gw0[b * 3 + c] = tot[6 + c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) a.gR0[b * 9 + c] = tot[9 + c];
  }
}

// defined in rollout_bwd_mw_fast.hip
bool use_multiwave_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p);   // this backward goes to the kernels above
long long mw_record_bytes(const MfRolloutDesc* d, int scalar_bytes = 4);      // bytes of the record the forward keeps for them (0: none)
int launch_rollout_bwd_mw_f32(const RolloutBwdArgs<float>& a, int G, int integ, bool xs_only, hipStream_t st);
int launch_rollout_bwd_mw_f64(const RolloutBwdArgs<double>& a, int G, int integ, bool xs_only, hipStream_t st);      // rollout_mw_f64.hip
// cells per side of a rollout's LDS gradient tile; 0 = none.  Measured (B = 64 x N = 223: 1.094 -> 1.038 ms; 256 x 64: 0.861 -> 0.845;
// 1024 x 32: 0.971 -> 0.967; 256 x 16: 0.692 -> 0.883 -- four tiles per wave collide in the LDS): whole-wave groups only
constexpr int mw_tile_edge(int G) { return G >= 64 ? 64 : 0; }

// one launch of the instantiation the arguments call for (S = float: rollout_bwd_mw_fast.hip; S = double: rollout_mw_f64.hip)
template <typename S>
int launch_rollout_bwd_mw_t(const RolloutBwdArgs<S>& a, int G, int integ, bool xs_only, hipStream_t st) {
  // LDS gradient tiles (rollout_bwd_mw_kernel.h) while every workgroup of the launch is resident with its tiles: 160 KB per CU,
  // 256 CUs.  MF_MW_TILE=0 keeps the register accumulators (A/B runs, parity of the two routes).
  static const bool tile_off = getenv("MF_MW_TILE") && atoi(getenv("MF_MW_TILE")) == 0;
  bool launched = false;
#define MF_LAUNCH(G_, XS_, T_, I_) MF_KLAUNCH((rollout_bwd_mw_kernel<S, G_, XS_, T_, I_>), dim3(grid), dim3(blk), 0, st, a)
#define MF_CASE(G_)                                                                                              \
  if (!launched && G == G_) {                                                                                    \
    launched = true;                                                                                             \
    constexpr int blk = G_ > 64 ? G_ : 64;                                                                       \
    constexpr int TE = mw_tile_edge(G_);                                                                         \
    constexpr long long lds = (long long)(G_ > 64 ? 1 : 64 / G_) * 2 * (TE + 1) * TE * (long long)sizeof(S) + 4096;                       \
    const unsigned grid = (unsigned)(((long long)a.B * G_ + blk - 1) / blk);                                     \
    const bool tile = TE > 0 && !tile_off && (long long)((grid + (unsigned)device_cus() - 1) / (unsigned)device_cus()) * lds <= 160 * 1024 && (long long)a.H * a.W < (1ll << 30); \
    const bool dyn = integ == MF_INTEG_DYNAMICS;                                                                 \
    if (tile) {                                                                                                  \
      if constexpr (TE > 0) {                                                                                    \
        if (dyn) { if (xs_only) MF_LAUNCH(G_, true, TE, MF_INTEG_DYNAMICS); else MF_LAUNCH(G_, false, TE, MF_INTEG_DYNAMICS); }          \
        else     { if (xs_only) MF_LAUNCH(G_, true, TE, MF_INTEG_ODEINT_EULER); else MF_LAUNCH(G_, false, TE, MF_INTEG_ODEINT_EULER); }  \
      }                                                                                                          \
    } else {                                                                                                     \
      if (dyn) { if (xs_only) MF_LAUNCH(G_, true, 0, MF_INTEG_DYNAMICS); else MF_LAUNCH(G_, false, 0, MF_INTEG_DYNAMICS); }              \
      else     { if (xs_only) MF_LAUNCH(G_, true, 0, MF_INTEG_ODEINT_EULER); else MF_LAUNCH(G_, false, 0, MF_INTEG_ODEINT_EULER); }      \
    }                                                                                                            \
  }
  MF_CASE(8) MF_CASE(16) MF_CASE(32) MF_CASE(64) MF_CASE(128) MF_CASE(256) MF_CASE(512)
#undef MF_CASE
#undef MF_LAUNCH
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no multi-wave kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (multi-wave) launch: ") + hipGetErrorString(e));
  return MF_OK;
}


}  // namespace mf
