// Fused DPhysics rollout, forward pass (gfx950).
//
// One kernel runs the whole T-step scan of `DPhysics.dphysics()`
// (/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:530-594): per step it does the
// per-contact-point height/friction sample (`interpolate_grid`, :385-455, bug-for-bug), the spring-damper + friction
// contact forces and the rigid-body update (`forward_kinematics`, :172-272), then one Euler step of either integrator
// (`dynamics` :467-497 / `dynamics_odeint` :499-528), and streams the six API outputs.
//
// Mapping (wave64): a rollout is owned by a group of G consecutive lanes (G = 1 .. 64), each lane owning PPL = 4
// consecutive contact points (N = 4: one lane per rollout).  Several points per lane give the in-order SIMD
// independent instruction streams to interleave -- the step is a latency-bound dependent chain otherwise.  The 18-float rigid-body state is replicated in every lane of
// the group, so the only cross-lane traffic per step is two all-reduces (sum of contact weights; 9-component wrench),
// done on DPP for G <= 16.  The time axis is a dependent chain and stays serial inside the lane; parallelism is
// over rollouts (and points).  Map cells are gathered straight from global memory: both maps are read-only and at
// 256x256x4 B x 2 = 512 KiB they live in every XCD's 4 MiB L2, with the few cells under a slowly moving robot
// (<= 0.2 cell per step) staying in the CU's L1 -- see DESIGN.md for why a per-workgroup LDS tile does not pay at N=4.
// Outputs are written time-major by default so a wave's stores of one step form contiguous segments.
#include "rollout_common.h"

namespace mf {

template <typename S>
struct RolloutArgs {
  int B, T, N, H, W, n_tracks, layout, map_shared, skip_snap;
  S mass, mg, k, damp, omega_max, res, d_max, dt, half_ly, sink;
  S Iinv[9];
  const S* z;
  const S* mu;
  const S* controls;
  const S* ts;
  const S* points;
  const int* part;
  S* x0;
  const S* xd0;
  const S* R0;
  const S* w0;
  S* Xs;
  S* Xds;
  S* Rs;
  S* Om;
  S* Fs;
  S* Ff;
  S* Xraw;
};

template <typename S, int G, int PPL, int INTEG>
__global__ void __launch_bounds__(256) rollout_fwd_kernel(const RolloutArgs<S> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid / G;
  const int gl = tid % G;
  if (b >= a.B) return;  // whole groups leave together; live groups never read dead lanes
  const S one = (S)1, zero = (S)0;
  const int HW = a.H * a.W, last = HW - 1;
  const S* zmap = a.z + (a.map_shared ? 0 : (size_t)b * HW);
  const S* mumap = a.mu ? a.mu + (a.map_shared ? 0 : (size_t)b * HW) : nullptr;

  // this lane's contact points
  S P[PPL][3];
  int part[PPL];
  bool act[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int i = gl * PPL + j;       // blocked: a lane owns PPL consecutive points (contiguous force rows per lane)
    act[j] = i < a.N;
    int ii = act[j] ? i : 0;
    P[j][0] = a.points[ii * 3 + 0];
    P[j][1] = a.points[ii * 3 + 1];
    P[j][2] = a.points[ii * 3 + 2];
    part[j] = act[j] ? a.part[ii] : -1;
  }

  // state, replicated across the group
  S x[3], xd[3], R[9], w[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    x[c] = a.x0[b * 3 + c];
    xd[c] = a.xd0[b * 3 + c];
    w[c] = a.w0[b * 3 + c];
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) R[c] = a.R0[b * 9 + c];

  // start at the terrain height: x.z <- mean_i interp(z, (P R^T + x)_i)   (dphysics.py:567-571)
  if (!a.skip_snap) {
    S acc = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      Cell<S> c = locate(px, py, a.d_max, a.res, a.H, last);
      S v = blend(c, zmap[c.ic], zmap[c.i_f], zmap[c.il], zmap[c.ifl]);
      acc += act[j] ? v : zero;
    }
    acc = group_sum<G>(acc);
    x[2] = acc / (S)a.N;
    if (gl == 0) a.x0[b * 3 + 2] = x[2];
  }

  const size_t row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)a.B : 1;  // rows between consecutive t
  const size_t row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)b : (size_t)b * a.T;

  S accFs[PPL][3], accFf[PPL][3];  // ODEINT: running impulses (extended state, dphysics.py:506-509)
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c) accFs[j][c] = accFf[j][c] = zero;

  // Stores of one output row, by lane 0 of the group (G == 1: every lane): no per-lane value selection on the
  // dependent issue stream; a wave's 64/G rows are contiguous in the time-major layout.
  auto emit_state = [&](size_t row) {
    if (G == 1 || gl == 0) {
      S* p = a.Xs + row * 3;
      p[0] = x[0] + R[2] * a.sink;  // Xs += Rs[..., :, 2] * m g / (k + 1e-6)   (dphysics.py:587-589)
      p[1] = x[1] + R[5] * a.sink;
      p[2] = x[2] + R[8] * a.sink;
      S* q = a.Xds + row * 3;
      q[0] = xd[0]; q[1] = xd[1]; q[2] = xd[2];
      S* o = a.Om + row * 3;
      o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
      S* rr = a.Rs + row * 9;
#pragma unroll
      for (int c = 0; c < 9; ++c) rr[c] = R[c];
      if (a.Xraw) {
        S* r = a.Xraw + row * 3;
        r[0] = x[0]; r[1] = x[1]; r[2] = x[2];
      }
    }
  };
  auto emit_forces = [&](size_t row, const S (*fs)[3], const S (*ff)[3]) {
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      if (act[j]) {
        size_t o = (row * a.N + (gl * PPL + j)) * 3;
        a.Fs[o + 0] = fs[j][0]; a.Fs[o + 1] = fs[j][1]; a.Fs[o + 2] = fs[j][2];
        a.Ff[o + 0] = ff[j][0]; a.Ff[o + 1] = ff[j][1]; a.Ff[o + 2] = ff[j][2];
      }
    }
  };

  int n_steps = a.T;
  if (INTEG == MF_INTEG_ODEINT_EULER) {
    emit_state(row0);           // y_0
    emit_forces(row0, accFs, accFf);
    n_steps = a.T - 1;
  }

  const S* ctrl = a.controls + (size_t)b * a.T * 2;
  S cv = ctrl[0], cw = ctrl[1];

  for (int n = 0; n < n_steps; ++n) {
    // prefetch the next step's controls (the lookup argmin|t - ts| is the step index on the grid, dphysics.py:183)
    const int nn = min(n + 1, a.T - 1);
    const S cv_next = ctrl[nn * 2 + 0], cw_next = ctrl[nn * 2 + 1];

    S r[PPL][3], vp[PPL][3], nrm[PPL][3], muq[PPL], cw8[PPL], Fr[PPL][3];
    S csum = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      // p = P R^T + x ; r = p - x   (:200)
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      S pz = P[j][0] * R[6] + P[j][1] * R[7] + P[j][2] * R[8] + x[2];
      r[j][0] = px - x[0]; r[j][1] = py - x[1]; r[j][2] = pz - x[2];
      // v_p = xd + w x r   (:204)
      vp[j][0] = xd[0] + (w[1] * r[j][2] - w[2] * r[j][1]);
      vp[j][1] = xd[1] + (w[2] * r[j][0] - w[0] * r[j][2]);
      vp[j][2] = xd[2] + (w[0] * r[j][1] - w[1] * r[j][0]);
      // height, normal, friction under the point   (:211-216)
      Cell<S> c = locate(px, py, a.d_max, a.res, a.H, last);
      S zc = zmap[c.ic], zf = zmap[c.i_f], zl = zmap[c.il], zfl = zmap[c.ifl];
      S mc = one, mf_ = one, ml = one, mfl = one;
      if (mumap) { mc = mumap[c.ic]; mf_ = mumap[c.i_f]; ml = mumap[c.il]; mfl = mumap[c.ifl]; }
      S zq = blend(c, zc, zf, zl, zfl);
      muq[j] = blend(c, mc, mf_, ml, mfl);
      S gx = (zf - zc) / a.res, gy = (zl - zc) / a.res;
      S nl = mf_max(mf_sqrt(gx * gx + gy * gy + one), (S)1e-6);
      nrm[j][0] = -gx / nl; nrm[j][1] = -gy / nl; nrm[j][2] = one / nl;
      // soft contact + spring-damper along the normal   (:220-230)
      S dh = pz - zq;
      S cj = one / (one + mf_exp((S)10 * dh));  // sigmoid(-10 dh)
      cj = act[j] ? cj : zero;
      cw8[j] = cj;
      csum += cj;
      S vn = vp[j][0] * nrm[j][0] + vp[j][1] * nrm[j][1] + vp[j][2] * nrm[j][2];
      S A = a.k * dh + a.damp * vn;
      Fr[j][0] = -(A * nrm[j][0]); Fr[j][1] = -(A * nrm[j][1]); Fr[j][2] = -(A * nrm[j][2]);
    }
    csum = group_sum<G>(csum);  // n_contact_pts (:231)

    // thrust direction = normalized first column of R   (:237)
    S el = mf_max(mf_sqrt(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]), (S)1e-6);
    S e0 = R[0] / el, e1 = R[3] / el, e2 = R[6] / el;
    S tv_lo = cv - cw * a.half_ly, tv_hi = cv + cw * a.half_ly;  // (:75-104)

    S sFr[3] = {zero, zero, zero}, sFf[3] = {zero, zero, zero}, sTau[3] = {zero, zero, zero};
    S Ff[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Fr[j][c] = mf_clamp(Fr[j][c] * cw8[j] / csum, -a.mg, a.mg);  // (:232-233)
      S Nn = mf_sqrt(Fr[j][0] * Fr[j][0] + Fr[j][1] * Fr[j][1] + Fr[j][2] * Fr[j][2]);          // (:238)
      S tv = (part[j] < 0) ? zero : ((part[j] & 1) ? tv_hi : tv_lo);
      S s0 = muq[j] * (tv * e0 - vp[j][0]);  // slip (:247); cmd = 0 for non-driving points
      S s1 = muq[j] * (tv * e1 - vp[j][1]);
      S s2 = muq[j] * (tv * e2 - vp[j][2]);
      S sn = s0 * nrm[j][0] + s1 * nrm[j][1] + s2 * nrm[j][2];
      Ff[j][0] = mf_clamp(Nn * (s0 - sn * nrm[j][0]), -a.mg, a.mg);  // (:248-251)
      Ff[j][1] = mf_clamp(Nn * (s1 - sn * nrm[j][1]), -a.mg, a.mg);
      Ff[j][2] = mf_clamp(Nn * (s2 - sn * nrm[j][2]), -a.mg, a.mg);
      if (!act[j]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Fr[j][c] = Ff[j][c] = zero;
      }
      S f0 = Fr[j][0] + Ff[j][0], f1 = Fr[j][1] + Ff[j][1], f2 = Fr[j][2] + Ff[j][2];
      sTau[0] += r[j][1] * f2 - r[j][2] * f1;  // r x (Fs + Ff)   (:255)
      sTau[1] += r[j][2] * f0 - r[j][0] * f2;
      sTau[2] += r[j][0] * f1 - r[j][1] * f0;
#pragma unroll
      for (int c = 0; c < 3; ++c) { sFr[c] += Fr[j][c]; sFf[c] += Ff[j][c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sFr[c] = group_sum<G>(sFr[c]);
      sFf[c] = group_sum<G>(sFf[c]);
      sTau[c] = group_sum<G>(sTau[c]);
    }
    // omega_d = clamp(I^-1 tau) (body-frame I with world-frame torque, as the reference)   (:256-257)
    S wd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      wd[c] = mf_clamp(a.Iinv[c * 3 + 0] * sTau[0] + a.Iinv[c * 3 + 1] * sTau[1] + a.Iinv[c * 3 + 2] * sTau[2],
                       -a.omega_max, a.omega_max);
    // xdd = (m g ghat + sum Fs + sum Ff) / m   (:264-266)
    S xdd[3] = {(sFr[0] + sFf[0]) / a.mass, (sFr[1] + sFf[1]) / a.mass, ((-a.mg + sFr[2]) + sFf[2]) / a.mass};

    const size_t row = row0 + (size_t)(INTEG == MF_INTEG_ODEINT_EULER ? n + 1 : n) * row_stride;
    if (INTEG == MF_INTEG_DYNAMICS) {
      // update_state (:274-288): xd += xdd h ; x += xd_new h ; w += wd h ; R <- R (I + K sin + K^2 (1 - cos))
      const S h = a.dt;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xd[c] = xd[c] + xdd[c] * h;
        x[c] = x[c] + xd[c] * h;
        w[c] = w[c] + wd[c] * h;
      }
      S th = mf_sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      S den = mf_max(th, (S)1e-6);
      S k0 = w[0] / den, k1 = w[1] / den, k2 = w[2] / den;  // K = [w]x / max(|w|, eps)
      S sn, cs;
      mf_sincos(th * h, &sn, &cs);
      S oc = one - cs;
      // K = [[0,-k2,k1],[k2,0,-k0],[-k1,k0,0]];  K^2 = k k^T - |k|^2 I (computed as the explicit product)
      S K[9] = {zero, -k2, k1, k2, zero, -k0, -k1, k0, zero};
      S M[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          S kk = K[i * 3 + 0] * K[0 * 3 + j2] + K[i * 3 + 1] * K[1 * 3 + j2] + K[i * 3 + 2] * K[2 * 3 + j2];
          M[i * 3 + j2] = ((i == j2 ? one : zero) + K[i * 3 + j2] * sn) + kk * oc;
        }
      S Rn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2)
          Rn[i * 3 + j2] = R[i * 3 + 0] * M[0 * 3 + j2] + R[i * 3 + 1] * M[1 * 3 + j2] + R[i * 3 + 2] * M[2 * 3 + j2];
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = Rn[c];
      emit_state(row);
      emit_forces(row, Fr, Ff);
    } else {
      // torchdiffeq fixed-grid euler: y_{n+1} = y_n + (t_{n+1} - t_n) f(t_n, y_n), f = (xd, xdd, [w]x R, wd, Fs, Ff)
      const S h = a.ts[n + 1] - a.ts[n];
      S dR[9];
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2) {
        dR[0 * 3 + j2] = w[1] * R[2 * 3 + j2] - w[2] * R[1 * 3 + j2];
        dR[1 * 3 + j2] = w[2] * R[0 * 3 + j2] - w[0] * R[2 * 3 + j2];
        dR[2 * 3 + j2] = w[0] * R[1 * 3 + j2] - w[1] * R[0 * 3 + j2];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[c] = x[c] + h * xd[c];  // OLD xd moves x
        xd[c] = xd[c] + h * xdd[c];
        w[c] = w[c] + h * wd[c];
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = R[c] + h * dR[c];
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          accFs[j][c] = accFs[j][c] + h * Fr[j][c];
          accFf[j][c] = accFf[j][c] + h * Ff[j][c];
        }
      emit_state(row);
      emit_forces(row, accFs, accFf);
    }
    cv = cv_next; cw = cw_next;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
template <typename S, int G, int PPL>
static int launch_gp(const RolloutArgs<S>& a, int integ, int block, hipStream_t st) {
  const long long threads = (long long)a.B * G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  if (integ == MF_INTEG_DYNAMICS)
    hipLaunchKernelGGL((rollout_fwd_kernel<S, G, PPL, MF_INTEG_DYNAMICS>), dim3(grid), dim3(block), 0, st, a);
  else
    hipLaunchKernelGGL((rollout_fwd_kernel<S, G, PPL, MF_INTEG_ODEINT_EULER>), dim3(grid), dim3(block), 0, st, a);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

template <typename S>
int rollout_fwd(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, void* stream) {
  MF_REQUIRE(d && p, MF_ERR_INVALID, "rollout_fwd: null descriptor");
  MF_REQUIRE(d->B > 0 && d->N > 0 && d->H > 1 && d->W > 0, MF_ERR_INVALID, "rollout_fwd: B, N, H, W must be positive");
  MF_REQUIRE(d->T >= 1, MF_ERR_INVALID, "rollout_fwd: T must be >= 1");
  MF_REQUIRE(d->n_tracks == 2 || d->n_tracks == 4, MF_ERR_INVALID, "n_tracks must be 2 or 4");
  MF_REQUIRE(d->integrator == MF_INTEG_DYNAMICS || d->integrator == MF_INTEG_ODEINT_EULER, MF_ERR_INVALID,
             "rollout_fwd: unknown integrator");
  MF_REQUIRE(d->layout == MF_LAYOUT_BATCH_MAJOR || d->layout == MF_LAYOUT_TIME_MAJOR, MF_ERR_INVALID,
             "rollout_fwd: unknown layout");
  MF_REQUIRE(p->z && p->controls && p->ts && p->points && p->part && p->x0 && p->xd0 && p->R0 && p->w0, MF_ERR_INVALID,
             "rollout_fwd: null input buffer");
  MF_REQUIRE(p->Xs && p->Xds && p->Rs && p->Omegas && p->Fs && p->Ff, MF_ERR_INVALID, "rollout_fwd: null output buffer");
  MF_REQUIRE((long long)d->H * d->W < (1ll << 30), MF_ERR_UNSUPPORTED, "rollout_fwd: grid too large");
  MF_REQUIRE(d->N <= 512, MF_ERR_UNSUPPORTED, "rollout_fwd: more than 512 contact points");
  int block = d->block ? d->block : 64;
  MF_REQUIRE(block == 64 || block == 128 || block == 256, MF_ERR_INVALID, "rollout_fwd: block must be 64, 128 or 256");

  RolloutArgs<S> a;
  a.B = d->B; a.T = d->T; a.N = d->N; a.H = d->H; a.W = d->W;
  a.n_tracks = d->n_tracks; a.layout = d->layout; a.map_shared = d->map_shared; a.skip_snap = d->skip_snap;
  a.mass = (S)d->mass; a.mg = (S)(d->mass * d->gravity); a.k = (S)d->stiffness; a.damp = (S)d->damping;
  a.omega_max = (S)d->omega_max; a.res = (S)d->grid_res; a.d_max = (S)d->d_max; a.dt = (S)d->dt;
  a.half_ly = (S)(d->robot_size_y / 2.0);
  a.sink = (S)(d->mass * d->gravity / (d->stiffness + 1e-6));
  for (int i = 0; i < 9; ++i) a.Iinv[i] = (S)d->Iinv[i];
  a.z = (const S*)p->z; a.mu = (const S*)p->mu; a.controls = (const S*)p->controls; a.ts = (const S*)p->ts;
  a.points = (const S*)p->points; a.part = p->part;
  a.x0 = (S*)p->x0; a.xd0 = (const S*)p->xd0; a.R0 = (const S*)p->R0; a.w0 = (const S*)p->w0;
  a.Xs = (S*)p->Xs; a.Xds = (S*)p->Xds; a.Rs = (S*)p->Rs; a.Om = (S*)p->Omegas; a.Fs = (S*)p->Fs; a.Ff = (S*)p->Ff;
  a.Xraw = (S*)p->Xraw;

  hipStream_t st = (hipStream_t)stream;
  const int N = d->N, integ = d->integrator;
  // Lane mapping: G lanes per rollout x PPL points per lane.  A single wave issues roughly one instruction per
  // 4-5 cycles whatever the dependences, so while the launch has few waves per SIMD (latency-bound, e.g. B = 1024, N = 4)
  // one point per lane minimises the instructions a wave must issue per step; once the chip is full the redundant
  // per-lane state update of that mapping costs throughput and 4 points per lane wins (measured crossover ~4 waves/SIMD).
  int g1 = 4;
  while (g1 < N) g1 <<= 1;                                   // lanes per rollout at one point per lane
  bool wide = g1 <= 64 && (long long)a.B * g1 / 64 <= 4096;
  if (d->points_per_lane == 1 && g1 <= 64) wide = true;
  if (d->points_per_lane == 4) wide = false;
#define MF_GO(G_, P_) return launch_gp<S, G_, P_>(a, integ, block, st)
  if (N <= 4) { if (wide) MF_GO(4, 1); MF_GO(1, 4); }
  if (N <= 8) { if (wide) MF_GO(8, 1); MF_GO(2, 4); }
  if (N <= 16) { if (wide) MF_GO(16, 1); MF_GO(4, 4); }
  if (N <= 32) { if (wide) MF_GO(32, 1); MF_GO(8, 4); }
  if (N <= 64) { if (wide) MF_GO(64, 1); MF_GO(16, 4); }
  if (N <= 128) { if (d->points_per_lane != 4 && (long long)a.B * 2 <= 4096) MF_GO(64, 2); MF_GO(32, 4); }
  if (N <= 256) MF_GO(64, 4);
  MF_GO(64, 8);
#undef MF_GO
}

}  // namespace mf

extern "C" int mf_rollout_fwd_f32(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, void* s) {
  return mf::rollout_fwd<float>(d, p, s);
}
extern "C" int mf_rollout_fwd_f64(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, void* s) {
  return mf::rollout_fwd<double>(d, p, s);
}
