// Fused DPhysics rollout, backward pass (gfx950): reverse-time adjoint of rollout_fwd.hip.
//
// Replaces the autograd graph the reference builds through `forward_kinematics` / `dynamics` / `dynamics_odeint`
// (/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:172-272, :467-528; T x ~300 nodes) with one
// kernel: walk the steps backwards, reload the state each step started from (they are the forward's own outputs, so
// checkpoints are free), recompute that step's intermediates, and apply the hand-derived vector-Jacobian product.
// Same lane mapping as the forward (G lanes per rollout, PPL points per lane, adjoint state replicated across the group,
// cross-lane sums on DPP).  Gradients w.r.t. the height / friction cells are scattered with hardware float atomics.
//
// Autograd conventions reproduced: clamp passes gradient iff lo <= x <= hi; `.long()` cell indices are
// constants (queries influence samples only through the fractions); |v| has zero gradient at v = 0;
// x / clamp(|x|, eps) differentiates through |x| only when |x| >= eps.
#include "rollout_common.h"

namespace mf {

template <typename S>
struct RolloutBwdArgs {
  int B, T, N, H, W, n_tracks, layout, map_shared, skip_snap;
  S mass, mg, k, damp, omega_max, res, d_max, dt, half_ly, sink;
  S Iinv[9];
  const S *z, *mu, *controls, *ts, *points;
  const int* part;
  const S *x_init, *xd0, *R0, *w0;
  const S *Xraw, *Xds, *Rs, *Om;
  const S *gXs, *gXds, *gRs, *gOm, *gFs, *gFf;
  S *gz, *gmu, *gcontrols, *gx0, *gxd0, *gR0, *gw0;
};

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

template <typename S>
__device__ __forceinline__ bool inside(S v, S lo, S hi) { return v >= lo && v <= hi; }

#define MF_CROSS(o, a, b)                    \
  do {                                       \
    (o)[0] = (a)[1] * (b)[2] - (a)[2] * (b)[1]; \
    (o)[1] = (a)[2] * (b)[0] - (a)[0] * (b)[2]; \
    (o)[2] = (a)[0] * (b)[1] - (a)[1] * (b)[0]; \
  } while (0)

template <typename S, int G, int PPL, int INTEG>
__global__ void __launch_bounds__(256) rollout_bwd_kernel(const RolloutBwdArgs<S> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid / G;
  const int gl = tid % G;
  if (b >= a.B) return;
  const S one = (S)1, zero = (S)0;
  const int HW = a.H * a.W, last = HW - 1;
  const size_t map_off = a.map_shared ? 0 : (size_t)b * HW;
  const S* zmap = a.z + map_off;
  const S* mumap = a.mu ? a.mu + map_off : nullptr;
  S* gzmap = a.gz + map_off;
  S* gmumap = (a.gmu && a.mu) ? a.gmu + map_off : nullptr;

  S P[PPL][3];
  int part[PPL];
  bool act[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int i = gl * PPL + j;       // blocked, as in the forward
    act[j] = i < a.N;
    int ii = act[j] ? i : 0;
    P[j][0] = a.points[ii * 3 + 0];
    P[j][1] = a.points[ii * 3 + 1];
    P[j][2] = a.points[ii * 3 + 2];
    part[j] = act[j] ? a.part[ii] : -1;
  }

  const size_t row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)a.B : 1;
  const size_t row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)b : (size_t)b * a.T;
  const S* ctrl = a.controls + (size_t)b * a.T * 2;
  S* gctrl = a.gcontrols + (size_t)b * a.T * 2;

  // adjoint of the state (x, xd, R, w) [+ the impulse accumulators of the ODEINT extended state]
  S lx[3] = {zero, zero, zero}, lxd[3] = {zero, zero, zero}, lw[3] = {zero, zero, zero}, lR[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) lR[c] = zero;
  S laFs[PPL][3], laFf[PPL][3];
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c) laFs[j][c] = laFf[j][c] = zero;

  auto add_upstream_state = [&](size_t row) {
    if (a.gXs) {
      const S* g = a.gXs + row * 3;
      S g0 = g[0], g1 = g[1], g2 = g[2];
      lx[0] += g0; lx[1] += g1; lx[2] += g2;
      lR[2] += g0 * a.sink; lR[5] += g1 * a.sink; lR[8] += g2 * a.sink;   // Xs = x + R[:,2] * sink
    }
    if (a.gXds) {
      const S* g = a.gXds + row * 3;
      lxd[0] += g[0]; lxd[1] += g[1]; lxd[2] += g[2];
    }
    if (a.gRs) {
      const S* g = a.gRs + row * 9;
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] += g[c];
    }
    if (a.gOm) {
      const S* g = a.gOm + row * 3;
      lw[0] += g[0]; lw[1] += g[1]; lw[2] += g[2];
    }
  };

  const int n_steps = (INTEG == MF_INTEG_ODEINT_EULER) ? a.T - 1 : a.T;
  if (INTEG == MF_INTEG_ODEINT_EULER && n_steps < a.T - 0) {
    // the last control of the grid is never used by the explicit scheme
    if (gl == 0) { gctrl[(a.T - 1) * 2 + 0] = zero; gctrl[(a.T - 1) * 2 + 1] = zero; }
  }

  for (int n = n_steps - 1; n >= 0; --n) {
    const size_t out_row = row0 + (size_t)(INTEG == MF_INTEG_ODEINT_EULER ? n + 1 : n) * row_stride;
    add_upstream_state(out_row);

    // state this step started from
    S x[3], xd[3], R[9], w[3];
    if (INTEG == MF_INTEG_DYNAMICS && n == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = a.x_init[b * 3 + c]; xd[c] = a.xd0[b * 3 + c]; w[c] = a.w0[b * 3 + c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = a.R0[b * 9 + c];
    } else {
      const size_t in_row = row0 + (size_t)(INTEG == MF_INTEG_ODEINT_EULER ? n : n - 1) * row_stride;
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = a.Xraw[in_row * 3 + c]; xd[c] = a.Xds[in_row * 3 + c]; w[c] = a.Om[in_row * 3 + c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = a.Rs[in_row * 9 + c];
    }
    const S cv = ctrl[n * 2 + 0], cw = ctrl[n * 2 + 1];

    // ---------------------------------------------------------------------------------------------------
    // forward recompute (identical arithmetic to rollout_fwd.hip)
    // ---------------------------------------------------------------------------------------------------
    Cell<S> cell[PPL];
    S zc4[PPL][4], mc4[PPL][4];
    S r[PPL][3], vp[PPL][3], nrm[PPL][3], nl[PPL], muq[PPL], cj[PPL], Aj[PPL], F0[PPL][3];
    S csum = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      S pz = P[j][0] * R[6] + P[j][1] * R[7] + P[j][2] * R[8] + x[2];
      r[j][0] = px - x[0]; r[j][1] = py - x[1]; r[j][2] = pz - x[2];
      vp[j][0] = xd[0] + (w[1] * r[j][2] - w[2] * r[j][1]);
      vp[j][1] = xd[1] + (w[2] * r[j][0] - w[0] * r[j][2]);
      vp[j][2] = xd[2] + (w[0] * r[j][1] - w[1] * r[j][0]);
      cell[j] = locate(px, py, a.d_max, a.res, a.H, last);
      const Cell<S>& c = cell[j];
      zc4[j][0] = zmap[c.ic]; zc4[j][1] = zmap[c.i_f]; zc4[j][2] = zmap[c.il]; zc4[j][3] = zmap[c.ifl];
      if (mumap) { mc4[j][0] = mumap[c.ic]; mc4[j][1] = mumap[c.i_f]; mc4[j][2] = mumap[c.il]; mc4[j][3] = mumap[c.ifl]; }
      else { mc4[j][0] = mc4[j][1] = mc4[j][2] = mc4[j][3] = one; }
      S zq = blend(c, zc4[j][0], zc4[j][1], zc4[j][2], zc4[j][3]);
      muq[j] = blend(c, mc4[j][0], mc4[j][1], mc4[j][2], mc4[j][3]);
      S gx = (zc4[j][1] - zc4[j][0]) / a.res, gy = (zc4[j][2] - zc4[j][0]) / a.res;
      nl[j] = mf_max(mf_sqrt(gx * gx + gy * gy + one), (S)1e-6);
      nrm[j][0] = -gx / nl[j]; nrm[j][1] = -gy / nl[j]; nrm[j][2] = one / nl[j];
      S dh = pz - zq;
      S cc = one / (one + mf_exp((S)10 * dh));
      cj[j] = act[j] ? cc : zero;
      csum += cj[j];
      S vn = vp[j][0] * nrm[j][0] + vp[j][1] * nrm[j][1] + vp[j][2] * nrm[j][2];
      Aj[j] = a.k * dh + a.damp * vn;
      F0[j][0] = -(Aj[j] * nrm[j][0]); F0[j][1] = -(Aj[j] * nrm[j][1]); F0[j][2] = -(Aj[j] * nrm[j][2]);
    }
    csum = group_sum<G>(csum);

    const S coln = mf_sqrt(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]);
    const S el = mf_max(coln, (S)1e-6);
    const S e[3] = {R[0] / el, R[3] / el, R[6] / el};
    const S tv_lo = cv - cw * a.half_ly, tv_hi = cv + cw * a.half_ly;

    S F1[PPL][3], Fr[PPL][3], Ff[PPL][3], Gf[PPL][3], st[PPL][3], slip[PPL][3], sn[PPL], Nn[PPL], tv[PPL];
    S sTau[3] = {zero, zero, zero};
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        F1[j][c] = F0[j][c] * cj[j] / csum;
        Fr[j][c] = mf_clamp(F1[j][c], -a.mg, a.mg);
      }
      Nn[j] = mf_sqrt(Fr[j][0] * Fr[j][0] + Fr[j][1] * Fr[j][1] + Fr[j][2] * Fr[j][2]);
      tv[j] = (part[j] < 0) ? zero : ((part[j] & 1) ? tv_hi : tv_lo);
#pragma unroll
      for (int c = 0; c < 3; ++c) slip[j][c] = muq[j] * (tv[j] * e[c] - vp[j][c]);
      sn[j] = slip[j][0] * nrm[j][0] + slip[j][1] * nrm[j][1] + slip[j][2] * nrm[j][2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        st[j][c] = slip[j][c] - sn[j] * nrm[j][c];
        Gf[j][c] = Nn[j] * st[j][c];
        Ff[j][c] = mf_clamp(Gf[j][c], -a.mg, a.mg);
      }
      if (!act[j]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Fr[j][c] = Ff[j][c] = zero;
      }
      S f[3] = {Fr[j][0] + Ff[j][0], Fr[j][1] + Ff[j][1], Fr[j][2] + Ff[j][2]};
      sTau[0] += r[j][1] * f[2] - r[j][2] * f[1];
      sTau[1] += r[j][2] * f[0] - r[j][0] * f[2];
      sTau[2] += r[j][0] * f[1] - r[j][1] * f[0];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) sTau[c] = group_sum<G>(sTau[c]);
    S wraw[3], wd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      wraw[c] = a.Iinv[c * 3 + 0] * sTau[0] + a.Iinv[c * 3 + 1] * sTau[1] + a.Iinv[c * 3 + 2] * sTau[2];
      wd[c] = mf_clamp(wraw[c], -a.omega_max, a.omega_max);
    }

    // ---------------------------------------------------------------------------------------------------
    // integrator backward: adjoint of the step's outputs -> (g_xdd, g_wd, g_Fs_i, g_Ff_i) + adjoint of its inputs
    // ---------------------------------------------------------------------------------------------------
    S gxdd[3], gwd[3], gFr[PPL][3], gFf[PPL][3];
    if (INTEG == MF_INTEG_ODEINT_EULER) {
      const S h = a.ts[n + 1] - a.ts[n];
      if (a.gFs) {
#pragma unroll
        for (int j = 0; j < PPL; ++j)
          if (act[j]) {
            const size_t o = (out_row * a.N + (gl * PPL + j)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) laFs[j][c] += a.gFs[o + c];
          }
      }
      if (a.gFf) {
#pragma unroll
        for (int j = 0; j < PPL; ++j)
          if (act[j]) {
            const size_t o = (out_row * a.N + (gl * PPL + j)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) laFf[j][c] += a.gFf[o + c];
          }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gxdd[c] = h * lxd[c];
        gwd[c] = h * lw[c];
        lxd[c] += h * lx[c];               // x' = x + h xd
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { gFr[j][c] = h * laFs[j][c]; gFf[j][c] = h * laFf[j][c]; }
      // R' = R + h [w]x R : column-wise dR_c = w x R_c
      S lRn[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        S gcol[3] = {h * lR[0 * 3 + c], h * lR[1 * 3 + c], h * lR[2 * 3 + c]};
        S rc[3] = {R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]};
        S t1[3], t2[3];
        MF_CROSS(t1, rc, gcol);            // d/dw of (w x R_c) . g  =  R_c x g
        lw[0] += t1[0]; lw[1] += t1[1]; lw[2] += t1[2];
        MF_CROSS(t2, gcol, w);             // d/dR_c                  =  g x w
        lRn[0 * 3 + c] = lR[0 * 3 + c] + t2[0];
        lRn[1 * 3 + c] = lR[1 * 3 + c] + t2[1];
        lRn[2 * 3 + c] = lR[2 * 3 + c] + t2[2];
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] = lRn[c];
    } else {
      const S h = a.dt;
      // forces of this step are outputs themselves
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const size_t o = (out_row * a.N + (gl * PPL + j)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gFr[j][c] = (a.gFs && act[j]) ? a.gFs[o + c] : zero;
          gFf[j][c] = (a.gFf && act[j]) ? a.gFf[o + c] : zero;
        }
      }
      // R' = R M(w'),  w' = w + wd h,  M = I + K sin(th h) + K^2 (1 - cos(th h)),  K = [w']x / max(|w'|, eps)
      S wn[3] = {w[0] + wd[0] * h, w[1] + wd[1] * h, w[2] + wd[2] * h};
      S th = mf_sqrt(wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]);
      S den = mf_max(th, (S)1e-6);
      S kv[3] = {wn[0] / den, wn[1] / den, wn[2] / den};
      S sn_, cs_;
      mf_sincos(th * h, &sn_, &cs_);
      S oc = one - cs_;
      S K[9] = {zero, -kv[2], kv[1], kv[2], zero, -kv[0], -kv[1], kv[0], zero};
      S K2[9], M[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          K2[i * 3 + j2] = K[i * 3 + 0] * K[0 * 3 + j2] + K[i * 3 + 1] * K[1 * 3 + j2] + K[i * 3 + 2] * K[2 * 3 + j2];
          M[i * 3 + j2] = ((i == j2 ? one : zero) + K[i * 3 + j2] * sn_) + K2[i * 3 + j2] * oc;
        }
      S gM[9], lRn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          gM[i * 3 + j2] = R[0 * 3 + i] * lR[0 * 3 + j2] + R[1 * 3 + i] * lR[1 * 3 + j2] + R[2 * 3 + i] * lR[2 * 3 + j2];   // R^T lR
          lRn[i * 3 + j2] = lR[i * 3 + 0] * M[j2 * 3 + 0] + lR[i * 3 + 1] * M[j2 * 3 + 1] + lR[i * 3 + 2] * M[j2 * 3 + 2];  // lR M^T
        }
      S ga = zero, gb = zero, gK[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) { ga += gM[c] * K[c]; gb += gM[c] * K2[c]; }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          // d(K K) -> gM K^T + K^T gM
          S t = zero;
#pragma unroll
          for (int m = 0; m < 3; ++m) t += gM[i * 3 + m] * K[j2 * 3 + m] + K[m * 3 + i] * gM[m * 3 + j2];
          gK[i * 3 + j2] = sn_ * gM[i * 3 + j2] + oc * t;
        }
      S gk[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
      S gth = ga * h * cs_ + gb * h * sn_;
      S gwn[3] = {gk[0] / den, gk[1] / den, gk[2] / den};
      if (th >= (S)1e-6) gth += -(gk[0] * wn[0] + gk[1] * wn[1] + gk[2] * wn[2]) / (den * den);
      if (th > zero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gwn[c] += gth * wn[c] / th;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lw[c] += gwn[c];
        gwd[c] = h * lw[c];                // w' = w + wd h
        lxd[c] += h * lx[c];               // x' = x + xd' h
        gxdd[c] = h * lxd[c];              // xd' = xd + xdd h
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] = lRn[c];
    }

    // ---------------------------------------------------------------------------------------------------
    // RHS backward
    // ---------------------------------------------------------------------------------------------------
    S gtau[3];
    {
      S m0 = inside(wraw[0], -a.omega_max, a.omega_max) ? gwd[0] : zero;
      S m1 = inside(wraw[1], -a.omega_max, a.omega_max) ? gwd[1] : zero;
      S m2 = inside(wraw[2], -a.omega_max, a.omega_max) ? gwd[2] : zero;
#pragma unroll
      for (int c = 0; c < 3; ++c) gtau[c] = a.Iinv[0 * 3 + c] * m0 + a.Iinv[1 * 3 + c] * m1 + a.Iinv[2 * 3 + c] * m2;   // Iinv^T
    }
    const S gsum[3] = {gxdd[0] / a.mass, gxdd[1] / a.mass, gxdd[2] / a.mass};

    S ge[3] = {zero, zero, zero}, gv = zero, gwc = zero, gS = zero;
    S gdh_p[PPL], gc_p[PPL], gn[PPL][3], gvp[PPL][3], gmuq[PPL], gr[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S f[3] = {Fr[j][0] + Ff[j][0], Fr[j][1] + Ff[j][1], Fr[j][2] + Ff[j][2]};
      S gf[3];
      MF_CROSS(gf, gtau, r[j]);            // tau += r x f : df = gtau x r
      MF_CROSS(gr[j], f, gtau);            //                dr = f x gtau
      S gFr_[3], gG[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gFr_[c] = gFr[j][c] + gsum[c] + gf[c];
        S gFf_ = gFf[j][c] + gsum[c] + gf[c];
        gG[c] = inside(Gf[j][c], -a.mg, a.mg) ? gFf_ : zero;
      }
      S gNn = gG[0] * st[j][0] + gG[1] * st[j][1] + gG[2] * st[j][2];
      S gst[3] = {Nn[j] * gG[0], Nn[j] * gG[1], Nn[j] * gG[2]};
      S gsn = -(gst[0] * nrm[j][0] + gst[1] * nrm[j][1] + gst[2] * nrm[j][2]);
      S gslip[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gn[j][c] = -sn[j] * gst[c] + gsn * slip[j][c];
        gslip[c] = gst[c] + gsn * nrm[j][c];
      }
      S cmdv[3] = {tv[j] * e[0] - vp[j][0], tv[j] * e[1] - vp[j][1], tv[j] * e[2] - vp[j][2]};
      gmuq[j] = gslip[0] * cmdv[0] + gslip[1] * cmdv[1] + gslip[2] * cmdv[2];
      S gcmd[3] = {muq[j] * gslip[0], muq[j] * gslip[1], muq[j] * gslip[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) gvp[j][c] = -gcmd[c];
      if (part[j] >= 0 && act[j]) {
        S gtv = gcmd[0] * e[0] + gcmd[1] * e[1] + gcmd[2] * e[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) ge[c] += tv[j] * gcmd[c];
        gv += gtv;
        gwc += ((part[j] & 1) ? a.half_ly : -a.half_ly) * gtv;
      }
      if (Nn[j] > zero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gFr_[c] += gNn * Fr[j][c] / Nn[j];
      }
      S gF1[3], d = zero, gA = zero;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gF1[c] = inside(F1[j][c], -a.mg, a.mg) ? gFr_[c] : zero;
        d += gF1[c] * F0[j][c];
      }
      gc_p[j] = d / csum;
      gS += -(d * cj[j]) / (csum * csum);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        S gF0 = gF1[c] * cj[j] / csum;
        gA += -(gF0 * nrm[j][c]);
        gn[j][c] += -Aj[j] * gF0;
      }
      gdh_p[j] = a.k * gA;
      S gvn = a.damp * gA;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gvp[j][c] += gvn * nrm[j][c];
        gn[j][c] += gvn * vp[j][c];
      }
    }
    gS = group_sum<G>(gS);

    S gx_[3] = {zero, zero, zero}, gxd_[3] = {zero, zero, zero}, gw_[3] = {zero, zero, zero}, gR_[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) gR_[c] = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      if (act[j]) {
        const Cell<S>& c = cell[j];
        S gc = gc_p[j] + gS;
        S gdh = gdh_p[j] + gc * ((S)-10) * cj[j] * (one - cj[j]);
        S gzq = -gdh;
        // n = u / |u|, u = (-gx, -gy, 1)
        S dotn = gn[j][0] * nrm[j][0] + gn[j][1] * nrm[j][1] + gn[j][2] * nrm[j][2];
        S gu0 = (gn[j][0] - dotn * nrm[j][0]) / nl[j], gu1 = (gn[j][1] - dotn * nrm[j][1]) / nl[j];
        S ggx = -gu0 / a.res, ggy = -gu1 / a.res;
        const S w00 = (one - c.fx) * (one - c.fy), w01 = (one - c.fx) * c.fy, w10 = c.fx * (one - c.fy), w11 = c.fx * c.fy;
        atomic_add(gzmap + c.ic, gzq * w00 - ggx - ggy);
        atomic_add(gzmap + c.i_f, gzq * w01 + ggx);
        atomic_add(gzmap + c.il, gzq * w10 + ggy);
        atomic_add(gzmap + c.ifl, gzq * w11);
        if (gmumap) {
          atomic_add(gmumap + c.ic, gmuq[j] * w00);
          atomic_add(gmumap + c.i_f, gmuq[j] * w01);
          atomic_add(gmumap + c.il, gmuq[j] * w10);
          atomic_add(gmumap + c.ifl, gmuq[j] * w11);
        }
        S zfx, zfy, mfx, mfy;
        blend_grad(c, zc4[j][0], zc4[j][1], zc4[j][2], zc4[j][3], &zfx, &zfy);
        blend_grad(c, mc4[j][0], mc4[j][1], mc4[j][2], mc4[j][3], &mfx, &mfy);
        S gp[3] = {(gzq * zfx + gmuq[j] * mfx) / a.res, (gzq * zfy + gmuq[j] * mfy) / a.res, gdh};
        // v_p = xd + w x r
        S t1[3], t2[3];
        MF_CROSS(t1, gvp[j], w);           // dr += gvp x w
        MF_CROSS(t2, r[j], gvp[j]);        // dw += r x gvp
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          gr[j][q] += t1[q];
          gw_[q] += t2[q];
          gxd_[q] += gvp[j][q];
          gx_[q] += gp[q];
          S qa = gp[q] + gr[j][q];         // p = R P + x,  r = p - x
          gR_[q * 3 + 0] += qa * P[j][0];
          gR_[q * 3 + 1] += qa * P[j][1];
          gR_[q * 3 + 2] += qa * P[j][2];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lx[c] += group_sum<G>(gx_[c]);
      lxd[c] += group_sum<G>(gxd_[c]);
      lw[c] += group_sum<G>(gw_[c]);
      ge[c] = group_sum<G>(ge[c]);
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) lR[c] += group_sum<G>(gR_[c]);
    gv = group_sum<G>(gv);
    gwc = group_sum<G>(gwc);
    if (coln >= (S)1e-6) {                 // e = col0(R) / max(|col0|, eps)
      S dote = ge[0] * e[0] + ge[1] * e[1] + ge[2] * e[2];
      lR[0] += (ge[0] - dote * e[0]) / coln;
      lR[3] += (ge[1] - dote * e[1]) / coln;
      lR[6] += (ge[2] - dote * e[2]) / coln;
    } else {
      lR[0] += ge[0] / el; lR[3] += ge[1] / el; lR[6] += ge[2] / el;
    }
    if (G == 1 || gl == 0) { gctrl[n * 2 + 0] = gv; gctrl[n * 2 + 1] = gwc; }
  }

  if (INTEG == MF_INTEG_ODEINT_EULER) add_upstream_state(row0);   // output 0 is the initial state itself

  // terrain snap of the initial height: x.z = mean_i blend(z; cell((R0 P_i + x0).xy))   (dphysics.py:567-571)
  S gx0[3] = {lx[0], lx[1], lx[2]};
  if (!a.skip_snap) {
    S R0[9], x0[2];
#pragma unroll
    for (int c = 0; c < 9; ++c) R0[c] = a.R0[b * 9 + c];
    x0[0] = a.x_init[b * 3 + 0]; x0[1] = a.x_init[b * 3 + 1];
    const S g = lx[2] / (S)a.N;
    S sx = zero, sy = zero, sR[6] = {zero, zero, zero, zero, zero, zero};
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      if (act[j]) {
        S px = P[j][0] * R0[0] + P[j][1] * R0[1] + P[j][2] * R0[2] + x0[0];
        S py = P[j][0] * R0[3] + P[j][1] * R0[4] + P[j][2] * R0[5] + x0[1];
        Cell<S> c = locate(px, py, a.d_max, a.res, a.H, last);
        S v0 = zmap[c.ic], v1 = zmap[c.i_f], v2 = zmap[c.il], v3 = zmap[c.ifl];
        atomic_add(gzmap + c.ic, g * (one - c.fx) * (one - c.fy));
        atomic_add(gzmap + c.i_f, g * (one - c.fx) * c.fy);
        atomic_add(gzmap + c.il, g * c.fx * (one - c.fy));
        atomic_add(gzmap + c.ifl, g * c.fx * c.fy);
        S dfx, dfy;
        blend_grad(c, v0, v1, v2, v3, &dfx, &dfy);
        S gpx = g * dfx / a.res, gpy = g * dfy / a.res;
        sx += gpx; sy += gpy;
#pragma unroll
        for (int q = 0; q < 3; ++q) { sR[q] += gpx * P[j][q]; sR[3 + q] += gpy * P[j][q]; }
      }
    }
    gx0[0] += group_sum<G>(sx);
    gx0[1] += group_sum<G>(sy);
    gx0[2] = zero;                          // the caller's x0.z is overwritten, so nothing flows to it
#pragma unroll
    for (int q = 0; q < 6; ++q) lR[q] += group_sum<G>(sR[q]);
  }
  if (gl == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (a.gx0) a.gx0[b * 3 + c] = gx0[c];
      a.gxd0[b * 3 + c] = lxd[c];
      a.gw0[b * 3 + c] = lw[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) a.gR0[b * 9 + c] = lR[c];
  }
}

template <typename S, int G, int PPL>
static int launch_bwd(const RolloutBwdArgs<S>& a, int integ, int block, hipStream_t st) {
  const long long threads = (long long)a.B * G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  if (integ == MF_INTEG_DYNAMICS)
    hipLaunchKernelGGL((rollout_bwd_kernel<S, G, PPL, MF_INTEG_DYNAMICS>), dim3(grid), dim3(block), 0, st, a);
  else
    hipLaunchKernelGGL((rollout_bwd_kernel<S, G, PPL, MF_INTEG_ODEINT_EULER>), dim3(grid), dim3(block), 0, st, a);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

template <typename S>
int rollout_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* stream) {
  MF_REQUIRE(d && p, MF_ERR_INVALID, "rollout_bwd: null descriptor");
  MF_REQUIRE(d->B > 0 && d->N > 0 && d->H > 1 && d->W > 0 && d->T >= 1, MF_ERR_INVALID, "rollout_bwd: B, T, N, H, W must be positive");
  MF_REQUIRE(d->n_tracks == 2 || d->n_tracks == 4, MF_ERR_INVALID, "n_tracks must be 2 or 4");
  MF_REQUIRE(d->integrator == MF_INTEG_DYNAMICS || d->integrator == MF_INTEG_ODEINT_EULER, MF_ERR_INVALID,
             "rollout_bwd: unknown integrator");
  MF_REQUIRE(p->z && p->controls && p->ts && p->points && p->part && p->x_init && p->xd0 && p->R0 && p->w0, MF_ERR_INVALID,
             "rollout_bwd: null input buffer");
  MF_REQUIRE(p->Xraw && p->Xds && p->Rs && p->Omegas, MF_ERR_INVALID, "rollout_bwd: null saved-state buffer");
  MF_REQUIRE(p->gz && p->gcontrols && p->gxd0 && p->gR0 && p->gw0, MF_ERR_INVALID, "rollout_bwd: null gradient output buffer");
  MF_REQUIRE(d->N <= 512, MF_ERR_UNSUPPORTED, "rollout_bwd: more than 512 contact points");
  int block = d->block ? d->block : 64;
  MF_REQUIRE(block == 64 || block == 128 || block == 256, MF_ERR_INVALID, "rollout_bwd: block must be 64, 128 or 256");

  RolloutBwdArgs<S> a;
  a.B = d->B; a.T = d->T; a.N = d->N; a.H = d->H; a.W = d->W;
  a.n_tracks = d->n_tracks; a.layout = d->layout; a.map_shared = d->map_shared; a.skip_snap = d->skip_snap;
  a.mass = (S)d->mass; a.mg = (S)(d->mass * d->gravity); a.k = (S)d->stiffness; a.damp = (S)d->damping;
  a.omega_max = (S)d->omega_max; a.res = (S)d->grid_res; a.d_max = (S)d->d_max; a.dt = (S)d->dt;
  a.half_ly = (S)(d->robot_size_y / 2.0);
  a.sink = (S)(d->mass * d->gravity / (d->stiffness + 1e-6));
  for (int i = 0; i < 9; ++i) a.Iinv[i] = (S)d->Iinv[i];
  a.z = (const S*)p->z; a.mu = (const S*)p->mu; a.controls = (const S*)p->controls; a.ts = (const S*)p->ts;
  a.points = (const S*)p->points; a.part = p->part;
  a.x_init = (const S*)p->x_init; a.xd0 = (const S*)p->xd0; a.R0 = (const S*)p->R0; a.w0 = (const S*)p->w0;
  a.Xraw = (const S*)p->Xraw; a.Xds = (const S*)p->Xds; a.Rs = (const S*)p->Rs; a.Om = (const S*)p->Omegas;
  a.gXs = (const S*)p->gXs; a.gXds = (const S*)p->gXds; a.gRs = (const S*)p->gRs; a.gOm = (const S*)p->gOmegas;
  a.gFs = (const S*)p->gFs; a.gFf = (const S*)p->gFf;
  a.gz = (S*)p->gz; a.gmu = (S*)p->gmu; a.gcontrols = (S*)p->gcontrols;
  a.gx0 = (S*)p->gx0; a.gxd0 = (S*)p->gxd0; a.gR0 = (S*)p->gR0; a.gw0 = (S*)p->gw0;

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
#define MF_GO(G_, P_) return launch_bwd<S, G_, P_>(a, integ, block, st)
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

extern "C" int mf_rollout_bwd_f32(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* s) {
  return mf::rollout_bwd<float>(d, p, s);
}
extern "C" int mf_rollout_bwd_f64(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* s) {
  return mf::rollout_bwd<double>(d, p, s);
}
