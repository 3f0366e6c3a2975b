// Fused DPhysics rollout, backward pass, COMPONENT-PARALLEL lane mapping (rollout_cp_common.h): the reverse-time adjoint of
// rollout_bwd_kernel.h with a rollout spread over a 16-lane row -- quad = contact point, lane c = component c of every vector /
// row c of R and of its adjoint, lane q = cell q of the bilinear footprint.  float32 fast math, rigid bodies of <= 4 contact
// points, both integrators (torchdiffeq fixed-grid Euler, dphysics.py:499-528; the semi-implicit Euler + Rodrigues step of
// `dynamics`, dphysics.py:428-466); everything else runs on the one-point-per-lane kernels.  Same derivation, same autograd
// conventions (clamp passes gradient iff inside, `.long()` indices are constants, |v| has zero gradient at 0);
// sums run in a different order.
//
// Why: at the BASELINE shape (1024 rollouts x 4 points) the G = 4 kernel is 64 waves issuing ~840 instructions per step --
// the launch is bound by the instruction stream of one wave.  Here a step is ~1/3 of that per wave (3-vector algebra is
// one instruction per lane, the 18-component adjoint update is divided over the lanes instead of repeated by each), and a
// lane owns ONE footprint cell: its gradient accumulator is a single register pair flushed by one atomic per map.
#pragma once
#include "rollout_bwd_kernel.h"
#include "rollout_cp_common.h"
#include <type_traits>

namespace mf {

// XS_ONLY: the loss reads the positions only (physics_loss, losses.py:102-127 -- every training caller): the other five upstream
// gradients are absent, so their loads, their additions and the adjoint of the impulse accumulators are compiled out
// (~19 of ~410 instructions per step).
// GCTRL = false: nobody asked for the gradient of the controls (a terrain fit, an encoder train step): its dot product, two
// sums over the contact points and the store are compiled out.
// LATE: the second half of a step's recompute (everything behind its two gathers) is placed after the vector-Jacobian chain of
// the step before it instead of beside it.  With at most one wave per SIMD nothing else hides the round trip of the gathers
// (B = 1024: 0.395 -> 0.368 ms, B = 4096: 0.63 -> 0.59 ms; dynamics(): 0.65 -> 0.47 ms); with two waves per SIMD the other
// wave does, and the early form is the faster one (B = 8192: 0.82 vs 0.93 ms).
//
// Tried and not kept -- producer / consumer waves (the recompute on a second SIMD of the CU, each step's 38 intermediates handed
// over through an LDS ring, counters instead of barriers; B <= 1024 leaves half of the SIMDs idle): the consumer's ~250
// instructions are one dependent chain, and a dependent VALU instruction issues every ~6.3 cycles against ~4.3 for an
// independent one (profiles/r1g_microbench_valu_issue.txt) -- the single wave fills exactly those bubbles with the recompute.
// Measured at B = 1024: 0.388 ms split vs 0.365 ms single wave (dynamics(): 0.478 vs 0.443).
// Also tried and not kept -- a three-stage single-wave pipeline (gathers of step n - 2, the rest of the recompute of step n - 1
// BESIDE the chain of step n, three register sets, unrolled by three): the same ~350 instructions per step, interleaved by
// the compiler instead of run one stream after the other, and slower: 0.399 vs 0.367 ms (dynamics(): 0.548 vs 0.448).
enum { kCpEarly = 0, kCpLate = 1, kCpSaved = 2, kCpStream = 3 };
#ifdef MF_STREAM_PROFILE      // A/B build: where the waves of the streaming backward spend their cycles (tools/stream_profile.py)
extern __device__ unsigned long long mf_stream_prof[16];
#define MF_PROF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define MF_PROF_ACC(name) unsigned long long name = 0
#define MF_PROF_SUM(name, t0) name += __builtin_readcyclecounter() - (t0)
#define MF_PROF_OUT(slot, val) do { if (lane == 0) atomicAdd(&mf_stream_prof[slot], (unsigned long long)(val)); } while (0)
#define MF_PROF_ADD(slot, t0) MF_PROF_OUT(slot, __builtin_readcyclecounter() - (t0))
#else
#define MF_PROF_T(v)
#define MF_PROF_ACC(name)
#define MF_PROF_SUM(name, t0)
#define MF_PROF_OUT(slot, val)
#define MF_PROF_ADD(slot, t0)
#endif
#ifdef MF_NO_WPE      // A/B build: no register limit on the streaming kernels
#define MF_STREAM_WPE
#else
#define MF_STREAM_WPE __attribute__((amdgpu_waves_per_eu((MODE == kCpStream && XS_ONLY && sizeof(S) == 4) ? 2 : 1)))
#endif
// WIN (round 5; early recompute from two waves per SIMD up): the accumulators' cell writes go to the workgroup's LDS window
// (rollout_bwd_kernel.h: win_open / win_emit / win_close) -- the host launches it only when every lane of every workgroup owns a rollout
// (no early exit in front of the barriers) on power-of-two maps.
// ONE1 (round 6): the fused physics loss in the early-recompute form -- its OWN instantiations: as a runtime switch with dummy tables (the
// streaming form's way) the three extra loads per row and 18 more registers cost every launch of the one-wave forms 6-10 % (2304 .. 8192
// rollouts: backward 0.36 -> 0.40, 0.38 -> 0.43, 0.65 -> 0.70 ms, profiles/r6_ab_one_wave_loss.txt), more than the fusion returned below
// 6144 rollouts.  Launched for early recompute only (more than one wave per SIMD): below, the loss's own two launches are cheaper.
template <typename S, int INTEG, bool XS_ONLY, bool GCTRL, int MODE, int SLOTS = 6, int BATCH = 3, bool ZMU = false, bool WIN = false, bool ONE1 = false>
// (streaming, positions-only loss: at most 256 registers, so that two workgroups -- six waves -- share a CU's four SIMDs)
__global__ void __launch_bounds__(MODE == kCpStream ? 192 : (WIN ? 512 : 256)) MF_STREAM_WPE
rollout_bwd_cp_kernel(const RolloutBwdArgs<S> a) {
  constexpr bool ODE = INTEG == MF_INTEG_ODEINT_EULER;
  constexpr bool LATE = MODE == kCpLate, STREAM = MODE == kCpStream, SAVED = MODE == kCpSaved || STREAM;
  using namespace cp;
  using M = Mth<S, std::is_same<S, float>::value>;      // float: fast math; double (the validation build): exact
  using Msk = typename MaskOf<S>::type;
  constexpr unsigned kS = (unsigned)sizeof(S), kC = 2u * kS;      // bytes of a scalar / of a control row (v, w)
  const int lane = threadIdx.x & 63;
  const int tid = STREAM ? blockIdx.x * 64 + lane : blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid >> 4;
  __shared__ S win[WIN ? 2 * kWinW * kWinW : 1];
  int wx0 = 0, wy0 = 0;
  unsigned win_flat0 = 0u, win_shift = 0u;
  if constexpr (WIN) {
    static_assert(MODE != kCpStream && sizeof(S) == 4, "the LDS gradient window serves the float32 one-wave forms");
    win_open<S, true>(a, win, (int)((blockIdx.x * blockDim.x) >> 4), &wx0, &wy0);
    __syncthreads();
    win_shift = 31u - (unsigned)__builtin_clz((unsigned)a.H);
    win_flat0 = (unsigned)wy0 + ((unsigned)wx0 << win_shift);
  }
  if (b >= a.B) return;      // (WIN: never taken -- host-checked)
  const int p = (tid >> 2) & 3, q = tid & 3, cc = q < 3 ? q : 2;
  const S one = S(1.0), zero = S(0.0);
  const int HW = a.H * a.W, last = HW - 1;
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S* zmap = a.z;
  const bool has_mu = a.mu != nullptr;
  const S* mumap = has_mu ? a.mu : a.z;
  const unsigned goff = a.map_shared ? (unsigned)(b % a.grad_copies) * (unsigned)HW : (unsigned)b * (unsigned)HW;
  S* gzmap = a.gz;
  const bool want_gmu = a.gmu != nullptr && has_mu;
  S* gmumap = want_gmu ? a.gmu : a.gz;

  // ---- per-lane constants (as the forward, rollout_fwd_cp_kernel.h) ----
  const bool act = p < a.N;
  const int pi = act ? p : 0;
  const S P0 = a.points[pi * 3 + 0], P1 = a.points[pi * 3 + 1], P2 = a.points[pi * 3 + 2];
  const int part = act ? a.part[pi] : -1;
  const S tv_v = part < 0 ? zero : one;
  const S tv_w = part < 0 ? zero : ((part & 1) ? a.half_ly : -a.half_ly);
  const S I0 = a.Iinv[cc * 3 + 0], I1 = a.Iinv[cc * 3 + 1], I2 = a.Iinv[cc * 3 + 2];     // row cc of I^-1
  const S J0 = a.Iinv[0 * 3 + cc], J1 = a.Iinv[1 * 3 + cc], J2 = a.Iinv[2 * 3 + cc];     // column cc (I^-T)
  const int cell_off = ((q & 1) ? a.H : 0) + ((q & 2) ? 1 : 0);
  const S wa_s = (q & 2) ? one : -one, wa_o = (q & 2) ? zero : one;
  const S wb_s = (q & 1) ? one : -one, wb_o = (q & 1) ? zero : one;
  const S n_mul = q < 2 ? -a.inv_res : zero, n_add = q < 2 ? zero : one;
  // cell gradient of the normal's finite differences: cells (c, f, l, fl) get (-ggx - ggy, +ggx, +ggy, 0); ggx sits in lane 0,
  // ggy in lane 1, t = quad_perm[1,0,1,1](gg) brings the partner over: nz += cgA * gg + cgB * t
  const S cgA = q == 0 ? -one : zero, cgB = q == 0 ? -one : (q == 3 ? zero : one);
  const S sel_xy = q < 2 ? one : zero;       // components that receive d(sample)/d(position) through the fractions
  const S mg = a.mg;
  // bit masks of the lane role for mask_or: a ternary on the role around lane sums becomes an exec-masked branch, and a
  // branch splits the basic block the two instruction streams of the loop are interleaved in
  const Msk lane0 = q == 0 ? ~(Msk)0 : (Msk)0, lane1 = q == 1 ? ~(Msk)0 : (Msk)0, lane2 = q >= 2 ? ~(Msk)0 : (Msk)0;

  // ---- adjoint of the state: component cc / row cc ----
  S lx = zero, lxd = zero, lw = zero, lR0 = zero, lR1 = zero, lR2 = zero;
  S laFs = zero, laFf = zero;     // adjoint of this point's impulse accumulators (extended ODE state)

  // rows: per-lane byte offsets (loop-invariant) + wave-uniform row offsets stepped by scalar arithmetic
  const unsigned row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)a.B : 1u;
  const unsigned row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)b : (unsigned)b * (unsigned)a.T;
  const unsigned v3 = (row0 * 3u + (unsigned)cc) * kS, v9 = (row0 * 9u + (unsigned)cc * 3u) * kS;
  const unsigned pcl = (unsigned)min(p, a.N - 1);      // inactive quads read a valid force row, masked at use
  // upstream rows: element strides 3 / 9 when present, 0 when the host substituted the zero row
  const unsigned u_xs = (row0 * (unsigned)a.sXs + (unsigned)cc) * kS, u_xds = (row0 * (unsigned)a.sXds + (unsigned)cc) * kS;
  const unsigned u_om = (row0 * (unsigned)a.sOm + (unsigned)cc) * kS, u_r = (row0 * (unsigned)a.sRs + (unsigned)cc * 3u) * kS;
  const unsigned u_fs = ((row0 * (unsigned)a.N + pcl) * (unsigned)a.sFs + (unsigned)cc) * kS;
  const unsigned u_ff = ((row0 * (unsigned)a.N + pcl) * (unsigned)a.sFf + (unsigned)cc) * kS;

  struct StateIn { S x, xd, w, R0, R1, R2, cv, cw, t0, t1; };
  // (a positions-only launch carries ONE upstream value per row -- and, for the fused loss of the one-wave forms (ONE1 below), the row's ground
  //  truth and weight, -1 = no stamp.  Two types, not one with ten members: the record-reading form keeps three of them in an array the
  //  compiler must be able to promote out of scratch)
  struct UpFull { S gXs, gXds, gOm, gR0, gR1, gR2, gFs, gFf; };
  struct UpOne1 { S gXs, lg, lw; };
  using UpIn = std::conditional_t<ONE1, UpOne1, UpFull>;
  static_assert(!ONE1 || (XS_ONLY && MODE != kCpStream), "ONE1: a positions-only upstream in the one-wave forms");
  // ODEINT: output row 0 is the initial state and step m maps row m -> row m + 1 (the last control is unused); DYNAMICS: step m
  // maps row m - 1 (the initial state for m = 0) -> row m
  const int n_steps = ODE ? a.T - 1 : a.T;
  const Rsrc rXraw = make_rsrc(a.Xraw), rXds = make_rsrc(a.Xds), rOm = make_rsrc(a.Om), rRs = make_rsrc(a.Rs);
  const Rsrc rgXs = make_rsrc(a.gXs), rgXds = make_rsrc(a.gXds), rgOm = make_rsrc(a.gOm), rgRs = make_rsrc(a.gRs);
  const Rsrc rgFs = make_rsrc(a.gFs), rgFf = make_rsrc(a.gFf), rCtrl = make_rsrc(a.controls), rGctrl = make_rsrc(a.gcontrols);
  const unsigned v_ctrl = (unsigned)b * (unsigned)a.T * kC;      // this rollout's control rows (bytes)
  const unsigned s3 = row_stride * 3u * kS, s9 = row_stride * 9u * kS;   // bytes between consecutive time steps
  const unsigned sg_xs = row_stride * (unsigned)a.sXs * kS, sg_xds = row_stride * (unsigned)a.sXds * kS, sg_om = row_stride * (unsigned)a.sOm * kS;
  const unsigned sg_r = row_stride * (unsigned)a.sRs * kS;
  const unsigned sg_fs = row_stride * (unsigned)a.N * (unsigned)a.sFs * kS, sg_ff = row_stride * (unsigned)a.N * (unsigned)a.sFf * kS;
  struct { S x, xd, w, R0, R1, R2; } ini = {zero, zero, zero, zero, zero, zero};
  if constexpr (!ODE) {
    ini.x = a.x_init[b * 3 + cc]; ini.xd = a.xd0[b * 3 + cc]; ini.w = a.w0[b * 3 + cc];
    ini.R0 = a.R0[b * 9 + cc * 3 + 0]; ini.R1 = a.R0[b * 9 + cc * 3 + 1]; ini.R2 = a.R0[b * 9 + cc * 3 + 2];
  }
  auto load_state = [&](int m, StateIn& s) {            // the state step m started from
    const unsigned um = __builtin_amdgcn_readfirstlane((unsigned)m);    // wave-uniform, and provably so (scalar offsets)
    if constexpr (ODE) {                                // = saved output row m
      s.x = bload1<S>(rXraw, v3, um * s3); s.xd = bload1<S>(rXds, v3, um * s3); s.w = bload1<S>(rOm, v3, um * s3);
      bload3(rRs, v9, um * s9, &s.R0, &s.R1, &s.R2);
      s.t0 = a.ts[m]; s.t1 = a.ts[m + 1 < a.T ? m + 1 : m];
    } else {                                            // = saved output row m - 1; step 0: the initial state, held in registers
      const bool init = um == 0u;                       // (a select between two base pointers becomes a branch around scalar loads)
      const unsigned ur = init ? 0u : um - 1u;
      S R0, R1, R2;
      const S x = bload1<S>(rXraw, v3, ur * s3), xd = bload1<S>(rXds, v3, ur * s3), w = bload1<S>(rOm, v3, ur * s3);
      bload3(rRs, v9, ur * s9, &R0, &R1, &R2);
      s.x = init ? ini.x : x; s.xd = init ? ini.xd : xd; s.w = init ? ini.w : w;
      s.R0 = init ? ini.R0 : R0; s.R1 = init ? ini.R1 : R1; s.R2 = init ? ini.R2 : R2;
    }
    bload2(rCtrl, v_ctrl, um * kC, &s.cv, &s.cw);
  };
  // ONE1 (round 6): `physics_loss` inside the early-recompute form of this kernel (4097 .. 8192 rollouts) -- a.gXs points at the forward's own Xs rows, dL/dXs is formed where the row is consumed, as the streaming form's
  // fetching waves and the saturated one-point-per-lane kernels (rollout_bwd_kernel.h LOSS) do.  The rows are requested from the last one
  // down (never up, sometimes twice): the stamp at or below the row at hand -- index, row, weight, all in VECTOR registers through an
  // opaque zero, so that the compiler does not pull a v_readfirstlane and its wait up to the loads -- steps down when the row passes it;
  // its predecessor's row and weight are requested a call ahead; the ground-truth address depends on the index alone.
  const S one1_scale = ONE1 ? S(2.0) * a.loss_gloss[0] * a.loss_inv_count : zero;
  const int* const o_near = ONE1 ? a.loss_near : nullptr;
  const S* const o_w = ONE1 ? a.loss_w : nullptr;
  const S* const o_gt = ONE1 ? a.loss_gt + ((size_t)b * (size_t)a.loss_T2) * 3u + (unsigned)cc : nullptr;
  // (the stamp state lives in VECTOR registers through an opaque zero: as scalars it took ~20 SGPRs the one-wave forms do not have -- 88 -> 106
  //  and 128 bytes of scratch --, and left to itself the compiler keeps a uniform load's result scalar with a v_readfirstlane, and its wait,
  //  right behind the load)
  int o_zero = 0;
  int o_j = 0, o_near_cur = 0, o_near_prev = 0;           // o_j: largest stamp whose row is <= the row last requested (starts at the last row)
  S o_w_cur = zero, o_w_prev = zero;
  if constexpr (ONE1) {
    asm volatile("v_mov_b32 %0, 0" : "=v"(o_zero));
    o_j = a.loss_T2 - 1 + o_zero;
    o_near_cur = o_near[o_j]; o_near_prev = o_near[max(o_j - 1, 0)];
    o_w_cur = o_w[o_j]; o_w_prev = o_w[max(o_j - 1, 0)];
  }
  auto load_upstream = [&](int orow, UpIn& u) {         // upstream gradients of output row `orow`
    const unsigned uo = __builtin_amdgcn_readfirstlane((unsigned)orow);
    if constexpr (ONE1) {
      const bool dec = (o_near_cur > (int)uo) & (o_j >= 0);      // the row has passed the current stamp: its predecessor takes over
      o_j -= dec ? 1 : 0;
      o_near_cur = dec ? o_near_prev : o_near_cur;
      o_w_cur = dec ? o_w_prev : o_w_cur;
      const int jp = max(o_j - 1, 0);
      o_near_prev = o_near[jp]; o_w_prev = o_w[jp];            // (first used by a later call)
      u.lw = ((o_near_cur == (int)uo) & (o_j >= 0)) ? o_w_cur : -one;      // (a weight is positive: -1 marks a row without a stamp)
      u.lg = o_gt[(size_t)(unsigned)max(o_j, 0) * 3u];
    }
    u.gXs = bload1<S>(rgXs, u_xs, uo * sg_xs);
    if constexpr (!XS_ONLY) {
      u.gXds = bload1<S>(rgXds, u_xds, uo * sg_xds); u.gOm = bload1<S>(rgOm, u_om, uo * sg_om);
      bload3(rgRs, u_r, uo * sg_r, &u.gR0, &u.gR1, &u.gR2);
      u.gFs = bload1<S>(rgFs, u_fs, uo * sg_fs); u.gFf = bload1<S>(rgFf, u_ff, uo * sg_ff);
    }
  };
  // The adjoint state (lx, lxd, lw, lR) is kept UN-SUMMED over the contact points: each quad holds its own part and the state
  // is the sum of the four.  Every use of it is linear, and only the two values met by per-point data -- the adjoints of the
  // linear and angular velocity -- need the sum every step: five lane sums per step (ten DPP adds, ~7 cycles each at one wave
  // per SIMD: tools/microbench/dpp_latency.hip) become six after the loop.  The upstream gradient of a state row goes to the
  // quad of point 0.
  const S first = p == 0 ? one : zero;
  auto add_upstream_masked = [&](const UpIn& u) {       // u: already zero outside point 0's quad
    lx += u.gXs;
    lR2 = mf_fma(u.gXs, a.sink, lR2);                     // Xs = x + R[:, 2] * sink
    if constexpr (!XS_ONLY) {
      lxd += u.gXds;
      lR0 += u.gR0; lR1 += u.gR1; lR2 += u.gR2;
      lw += u.gOm;
    }
  };
  auto add_upstream_state = [&](const UpIn& u) {
    UpIn m = u;
    S gXs_row = u.gXs;
    if constexpr (ONE1) {      // the row's slot holds Xs itself; a bitwise merge on the stamp (no branch)
      const Msk smask = u.lw > zero ? ~(Msk)0 : (Msk)0;
      gXs_row = bfi(smask, cp_loss_grad(one1_scale, u.gXs, u.lg, u.lw), zero);
    }
    m.gXs = first * gXs_row;
    if constexpr (!XS_ONLY) { m.gXds = first * u.gXds; m.gR0 = first * u.gR0; m.gR1 = first * u.gR1; m.gR2 = first * u.gR2; m.gOm = first * u.gOm; }
    add_upstream_masked(m);
  };

  // ODEINT: the last control is never used by the explicit scheme.  DYNAMICS uses all T of them: the deferred store below
  // writes these zeros to the last row first, then that step's own result over them (same lanes, program order).
  if constexpr (GCTRL && ODE) bstore2(rGctrl, v_ctrl, (unsigned)(a.T - 1) * kC, zero, zero);

  // Cell-gradient accumulator of this lane's footprint cell: contributions of consecutive steps to the SAME cell (a robot
  // moves <= 0.2 cell per step) add up in registers; when the lane's cell changes, the old pair goes to a stash that is
  // flushed with one atomic per map in the NEXT iteration, after that step's loads (vmcnt retires in order).
  unsigned acc_idx = 0u, st_idx = 0u;
  S acc_z = zero, acc_m = zero, st_z = zero, st_m = zero;
  bool st_pending = false;
  auto emit = [&](unsigned idx, S vz, S vm) {
    if constexpr (WIN) {
      if (win_emit(win, win_flat0, win_shift, (unsigned)(a.H - 1), idx, vz, vm, want_gmu)) return;
    }
    atomic_add(at32(gzmap, goff + idx), vz);
    if (want_gmu) atomic_add(at32(gmumap, goff + idx), vm);
  };
  auto flush_stash = [&]() {
    if (st_pending) emit(st_idx, st_z, st_m);
    st_pending = false;
  };

  unsigned gctrl_pending = (unsigned)(a.T - 1) * kC;      // wave-uniform byte offset of the control row written next
  S gv_pending = zero, gwc_pending = zero;

  // Everything of a step that does not depend on the adjoint: the forward recompute (the arithmetic of
  // rollout_fwd_cp_kernel.h) from the saved state the step started from.  The adjoint recurrence is the only serial part of
  // the backward, so the loop is a two-stage software pipeline: while the vector-Jacobian chain of step n runs, the
  // recompute of step n - 1 (its gathers included) runs beside it in the same basic block -- two independent instruction
  // streams that fill each other's dependency stalls -- and the saved rows are loaded two steps ahead.
  struct Rec {
    S x, xd, cv, cw, r, pc, fr, mc;                  // between the two halves of the recompute
    S R0, R1, R2, w, r1, r2, w1, w2, vp, e, il, coln2, tv, h;
    S wq, wa, wb, zc, mcv, mub, nrm, inl, cj, inv_csum, A, F0, F1, Fr, Nn, cmdv, s, sn, stv, Gf, f1, f2, wraw;
    int idx;
  };
  // first half: the footprint cell of this lane and its two gathers -- issued a whole vector-Jacobian chain (~1000 cycles)
  // before the second half consumes them
  auto recompute_gather = [&](const StateIn& st, Rec& k) {
    k.x = st.x; k.xd = st.xd; k.w = st.w; k.cv = st.cv; k.cw = st.cw;
    k.R0 = st.R0; k.R1 = st.R1; k.R2 = st.R2;
    k.h = ODE ? st.t1 - st.t0 : a.dt;
    k.r = cp_body_r(P0, P1, P2, st.R0, st.R1, st.R2);      // (the forward's own formulas, rollout_cp_common.h: same bits, same decisions)
    k.pc = k.r + st.x;
    const S lim = S(262144.0);
    const S uq = M::cell_coord(k.pc, a.d_max, a.res, a.inv_res);
    const int ui = (int)M::clamp(uq, -lim, lim);
    k.fr = uq - (S)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
    k.idx = min(max(base + cell_off, 0), last);
    k.zc = ld32(zmap, moff + (unsigned)k.idx);
    k.mc = ld32(mumap, moff + (unsigned)k.idx);
  };
  auto recompute = [&](Rec& k) {
    const S xd = k.xd, w = k.w, r = k.r, pc = k.pc, fr = k.fr;
    k.wa = mf_fma(wa_s, dpp<kB0>(fr), wa_o); k.wb = mf_fma(wb_s, dpp<kB1>(fr), wb_o);
    k.wq = k.wa * k.wb;
    k.r1 = dpp<kRot1>(r); k.r2 = dpp<kRot2>(r);
    k.w1 = dpp<kRot1>(w); k.w2 = dpp<kRot2>(w);
    k.vp = cp_vel(xd, w, r);
    k.coln2 = dot3(k.R0, k.R0);
    k.il = M::inv_len(k.coln2);
    k.e = k.R0 * k.il;
    k.tv = cp_track(tv_v, tv_w, k.cv, k.cw);
    const S zq = dot4(k.wq, k.zc);
    k.mcv = has_mu ? k.mc : one;
    k.mub = dot4(k.wq, k.mcv);
    const S dz = k.zc - dpp<kB0>(k.zc);
    const S u = mf_fma(dpp<kN12>(dz), n_mul, n_add);
    k.inl = M::inv_len(dot3(u, u));
    k.nrm = u * k.inl;
    const S dh = dpp<kB2>(pc) - zq;
    S cj = M::sigmoid_m10(dh);
    k.cj = act ? cj : zero;
    k.inv_csum = M::div(one, sum_points(k.cj));
    const S vn = dot3(k.vp, k.nrm);
    k.A = cp_normal_force(a.k, dh, a.damp, vn);
    k.F0 = -(k.A * k.nrm);
    k.F1 = cp_spring(k.A, k.nrm, k.cj, k.inv_csum);
    k.Fr = M::clamp(k.F1, -mg, mg);
    k.Nn = M::sqrt(dot3(k.Fr, k.Fr));
    k.cmdv = cp_cmd(k.tv, k.e, k.vp);
    k.s = k.mub * k.cmdv;
    k.sn = dot3(k.s, k.nrm);
    k.stv = cp_tangent(k.s, k.sn, k.nrm);
    k.Gf = k.Nn * k.stv;
    const S Ff = M::clamp(k.Gf, -mg, mg);
    const S f = k.Fr + Ff;
    k.f1 = dpp<kRot1>(f); k.f2 = dpp<kRot2>(f);
    const S Tsum = sum_points(unrot(cross_pre(r, f)));
    k.wraw = cp_wraw(I0, I1, I2, Tsum);
  };

  // vector-Jacobian product of step n given its recomputed intermediates and the upstream gradient of the forces it fed
  auto vjp = [&](int n, const Rec& k, const UpIn& up) {
    const S h = k.h, w1 = k.w1, w2 = k.w2, r1 = k.r1, r2 = k.r2, nrm = k.nrm, cj = k.cj, inv_csum = k.inv_csum;
    // ---- integrator backward: adjoint of the step's outputs -> (g_xdd, g_wd, g_Fs, g_Ff) ----
    S gFr_up = zero, gFf_up = zero, gxdd, gwd;
    if constexpr (ODE) {   // torchdiffeq fixed-grid Euler
      if constexpr (!XS_ONLY) {
        laFs += act ? up.gFs : zero;
        laFf += act ? up.gFf : zero;
        gFr_up = h * laFs; gFf_up = h * laFf;
      }
      gxdd = h * sum_points(lxd); gwd = h * sum_points(lw);
      lxd = mf_fma(h, lx, lxd);                               // x' = x + h xd
      // R' = R + h [w]x R, column by column: d/dw of (w x R_j) . g_j = R_j x g_j ; d/dR_j = g_j x w   (g_j = h lR[:, j])
      const S g0 = h * lR0, g1 = h * lR1, g2 = h * lR2;
      const S g01 = dpp<kRot1>(g0), g02 = dpp<kRot2>(g0), g11 = dpp<kRot1>(g1), g12 = dpp<kRot2>(g1), g21 = dpp<kRot1>(g2), g22 = dpp<kRot2>(g2);
      lw += (dpp<kRot1>(k.R0) * g02 - dpp<kRot2>(k.R0) * g01) + (dpp<kRot1>(k.R1) * g12 - dpp<kRot2>(k.R1) * g11) + (dpp<kRot1>(k.R2) * g22 - dpp<kRot2>(k.R2) * g21);
      lR0 += g01 * w2 - g02 * w1;
      lR1 += g11 * w2 - g12 * w1;
      lR2 += g21 * w2 - g22 * w1;
    } else {
      // xd' = xd + xdd h, x' = x + xd' h, w' = w + wd h, R' = R M(w'),  M = I + K sin(th h) + K^2 (1 - cos(th h)),
      // K = [kv]x, kv = w' / max(|w'|, eps), th = |w'|.  The forces of this step are outputs themselves.
      // With G = R^T lR (the gradient of M):  <G, K> = -kv . ax(G),  <G, K^2> = kv^T G kv - |kv|^2 tr G,
      // d/dkv = -sin ax(G) - (1 - cos) (2 tr(G) kv - (G + G^T) kv),  ax(G)_l = G[l+1][l+2] - G[l+2][l+1]
      // -- no 3x3 product is ever formed: a lane holds row c of R and lR, so  (G kv)_m = R[:, m] . (lR kv),
      // (G^T kv)_j = (R kv) . lR[:, j],  ax(G) = sum over the rows of (row of R) x (row of lR),  tr G = sum of their dots.
      if constexpr (!XS_ONLY) { gFr_up = act ? up.gFs : zero; gFf_up = act ? up.gFf : zero; }
      const S wd = M::clamp(k.wraw, -a.omega_max, a.omega_max);
      const S wn = mf_fma(wd, h, k.w);
      const S th2 = dot3(wn, wn);
      const S idn = M::inv_len(th2);                    // 1 / max(th, 1e-6)
      const S kv = wn * idn;
      const S th = M::sqrt(th2);
      S sn_, oc;
      M::sincos_small(th * h, &sn_, &oc);
      const S kk = dot3(kv, kv);
      const S kv1 = dpp<kRot1>(kv), kv2 = dpp<kRot2>(kv);
      const S ok = oc * kv;
      const S m0 = mf_fma(ok, kv, mf_fma(-oc, kk, one)), m1 = mf_fma(ok, kv1, -(sn_ * kv2)), m2 = mf_fma(ok, kv2, sn_ * kv1);   // M[c][c], M[c][c+1], M[c][c+2]
      const S q0 = dpp<kB0>(kv), q1 = dpp<kB1>(kv), q2 = dpp<kB2>(kv);
      const S Lk = lR0 * q0 + lR1 * q1 + lR2 * q2, Rk = k.R0 * q0 + k.R1 * q1 + k.R2 * q2;
      const S Gk0 = dot3(k.R0, Lk), Gk1 = dot3(k.R1, Lk), Gk2 = dot3(k.R2, Lk);
      const S Gt0 = dot3(Rk, lR0), Gt1 = dot3(Rk, lR1), Gt2 = dot3(Rk, lR2);
      const S trG = sum3(k.R0 * lR0 + k.R1 * lR1 + k.R2 * lR2);
      const S a0 = sum3(k.R1 * lR2 - k.R2 * lR1), a1 = sum3(k.R2 * lR0 - k.R0 * lR2), a2 = sum3(k.R0 * lR1 - k.R1 * lR0);
      const S ga = -(q0 * a0 + q1 * a1 + q2 * a2);
      const S gb = (q0 * Gk0 + q1 * Gk1 + q2 * Gk2) - kk * trG;
      S gth = ga * h * (one - oc) + gb * h * sn_;
      // component c of a replicated triple: bit masks on the lane role (a ternary on it turns into branches)
      const S ac = mask_or(mask_or(mask_or(zero, a0, lane0), a1, lane1), a2, lane2);
      const S sc = mask_or(mask_or(mask_or(zero, Gk0 + Gt0, lane0), Gk1 + Gt1, lane1), Gk2 + Gt2, lane2);
      const S gk = -(sn_ * ac) - oc * (S(2.0) * trG * kv - sc);
      const S gkw = dot3(gk, wn);
      const S idn2 = th2 >= S(1e-12) ? idn * idn : zero;      // through max(th, eps) only when th >= eps
      gth = mf_fma(-gkw, idn2, gth);
      const S ith = th2 > zero ? M::div(one, th) : zero;    // d|w'|/dw' = w' / |w'|, 0 at 0
      lw += mf_fma(gth * ith, wn, gk * idn);
      gwd = h * sum_points(lw);
      lxd = mf_fma(h, lx, lxd);
      gxdd = h * sum_points(lxd);
      // lR <- lR M^T: column m of the result = sum_j lR[:, j] M[m][j], M[m][j] sits in lane m as m_{(j - m) % 3}
      const S n0 = lR0 * dpp<kB0>(m0) + lR1 * dpp<kB0>(m1) + lR2 * dpp<kB0>(m2);
      const S n1 = lR0 * dpp<kB1>(m2) + lR1 * dpp<kB1>(m0) + lR2 * dpp<kB1>(m1);
      const S n2 = lR0 * dpp<kB2>(m1) + lR1 * dpp<kB2>(m2) + lR2 * dpp<kB2>(m0);
      lR0 = n0; lR1 = n1; lR2 = n2;
    }
    // ---- RHS backward ----
    const S mwd = inside(k.wraw, -a.omega_max, a.omega_max) ? gwd : zero;
    const S gtau = J0 * dpp<kB0>(mwd) + J1 * dpp<kB1>(mwd) + J2 * dpp<kB2>(mwd);      // I^-T m
    const S gsum = gxdd * a.inv_mass;
    const S gt1 = dpp<kRot1>(gtau), gt2 = dpp<kRot2>(gtau);
    const S gf = gt1 * r2 - gt2 * r1;                 // tau += r x f : df = gtau x r
    S gr = k.f1 * gt2 - k.f2 * gt1;                   //                dr = f x gtau
    S gFr = gFr_up + gsum + gf;
    const S gFf_ = gFf_up + gsum + gf;
    const S gG = inside(k.Gf, -mg, mg) ? gFf_ : zero;
    const S gNn = dot3(gG, k.stv);
    const S gst = k.Nn * gG;
    const S gsn = -dot3(gst, nrm);
    S gn = gsn * k.s - k.sn * gst;
    const S gslip = gst + gsn * nrm;
    const S gmuq = dot3(gslip, k.cmdv);
    const S gcmd = k.mub * gslip;
    S gvp = -gcmd;
    const S ge_p = k.tv * gcmd;
    S gv_p = zero, gwc_p = zero;
    if constexpr (GCTRL) {
      const S gtv = dot3(gcmd, k.e);                   // tv_v = tv_w = 0 for non-driving points
      gv_p = tv_v * gtv; gwc_p = tv_w * gtv;
    }
    gFr = mf_fma(k.Nn > zero ? gNn * M::div(one, k.Nn) : zero, k.Fr, gFr);
    const S gF1 = inside(k.F1, -mg, mg) ? gFr : zero;
    const S dF = dot3(gF1, k.F0);
    const S gc_p = dF * inv_csum;
    const S gS = sum_points(-(dF * cj) * inv_csum * inv_csum);
    const S gF0 = gF1 * cj * inv_csum;
    const S gA = -dot3(gF0, nrm);
    gn = mf_fma(-k.A, gF0, gn);
    const S gdh_p = a.k * gA;
    const S gvn = a.damp * gA;
    gvp = mf_fma(gvn, nrm, gvp);
    gn = mf_fma(gvn, k.vp, gn);
    const S gcw = gc_p + gS;
    const S gdh = gdh_p + gcw * (-S(10.0)) * cj * (one - cj);
    const S gzq = -gdh;
    // n = u / |u|, u = (-gx, -gy, 1): components 0, 1 carry the finite differences
    const S dotn = dot3(gn, nrm);
    const S gg = -((gn - dotn * nrm) * k.inl) * a.inv_res;      // lane 0: ggx, lane 1: ggy
    const S ggp = dpp<0x51>(gg);                                  // quad_perm [1,0,1,1]
    const S nz = mf_fma(gzq, k.wq, cgA * gg + cgB * ggp);
    const S nm = gmuq * k.wq;
    {   // this lane's cell accumulator
      const unsigned ni = (unsigned)k.idx;
      const bool same = !act | (ni == acc_idx);          // absent points contribute exact zeros: never flushed
      st_pending = !same;
      st_idx = acc_idx; st_z = acc_z; st_m = acc_m;
      acc_idx = act ? ni : acc_idx;
      acc_z = same ? acc_z + nz : nz;
      acc_m = same ? acc_m + nm : nm;
    }
    // d(sample)/d(position) through the fractions only: d wq / d fx = wa_s * wb, d wq / d fy = wb_s * wa
    const S vq = gzq * k.zc + gmuq * k.mcv;
    const S gpx = dot4(vq, wa_s * k.wb), gpy = dot4(vq, wb_s * k.wa);
    const S gp = mask_or(mask_or(mask_or(zero, gpx * a.inv_res, lane0), gpy * a.inv_res, lane1), gdh, lane2);
    // v_p = xd + w x r
    const S gvp1 = dpp<kRot1>(gvp), gvp2 = dpp<kRot2>(gvp);
    gr += gvp1 * w2 - gvp2 * w1;                           // dr += gvp x w
    const S gw_p = r1 * gvp2 - r2 * gvp1;              // dw += r x gvp
    const S qa = gp + gr;                              // p = R P + x, r = p - x
    // sums over the contact points
    lx += gp; lxd += gvp; lw += gw_p;                      // (each quad's own part)
    lR0 = mf_fma(qa, P0, lR0); lR1 = mf_fma(qa, P1, lR1); lR2 = mf_fma(qa, P2, lR2);
    const S ge = ge_p;
    S gv = zero, gwc = zero;
    if constexpr (GCTRL) { gv = sum_points(gv_p); gwc = sum_points(gwc_p); }
    {   // e = col0(R) / max(|col0|, eps): through |col0| only when it is >= eps
      const S dote = dot3(ge, k.e) * (k.coln2 >= S(1e-12) ? one : zero);   // (a ternary around the lane sum becomes a branch)
      lR0 = mf_fma(ge - dote * k.e, k.il, lR0);
    }
    gctrl_pending = __builtin_amdgcn_readfirstlane((unsigned)n * kC); gv_pending = gv; gwc_pending = gwc;      // stored by the next iteration (or after the loop)
  };

  // One iteration.  Memory operations in program order (vmcnt is one in-order counter over loads, stores and atomics):
  // prefetch of the rows of step n - 2 and of output row n | gathers of step n - 1 | the atomics and the control-gradient
  // store deferred from step n + 1 -- so no wait of this or the next iteration covers a younger store.  The gathers are
  // consumed beside the vector-Jacobian chain of step n or (LATE) after it.
  auto body = [&](int n, const Rec& rec, Rec& rec_next, const StateIn& s_prev, StateIn& s_pp, const UpIn& up, UpIn& up_next) {
    add_upstream_state(up);
    load_state(max(n - 2, 0), s_pp);
    load_upstream(ODE ? n : max(n - 1, 0), up_next);      // the row step n - 1 produced (ODEINT's row 0: added after the loop)
    recompute_gather(s_prev, rec_next);   // step n - 1 (after step 0: a harmless repeat of step 0)
    if constexpr (!LATE) recompute(rec_next);
    flush_stash();
    if constexpr (GCTRL) bstore2(rGctrl, v_ctrl, gctrl_pending, gv_pending, gwc_pending);      // every lane of the row: same address, same value
    vjp(n, rec, up);
    if constexpr (LATE) {
      // the gathered values pass through an empty asm that also reads the adjoint the chain ends in: their consumers cannot
      // be scheduled ahead of it (left alone, the compiler puts them a few instructions behind the gathers)
      asm("" : "+v"(rec_next.zc), "+v"(rec_next.mc) : "v"(lR0));
      recompute(rec_next);
    }
  };

  StateIn sA, sB;
  UpIn uA, uB;
  Rec recA, recB;
  int n = n_steps - 1;
  if constexpr (SAVED) {
    // The forward kept its COMPACT per-step record (layout: rollout_cp_common.h): per lane and step one 16-byte quad -- the cell
    // coordinates, the contact weight, A = k dh + d v_n and the unclamped angular acceleration.  The
    // reading wave re-gathers its footprint cell from the L2 (gather_cells) and rebuilds the rest with the forward's own
    // instructions (rebuild): no cell arithmetic from positions, no exponentials, no square roots, and -- because the
    // values that decide a clamp or the sign at the |F_n| kink are the forward's own -- no way to differentiate a different
    // function than the forward evaluated.  256 B per rollout-step (round 2: 1 KiB).
    typedef S f4v __attribute__((ext_vector_type(4)));
    struct Saved { f4v q; S zc, mc; int idx; };
    const char* const prec = reinterpret_cast<const char*>(a.rec) + (size_t)tid * kRecBytesPerLane<S>;
    const unsigned rec_step = (unsigned)a.B * 16u * kRecBytesPerLane<S>;      // bytes between consecutive steps
    auto load_saved = [&](int m, Saved& v) {
      v.q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(prec + (size_t)__builtin_amdgcn_readfirstlane((unsigned)m) * (size_t)rec_step));
    };
    // the lane's footprint cell from the recorded cell coordinates (lanes 0, 1 of the quad hold u_x, u_y), and its two gathers
    auto gather_cells = [&](Saved& v) {
      const S lim = S(262144.0);
      const int ui = (int)M::clamp(v.q.x, -lim, lim);                  // as the forward's footprint(): trunc toward zero
      const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
      v.idx = min(max(base + cell_off, 0), last);
#ifdef MF_STREAM_NO_GATHER   // A/B build: the cells as constants -- what the second round trip of the fetching wave costs
      v.zc = S(0.1) * (S)q; v.mc = S(0.8);
      return;
#endif
      // (Measured and dropped, round 4: lanes q and q ^ 2 hold neighbouring cells and could share ONE two-element load per map, as
      //  gather4 does for one point per lane -- 32 addresses per wave instead of 64.  4096 rollouts: 0.446 -> 0.49 ms; a dwordx2 at a
      //  4-byte-aligned address costs the L1 more than the two lane accesses it replaces.)
      if constexpr (ZMU) {      // the shared maps interleaved: the cell's (z, mu) as ONE aligned 8-byte load -- half the address lookups of
                                // the step's gathers where four such waves share a CU's L1 (B = 4096: 0.45 -> 0.40 ms)
        const Pk2<S> zm = *reinterpret_cast<const Pk2<S>*>(reinterpret_cast<const char*>(a.zmu) + (size_t)((unsigned)v.idx * (unsigned)(2 * sizeof(S))));
        v.zc = zm.a; v.mc = zm.b;
        return;
      }
      v.zc = ld32(zmap, moff + (unsigned)v.idx);
      v.mc = ld32(mumap, moff + (unsigned)v.idx);
    };
    auto rebuild = [&](const StateIn& st, const Saved& v, Rec& k) {
      k.R0 = st.R0; k.R1 = st.R1; k.R2 = st.R2; k.w = st.w;
      k.h = ODE ? st.t1 - st.t0 : a.dt;
      const S r = cp_body_r(P0, P1, P2, st.R0, st.R1, st.R2);
      k.r1 = dpp<kRot1>(r); k.r2 = dpp<kRot2>(r);
      k.w1 = dpp<kRot1>(st.w); k.w2 = dpp<kRot2>(st.w);
      k.vp = cp_vel(st.xd, st.w, r);
      const S lim = S(262144.0);
      const S uq = v.q.x;
      const S fr = uq - (S)(int)M::clamp(uq, -lim, lim);
      k.wa = mf_fma(wa_s, dpp<kB0>(fr), wa_o); k.wb = mf_fma(wb_s, dpp<kB1>(fr), wb_o);
      k.wq = k.wa * k.wb;
      k.cj = v.q.y; k.wraw = v.q.z; k.A = v.q.w;
      k.idx = v.idx; k.zc = v.zc; k.mcv = has_mu ? v.mc : one;
      k.mub = dot4(k.wq, k.mcv);
      const S dz = v.zc - dpp<kB0>(v.zc);
      const S u = mf_fma(dpp<kN12>(dz), n_mul, n_add);
      k.inl = M::inv_len(dot3(u, u));
      k.nrm = u * k.inl;
      k.inv_csum = M::div(one, sum_points(k.cj));
      k.coln2 = dot3(st.R0, st.R0);
      k.il = M::inv_len(k.coln2);
      k.e = st.R0 * k.il;
      k.tv = cp_track(tv_v, tv_w, st.cv, st.cw);
      k.F0 = -(k.A * k.nrm);
      k.F1 = cp_spring(k.A, k.nrm, k.cj, k.inv_csum);
      k.Fr = M::clamp(k.F1, -mg, mg);
      k.Nn = M::sqrt(dot3(k.Fr, k.Fr));
      k.cmdv = cp_cmd(k.tv, k.e, k.vp);
      k.s = k.mub * k.cmdv;
      k.sn = dot3(k.s, k.nrm);
      k.stv = cp_tangent(k.s, k.sn, k.nrm);
      k.Gf = k.Nn * k.stv;
      const S f = k.Fr + M::clamp(k.Gf, -mg, mg);
      k.f1 = dpp<kRot1>(f); k.f2 = dpp<kRot2>(f);
    };
    // fused physics loss (MfRolloutLoss): a.gXs points at the forward's Xs rows, dL/dXs of a row is formed where it is consumed
    const bool loss_on = STREAM && a.loss_gt != nullptr;      // wave-uniform
    const Msk loss_mask = loss_on ? ~(Msk)0 : (Msk)0;
    const S loss_scale = loss_on ? S(2.0) * a.loss_gloss[0] * a.loss_inv_count : zero;      // as csrc/physics_loss.hip: (2 gloss) / count
    const S* const loss_gt_lane = loss_on ? a.loss_gt + ((size_t)b * (size_t)a.loss_T2) * 3u + (unsigned)cc : a.z;
    const int* const l_row_stamp = loss_on ? a.loss_row_stamp : reinterpret_cast<const int*>(a.ts);      // dummies: T valid words
    const S* const l_row_w = loss_on ? a.loss_row_w : a.ts;
    const int l_T2m1 = loss_on ? a.loss_T2 - 1 : 0;
    if constexpr (STREAM) {
      // MODE = kCpStream (default integrator): a SECOND wave of the workgroup fetches -- the rows and the record of three steps per
      // batch straight into registers -- and, since it has the time, turns each step into the COEFFICIENTS of its vector-Jacobian
      // product (struct Coef) before writing them into an LDS ring; the first wave reads a step's coefficients from LDS and runs
      // the adjoint recurrence on them (`chain`).  Two LDS counters (steps written / steps read) instead of barriers; LDS
      // executes a wave's operations in order, so a counter written after a slot is seen after it.  The fetching wave is a
      // three-stage pipeline over batches: HBM loads of batch k + 2 | cell gathers (L2) of batch k + 1 | rebuild + ring
      // writes of batch k -- three register sets, unrolled by three, so that neither round trip is ever waited for.
      // Addresses: constant scalar bases + RUNNING 32-bit per-lane byte offsets, one vector subtract per array and step.
      const unsigned un = (unsigned)max(n, 0);
      unsigned o3 = v3 + un * s3, o9 = v9 + un * s9, oc = v_ctrl + un * kC, orc = un * rec_step;
      // (the row a step PRODUCES: default integrator row m + 1, dynamics() row m)
      constexpr unsigned kUp = ODE ? 1u : 0u;
      unsigned og_xs = u_xs + (un + kUp) * sg_xs, og_xds = u_xds + (un + kUp) * sg_xds, og_om = u_om + (un + kUp) * sg_om;
      unsigned og_r = u_r + (un + kUp) * sg_r, og_fs = u_fs + (un + kUp) * sg_fs, og_ff = u_ff + (un + kUp) * sg_ff;
      int ti = max(n, 0);                        // the requested step (default integrator: index of its lower time stamp)
      auto request_state = [&](StateIn& d, Saved& v) {      // rows + record of the step the offsets point at
        d.x = zero;                                            // (positions are not needed: the record replaces what used them)
#ifdef MF_STREAM_NO_FETCH    // A/B build: no loads at all -- times the rebuild + ring writes + the computing wave
        d.xd = S(0.1); d.w = S(0.01); d.R0 = cc == 0 ? one : zero; d.R1 = cc == 1 ? one : zero; d.R2 = cc == 2 ? one : zero; d.cv = S(0.5); d.cw = S(0.1);
        d.t1 = S(0.01); d.t0 = zero; v.q = f4v{S(100.3), S(0.2), S(0.1), S(50.0)};
        return;
#endif
        if constexpr (ODE) {
          d.xd = bload1<S>(rXds, o3, 0u); d.w = bload1<S>(rOm, o3, 0u);
          bload3(rRs, o9, 0u, &d.R0, &d.R1, &d.R2);
          d.t1 = a.ts[ti + 1]; d.t0 = a.ts[ti];             // (every fetched step is a real one: 0 <= ti <= T - 2)
        } else {
          // dynamics(): step m starts from output row m - 1 (step 0: from the initial state, held in registers) and ends in row m,
          // whose angular velocity is the w' of its Rodrigues step -- loaded, not recomputed
          const bool init = ti == 0;                          // wave-uniform
          const unsigned p3 = init ? o3 : o3 - s3, p9 = init ? o9 : o9 - s9;
          const S xd = bload1<S>(rXds, p3, 0u), w = bload1<S>(rOm, p3, 0u);
          S R0, R1, R2;
          bload3(rRs, p9, 0u, &R0, &R1, &R2);
          d.xd = init ? ini.xd : xd; d.w = init ? ini.w : w;
          d.R0 = init ? ini.R0 : R0; d.R1 = init ? ini.R1 : R1; d.R2 = init ? ini.R2 : R2;
          d.x = bload1<S>(rOm, o3, 0u);                          // (the unused position slot carries w' of row m)
          d.t1 = a.dt; d.t0 = zero;
        }
        bload2(rCtrl, oc, 0u, &d.cv, &d.cw);
        v.q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(prec + (size_t)orc));
      };
      auto request_up = [&](UpIn& u) {                       // upstream gradients of the row that step produced
#ifdef MF_STREAM_NO_FETCH
        u.gXs = S(0.001);
        if constexpr (!XS_ONLY) { u.gXds = u.gOm = u.gR0 = u.gR1 = u.gR2 = u.gFs = u.gFf = S(0.001); }
        return;
#endif
        u.gXs = bload1<S>(rgXs, og_xs, 0u);
        if constexpr (!XS_ONLY) {
          u.gXds = bload1<S>(rgXds, og_xds, 0u); u.gOm = bload1<S>(rgOm, og_om, 0u);
          bload3(rgRs, og_r, 0u, &u.gR0, &u.gR1, &u.gR2);
          u.gFs = bload1<S>(rgFs, og_fs, 0u); u.gFf = bload1<S>(rgFf, og_ff, 0u);
        }
      };
      auto step_back_up = [&]() {
        og_xs -= sg_xs;
        if constexpr (!XS_ONLY) { og_xds -= sg_xds; og_om -= sg_om; og_r -= sg_r; og_fs -= sg_fs; og_ff -= sg_ff; }
      };
      constexpr int kSlots = SLOTS, kPlanesD = ODE ? 0 : 6, kPlanes = (XS_ONLY ? 10 : 12) + kPlanesD;      // dynamics(): six more (struct CoefD)
      constexpr bool kPow2 = (kSlots & (kSlots - 1)) == 0;
      __shared__ f4v ring[kSlots * kPlanes * 64];
      __shared__ int flags[2];
      typedef __attribute__((address_space(3))) volatile int LdsCounter;      // (a generic volatile pointer would make FLAT accesses)
      LdsCounter* vflags = (LdsCounter*)flags;
      if (threadIdx.x == 0) { flags[0] = 0; flags[1] = 0; }
      __syncthreads();
      // MF_LOSS_VALUE_IN_BACKWARD: the loss VALUE as well.  The fetching waves add up the weighted squared errors of the stamped rows
      // they convert into dL/dXs anyway (the computing wave row 0's); at the end one partial sum per workgroup in a fixed order, and the
      // workgroup that takes the last ticket adds the partial sums in index order (as rollout_fwd_cp_kernel.h's LOSS kernels do).
      const bool loss_val = loss_on && a.loss_out != nullptr;      // wave-uniform
      __shared__ S l_part[3 * 16 + 64];
      auto loss_value_finish = [&](S acc) {     // every wave of the workgroup, once
        if (!loss_val) return;
        const int wv = (int)(threadIdx.x >> 6);
        if (lane < 16) l_part[wv * 16 + lane] = zero;             // (LDS executes a wave's operations in order)
        if (p == 0 && q < 3) l_part[wv * 16 + (lane >> 4) * 4 + q] = acc;
        __syncthreads();
        if (wv != 0) return;
        unsigned last_wg = 0u;
        if (lane == 0) {
          S tot = zero;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < 3; ++v) tot += (l_part[v * 16 + r * 4 + 0] + l_part[v * 16 + r * 4 + 1]) + l_part[v * 16 + r * 4 + 2];
          __builtin_nontemporal_store(tot, a.loss_partial + blockIdx.x);
          __threadfence();
          last_wg = atomicAdd(a.loss_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        last_wg = __builtin_amdgcn_readfirstlane(last_wg);
        if (last_wg) {                                            // every workgroup has written its partial sum: the mean, in index order
          __threadfence();
          const int n_act = min(64, (a.B - (int)blockIdx.x * 4) * 16);      // live lanes of this (possibly trailing) workgroup: the first n_act
          S tot = zero;
          for (unsigned k2 = (unsigned)lane; k2 < gridDim.x; k2 += (unsigned)n_act) tot += __builtin_nontemporal_load(a.loss_partial + k2);
          l_part[48 + lane] = tot;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations execute in order)
          if (lane == 0) {
            S sum = zero;
            for (int k2 = 0; k2 < n_act; ++k2) sum += l_part[48 + k2];
            a.loss_out[0] = sum * a.loss_inv_count;
            *a.loss_ticket = 0u;
          }
        }
      };
      if (threadIdx.x >= 64) {
        // ---------------- the two fetching waves ----------------
        // (Measured and dropped, round 4: handing the computing role to the wave that has its SIMD to itself where two workgroups share a
        //  CU -- by the hardware SIMD ids, tools/microbench/wave_placement.hip shows the pattern -- made 1536 / 2048 rollouts slower:
        //  0.274 / 0.292 -> 0.296 / 0.304 ms.)
        // The steps go out in batches of three; fetching wave k (0 / 1) takes every other batch -- one wave's instruction
        // stream (~150 instructions per step: rebuild, gates, coefficient products, ten LDS writes) could not stay ahead of the
        // computing wave's ~1000 cycles per step once the cell gathers had joined it (measured: 0.221 ms alone at B = 1024
        // against the chain's 0.20).  Each runs its own three-stage pipeline over its batches.  Ring: six or twelve slots, a
        // batch's first step (ordinal 3 j) sits in slot 3 j mod kSlots (six slots: wave k owns slots 3k .. 3k + 2); steps are
        // published -- `steps written` advanced -- in order: a wave waits for the other one's previous batch.
        // BATCH steps per batch (3; 2 where six waves share a CU's four SIMDs at 256 registers each: three register sets of three steps
        // spilled 200-250 bytes there, three sets of two do not)
        static_assert(kSlots % BATCH == 0 && (BATCH == 2 || BATCH == 3), "a batch must not wrap around the ring");
        struct Slot { StateIn st; Saved sv; UpIn up; S lg, lw; int sj; };      // stamp of the row, its weight and ground truth (fused loss)
        unsigned zero_lane = 0u;                                                   // 0, as a per-lane value the compiler cannot see through:
        asm("" : "+v"(zero_lane));                                                 // keeps the loads of the stamp tables VECTOR loads
        const int fk = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - 1;      // (wave-uniform, and provably so: scalar branches)
        int m = n;                                  // m: the step the offsets point at
        S l_acc = zero;                         // this wave's share of the loss value
        auto fetch = [&](Slot& r) {                 // everything of step m; then the offsets move to step m - 1
          request_state(r.st, r.sv);                // (past step 0 the offsets wrap around; nothing reads them again)
          request_up(r.up);
          // The stamp of the output row this step produced (ti + kUp) and its weight -- UNCONDITIONAL vector loads of a wave-uniform
          // address: without a fused loss they read a dummy table (the time grid) and the result is never used.  Unconditional,
          // because loads inside a branch leave the wait-count pass unable to tell how many are in flight behind the step's other
          // requests (it then waits for everything: 0.21 -> 0.27 ms); vector, because scalar loads would be consumed at once (two
          // dependent round trips in front of the step's other requests).  The ground truth they address is requested a stage
          // later, with the cell gathers.
          r.sj = ld32(l_row_stamp, (unsigned)ti + kUp + zero_lane);
          r.lw = ld32(l_row_w, (unsigned)ti + kUp + zero_lane);            // 0 where the row carries no stamp -> gradient 0
          r.lg = zero;
          o3 -= s3; o9 -= s9; oc -= kC; orc -= rec_step; --ti;
          step_back_up();
          --m;
        };
        constexpr unsigned kB = (unsigned)BATCH;
        auto skip3 = [&]() {                        // over the other wave's batch
          o3 -= kB * s3; o9 -= kB * s9; oc -= kB * kC; orc -= kB * rec_step; ti -= BATCH; m -= BATCH;
          og_xs -= kB * sg_xs;
          if constexpr (!XS_ONLY) { og_xds -= kB * sg_xds; og_om -= kB * sg_om; og_r -= kB * sg_r; og_fs -= kB * sg_fs; og_ff -= kB * sg_ff; }
        };
        // ... and everything of the step's vector-Jacobian product that does not depend on the adjoint is done HERE, on the
        // waves that have time: the rebuild, the clamp gates, 1 / |F_n|, and every product of two such values the chain would
        // form -- the ring carries the chain's COEFFICIENTS (struct Coef below), forty floats per lane and step.
        MF_PROF_T(t_fetcher);
        MF_PROF_ACC(acc_room); MF_PROF_ACC(acc_pub);
        auto room = [&](int upto) {                 // until the steps with ordinals < upto may be written (their slots have been read)
          MF_PROF_T(t0);
          while (upto - __builtin_amdgcn_readfirstlane(vflags[1]) > kSlots) __builtin_amdgcn_s_sleep(2);
          asm volatile("" ::: "memory");
          MF_PROF_SUM(acc_room, t0);
        };
        auto publish = [&](int from, int upto) {    // steps [from, upto) are in the ring: in order behind the other wave's
          asm volatile("" ::: "memory");
          MF_PROF_T(t0);
          while (__builtin_amdgcn_readfirstlane(vflags[0]) != from) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");
          MF_PROF_SUM(acc_pub, t0);
          vflags[0] = upto;
        };
        // (per step: everything is computed BEFORE the wave asks for room in the ring, the ten writes follow the grant, and the step is
        //  published at once -- with whole batches behind one grant the computing wave idled 9 % of its time while a batch was rebuilt)
        auto put = [&](const Slot& r, unsigned slot, int o) {
            Rec k;
            rebuild(r.st, r.sv, k);
            const S mw = inside(k.wraw, -a.omega_max, a.omega_max) ? one : zero;
            const S mG = inside(k.Gf, -mg, mg) ? one : zero;
            const S mF1 = inside(k.F1, -mg, mg) ? one : zero;
            const S invNn = k.Nn > zero ? M::div(one, k.Nn) : zero;
            const S cmask = k.coln2 >= S(1e-12) ? one : zero;
            const S cs = k.cj * k.inv_csum;
            // (the upstream gradient of a state row goes to point 0's quad: add_upstream_masked)
            const f4v p0 = f4v{k.R0, k.R1, k.R2, k.h};
            const f4v p1 = f4v{k.w1, k.w2, k.r1, k.r2};
            const f4v p2 = f4v{k.f1, k.f2, mw, mG * k.stv};
            const f4v p3 = f4v{mG * k.Nn, k.nrm, k.s, k.sn};
            const f4v p4 = f4v{k.cmdv, k.mub, k.tv * k.mub, k.Fr * invNn};
            const f4v p5 = f4v{mF1 * cs, mF1 * k.nrm, k.A, -(a.k * cs)};
            const f4v p6 = f4v{-(a.damp * cs), -(k.A * k.inv_csum), k.A * k.cj * k.inv_csum * k.inv_csum, -S(10.0) * k.cj * (one - k.cj)};
            const f4v p7 = f4v{k.vp, -(k.inl * a.inv_res), k.wq, idx_as(zero, k.idx)};
            const f4v p8 = f4v{k.zc, k.mcv, wa_s * k.wb * a.inv_res, wb_s * k.wa * a.inv_res};
            // (fused physics loss: the row's gXs slot holds Xs itself; dL/dXs from it, the stamp's ground truth and weight)
            // (a bitwise merge, not a select on `loss_on`: with the value's term next to it the compiler turned the select into a
            //  branch around both -- nine exec-masked blocks per batch, 0.217 -> 0.229 ms)
            // (rows without a stamp are masked by the STAMP, not by their weight 0: a rollout that diverged after its last stamp has
            //  non-finite rows there, and inf * 0 would put NaN into dL/dXs and into the value -- the unfused route and the reference
            //  never touch those rows)
            const Msk smask = r.sj >= 0 ? ~(Msk)0 : (Msk)0;
            const S gXs_row = bfi(loss_mask, bfi(smask, cp_loss_grad(loss_scale, r.up.gXs, r.lg, r.lw), zero), r.up.gXs);
            l_acc += bfi(smask, cp_loss_term(r.up.gXs, r.lg, r.lw), zero);      // (the value: used with MF_LOSS_VALUE_IN_BACKWARD only)
            const f4v p9 = f4v{k.e, k.il, cmask * k.e * k.il, first * gXs_row};
            // dynamics(): everything of the Rodrigues step R' = R M(w'), M = I + K sin(th h) + K^2 (1 - cos(th h)), K = [kv]x,
            // kv = w' / max(|w'|, eps), that does not depend on the adjoint -- from w' as the forward left it in row m
            f4v d0 = {zero, zero, zero, zero}, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0;
            if constexpr (!ODE) {
              const S h = a.dt, wn = r.st.x;
              const S th2 = dot3(wn, wn);
              const S idn = M::inv_len(th2);                    // 1 / max(th, 1e-6)
              const S kv = wn * idn;
              const S th = M::sqrt(th2);
              S sn_, oc;
              M::sincos_small(th * h, &sn_, &oc);
              const S kk = dot3(kv, kv);
              const S kv1 = dpp<kRot1>(kv), kv2 = dpp<kRot2>(kv);
              const S ok = oc * kv;
              const S m0 = mf_fma(ok, kv, mf_fma(-oc, kk, one)), m1 = mf_fma(ok, kv1, -(sn_ * kv2)), m2 = mf_fma(ok, kv2, sn_ * kv1);   // M[c][c], M[c][c+1], M[c][c+2]
              const S q0 = dpp<kB0>(kv), q1 = dpp<kB1>(kv), q2 = dpp<kB2>(kv);
              const S Rk = k.R0 * q0 + k.R1 * q1 + k.R2 * q2;
              const S idn2 = th2 >= S(1e-12) ? idn * idn : zero;      // through max(th, eps) only when th >= eps
              const S ith = th2 > zero ? M::div(one, th) : zero;    // d|w'|/dw' = w' / |w'|, 0 at 0
              d0 = f4v{wn, kv, q0, q1};
              d1 = f4v{q2, Rk, kk, sn_};
              d2 = f4v{oc, h * (one - oc), h * sn_, idn};
              d3 = f4v{idn2, ith, dpp<kB0>(m0), dpp<kB0>(m1)};          // M[0][0], M[0][1]
              d4 = f4v{dpp<kB0>(m2), dpp<kB1>(m2), dpp<kB1>(m0), dpp<kB1>(m1)};      // M[0][2], M[1][0], M[1][1], M[1][2]
              d5 = f4v{dpp<kB2>(m1), dpp<kB2>(m2), dpp<kB2>(m0), zero};  // M[2][0], M[2][1], M[2][2]
            }
            room(o + 1);
            f4v* out = ring + slot * (unsigned)(kPlanes * 64) + lane;
            out[0] = p0; out[64] = p1; out[128] = p2; out[192] = p3; out[256] = p4; out[320] = p5; out[384] = p6; out[448] = p7; out[512] = p8; out[576] = p9;
            if constexpr (!XS_ONLY) {
              out[640] = f4v{first * r.up.gXds, first * r.up.gOm, r.up.gFs, r.up.gFf};
              out[704] = f4v{first * r.up.gR0, first * r.up.gR1, first * r.up.gR2, zero};
            }
            if constexpr (!ODE) {
              constexpr int D0 = (XS_ONLY ? 10 : 12) * 64;
              out[D0] = d0; out[D0 + 64] = d1; out[D0 + 128] = d2; out[D0 + 192] = d3; out[D0 + 256] = d4; out[D0 + 320] = d5;
            }
            publish(o, o + 1);
          };
        int ord = BATCH * fk;                       // ordinal of the first step of this wave's next batch to WRITE
        unsigned sbase = kB * (unsigned)fk;         // ... and its ring slot: (BATCH j) mod kSlots
        auto fetch3 = [&](Slot (&s)[BATCH]) {
#pragma unroll
          for (int i = 0; i < BATCH; ++i) fetch(s[i]);
          skip3();
        };
        auto gather1 = [&](Slot& r) {
          gather_cells(r.sv);
          r.lg = loss_gt_lane[(size_t)(unsigned)min(max(r.sj, 0), l_T2m1) * 3u];      // (no fused loss: element 0 of the height map, unused)
        };
        auto gather3 = [&](Slot (&s)[BATCH]) {
#pragma unroll
          for (int i = 0; i < BATCH; ++i) gather1(s[i]);
        };
        auto put3 = [&](const Slot (&s)[BATCH]) {
#pragma unroll
          for (int i = 0; i < BATCH; ++i) put(s[i], sbase + (unsigned)i, ord + i);
          ord += 2 * BATCH;
          if constexpr (kSlots != 2 * BATCH) sbase = sbase + 2u * kB >= (unsigned)kSlots ? sbase + 2u * kB - (unsigned)kSlots : sbase + 2u * kB;
        };
        Slot A[BATCH], B[BATCH], C[BATCH];
        const int n_full = (n + 1) / BATCH;         // whole batches of BATCH steps in the launch (n + 1 steps)
        int full = (n_full - fk + 1) / 2;           // ... of which this wave takes every other one, starting with batch fk
        if (fk == 1) skip3();
        if (full >= 2) {
          fetch3(A); fetch3(B); gather3(A);
          full -= 2;                                // invariant: A fetched and gathered, B fetched, `full` batches not yet fetched
          while (full >= 3) {
            fetch3(C); gather3(B); put3(A);
            fetch3(A); gather3(C); put3(B);
            fetch3(B); gather3(A); put3(C);
            full -= 3;
          }
          if (full == 0) { gather3(B); put3(A); put3(B); }
          else if (full == 1) { fetch3(C); gather3(B); put3(A); gather3(C); put3(B); put3(C); }
          else { fetch3(C); gather3(B); put3(A); fetch3(A); gather3(C); put3(B); gather3(A); put3(C); put3(A); }
        } else if (full == 1) { fetch3(A); gather3(A); put3(A); }
        // the last one or two steps: the wave whose turn it would be (its offsets stand on them, its half of the ring is next)
        if (fk == (n_full & 1)) {
          int o = BATCH * n_full;
          unsigned sl = sbase;
          while (m >= 0) {                          // (fetch moves m)
            Slot r0;
            fetch(r0);
            gather1(r0);
            put(r0, sl, o);
            ++sl; ++o;
          }
        }
        MF_PROF_ADD(2 + 4 * fk, t_fetcher);
        MF_PROF_OUT(0 + 4 * fk, acc_room); MF_PROF_OUT(1 + 4 * fk, acc_pub);
        loss_value_finish(l_acc);
        return;
      }
        // ---------------- the computing wave ----------------
        int consumed = 0, seen = 0;                   // records read so far; the fetching wave's counter as last read
        unsigned rslot = 0u;                          // ring slot read next (running, wraps at kSlots)
        UpIn uZ;                                      // the upstream gradient of output row 0 (the initial state): not in the ring
        load_upstream(0, uZ);
        S l_row0 = zero;                          // row 0's term of the loss value (the other rows' are the fetching waves')
        if (loss_on && ODE) {      // (dynamics(): row 0 is the row step 0 produced -- the fetching waves' like every other)
          const S g0 = loss_gt_lane[(size_t)max(a.loss_row_stamp[0], 0) * 3u], w0 = a.loss_row_w[0];
          const bool stamped = a.loss_row_stamp[0] >= 0;
          l_row0 = stamped ? cp_loss_term(uZ.gXs, g0, w0) : zero;
          uZ.gXs = stamped ? cp_loss_grad(loss_scale, uZ.gXs, g0, w0) : zero;
        }
        // The coefficients of a step's vector-Jacobian product, as the fetching wave leaves them in the ring.  With
        // cs = c / sum c, the gates mG, mF1 (1 / 0) and d1 = gFr . (mF1 n) -- the one lane sum that serves F0 = -A n and n both:
        struct Coef {
          S R0, R1, R2, h, w1, w2, r1, r2, f1, f2, mw;
          S stvG, NnG, nrm, s, sn;          // mG stv, mG |F_n|
          S cmdv, mub, tvm, FrN;            // tv mu_b, Fr / |F_n|
          S csm, nm1, A, kcs;               // mF1 cs, mF1 n, A, -k cs          (g_dh' = kcs d1)
          S dcs, Aic, Acc, dcj;             // -d cs (g_vn = dcs d1), -A / sum c (g_c = Aic d1), A c / (sum c)^2 (g_S = sum_points(Acc d1)), -10 c (1 - c)
          S vp, inlr, wq, zc, mcv, wsb, wsa;      // -(1 / |u|) / res; d wq / d(fx, fy) / res
          S e, il, eci;                     // gate_col0 e / |col0|
          int idx;
          // dynamics() only (struct "CoefD", six more planes): the adjoint-independent half of the Rodrigues step's backward
          S wn, kv, q0, q1, q2, Rk, kk, sn_, oc, cga, cgb, idn, idn2, ith;      // cga = h (1 - cos), cgb = h sin
          S M00, M01, M02, M10, M11, M12, M20, M21, M22;                      // M[m][j], every lane holds all nine
        };
        // (the loop asks for the two steps of a trip at once, and reports them read at once: the counters cost an LDS
        //  instruction each, ~14 cycles of this wave)
        MF_PROF_T(t_compute);
        MF_PROF_ACC(acc_wait);
        auto ensure = [&](int k) {                    // until k more steps are in the ring
          MF_PROF_T(t0);
          int have = __builtin_amdgcn_readfirstlane(seen);
          while (have < consumed + k) have = __builtin_amdgcn_readfirstlane(vflags[0]);
          asm volatile("" ::: "memory");
          MF_PROF_SUM(acc_wait, t0);
        };
        auto grab = [&](Coef& c, UpIn& up) {          // the next step out of the ring (it is there: ensure)
          const f4v* o = ring + (kPow2 ? (unsigned)(consumed & (kSlots - 1)) : rslot) * (unsigned)(kPlanes * 64) + lane;
          if constexpr (!kPow2) rslot = rslot + 1u == (unsigned)kSlots ? 0u : rslot + 1u;
          const f4v c0 = o[0], c1 = o[64], c2 = o[128], c3 = o[192], c4 = o[256], c5 = o[320], c6 = o[384], c7 = o[448], c8 = o[512], c9 = o[576];
          const S idx_bits = c7.w;       // (__builtin_bit_cast applied to the element expression itself reads element 0 of the vector)
          c.R0 = c0.x; c.R1 = c0.y; c.R2 = c0.z; c.h = c0.w;
          c.w1 = c1.x; c.w2 = c1.y; c.r1 = c1.z; c.r2 = c1.w;
          c.f1 = c2.x; c.f2 = c2.y; c.mw = c2.z; c.stvG = c2.w;
          c.NnG = c3.x; c.nrm = c3.y; c.s = c3.z; c.sn = c3.w;
          c.cmdv = c4.x; c.mub = c4.y; c.tvm = c4.z; c.FrN = c4.w;
          c.csm = c5.x; c.nm1 = c5.y; c.A = c5.z; c.kcs = c5.w;
          c.dcs = c6.x; c.Aic = c6.y; c.Acc = c6.z; c.dcj = c6.w;
          c.vp = c7.x; c.inlr = c7.y; c.wq = c7.z; c.idx = idx_of(idx_bits);
          c.zc = c8.x; c.mcv = c8.y; c.wsb = c8.z; c.wsa = c8.w;
          c.e = c9.x; c.il = c9.y; c.eci = c9.z; up.gXs = c9.w;
          if constexpr (!XS_ONLY) {
            const f4v g0 = o[640];
            const S* g1 = reinterpret_cast<const S*>(o + 704);      // (three floats: an unused fourth would be a free register to the allocator)
            up.gXds = g0.x; up.gOm = g0.y; up.gFs = g0.z; up.gFf = g0.w; up.gR0 = g1[0]; up.gR1 = g1[1]; up.gR2 = g1[2];
          }
          if constexpr (!ODE) {
            constexpr int D0 = (XS_ONLY ? 10 : 12) * 64;
            const f4v e0 = o[D0], e1 = o[D0 + 64], e2 = o[D0 + 128], e3 = o[D0 + 192], e4 = o[D0 + 256];
            const S* e5 = reinterpret_cast<const S*>(o + D0 + 320);
            c.wn = e0.x; c.kv = e0.y; c.q0 = e0.z; c.q1 = e0.w;
            c.q2 = e1.x; c.Rk = e1.y; c.kk = e1.z; c.sn_ = e1.w;
            c.oc = e2.x; c.cga = e2.y; c.cgb = e2.z; c.idn = e2.w;
            c.idn2 = e3.x; c.ith = e3.y; c.M00 = e3.z; c.M01 = e3.w;
            c.M02 = e4.x; c.M10 = e4.y; c.M11 = e4.z; c.M12 = e4.w;
            c.M20 = e5[0]; c.M21 = e5[1]; c.M22 = e5[2];
          }
          ++consumed;
        };
        auto release = [&]() {                        // the steps grabbed so far may be overwritten
          seen = vflags[0];
          asm volatile("" ::: "memory");
          vflags[1] = consumed;                        // (LDS runs a wave's operations in order: the reads of grab are done by then)
        };
        auto take = [&](Coef& c, UpIn& up) { ensure(1); grab(c, up); release(); };
        // The chain of `vjp` (default integrator) on those coefficients.
        auto chain = [&](int n, const Coef& c, const UpIn& up) {
          const S h = c.h;
          S gFr_up = zero, gFf_up = zero, gxdd, gwd;
          if constexpr (ODE) {
          if constexpr (!XS_ONLY) {
            laFs += act ? up.gFs : zero;
            laFf += act ? up.gFf : zero;
            gFr_up = h * laFs; gFf_up = h * laFf;
          }
          gxdd = h * sum_points(lxd); gwd = h * sum_points(lw);
          lxd = mf_fma(h, lx, lxd);
          const S g0 = h * lR0, g1 = h * lR1, g2 = h * lR2;
          const S g01 = dpp<kRot1>(g0), g02 = dpp<kRot2>(g0), g11 = dpp<kRot1>(g1), g12 = dpp<kRot2>(g1), g21 = dpp<kRot1>(g2), g22 = dpp<kRot2>(g2);
          lw += (dpp<kRot1>(c.R0) * g02 - dpp<kRot2>(c.R0) * g01) + (dpp<kRot1>(c.R1) * g12 - dpp<kRot2>(c.R1) * g11) + (dpp<kRot1>(c.R2) * g22 - dpp<kRot2>(c.R2) * g21);
          lR0 += g01 * c.w2 - g02 * c.w1;
          lR1 += g11 * c.w2 - g12 * c.w1;
          lR2 += g21 * c.w2 - g22 * c.w1;
          } else {
            // dynamics(): the adjoint of  xd' = xd + xdd h, x' = x + xd' h, w' = w + wd h, R' = R M(w')  on the fetching waves'
            // coefficients (the derivation: `vjp` above); the forces of this step are outputs themselves
            if constexpr (!XS_ONLY) { gFr_up = act ? up.gFs : zero; gFf_up = act ? up.gFf : zero; }
            const S Lk = lR0 * c.q0 + lR1 * c.q1 + lR2 * c.q2;
            const S Gk0 = dot3(c.R0, Lk), Gk1 = dot3(c.R1, Lk), Gk2 = dot3(c.R2, Lk);
            const S Gt0 = dot3(c.Rk, lR0), Gt1 = dot3(c.Rk, lR1), Gt2 = dot3(c.Rk, lR2);
            const S trG = sum3(c.R0 * lR0 + c.R1 * lR1 + c.R2 * lR2);
            const S a0 = sum3(c.R1 * lR2 - c.R2 * lR1), a1 = sum3(c.R2 * lR0 - c.R0 * lR2), a2 = sum3(c.R0 * lR1 - c.R1 * lR0);
            const S ga = -(c.q0 * a0 + c.q1 * a1 + c.q2 * a2);
            const S gb = (c.q0 * Gk0 + c.q1 * Gk1 + c.q2 * Gk2) - c.kk * trG;
            S gth = ga * c.cga + gb * c.cgb;
            const S ac = mask_or(mask_or(mask_or(zero, a0, lane0), a1, lane1), a2, lane2);
            const S sc = mask_or(mask_or(mask_or(zero, Gk0 + Gt0, lane0), Gk1 + Gt1, lane1), Gk2 + Gt2, lane2);
            const S gk = -(c.sn_ * ac) - c.oc * (S(2.0) * trG * c.kv - sc);
            const S gkw = dot3(gk, c.wn);
            gth = mf_fma(-gkw, c.idn2, gth);
            lw += mf_fma(gth * c.ith, c.wn, gk * c.idn);
            gwd = h * sum_points(lw);
            lxd = mf_fma(h, lx, lxd);
            gxdd = h * sum_points(lxd);
            const S n0 = lR0 * c.M00 + lR1 * c.M01 + lR2 * c.M02;      // lR <- lR M^T
            const S n1 = lR0 * c.M10 + lR1 * c.M11 + lR2 * c.M12;
            const S n2 = lR0 * c.M20 + lR1 * c.M21 + lR2 * c.M22;
            lR0 = n0; lR1 = n1; lR2 = n2;
          }
          // ---- RHS backward ----
          const S mwd = c.mw * gwd;
          const S gtau = J0 * dpp<kB0>(mwd) + J1 * dpp<kB1>(mwd) + J2 * dpp<kB2>(mwd);      // I^-T m
          const S gt1 = dpp<kRot1>(gtau), gt2 = dpp<kRot2>(gtau);
          const S gq = mf_fma(gxdd, a.inv_mass, gt1 * c.r2 - gt2 * c.r1);      // to both force outputs: sum + (tau += r x f)
          S gr = c.f1 * gt2 - c.f2 * gt1;
          S gFr = gFr_up + gq;
          const S gFf_ = gFf_up + gq;
          const S gNn = dot3(gFf_, c.stvG);
          const S gst = c.NnG * gFf_;
          const S gsn = -dot3(gst, c.nrm);
          S gn = gsn * c.s - c.sn * gst;
          const S gslip = mf_fma(gsn, c.nrm, gst);
          const S gmuq = dot3(gslip, c.cmdv);
          const S gcmd = c.mub * gslip;
          S gvp = -gcmd;
          const S ge_p = c.tvm * gslip;
          S gv_p = zero, gwc_p = zero;
          if constexpr (GCTRL) {
            const S gtv = dot3(gcmd, c.e);                 // tv_v = tv_w = 0 for non-driving points
            gv_p = tv_v * gtv; gwc_p = tv_w * gtv;
          }
          gFr = mf_fma(gNn, c.FrN, gFr);
          const S gF0 = gFr * c.csm;
          const S d1 = dot3(gFr, c.nm1);
          gn = mf_fma(-c.A, gF0, gn);
          const S gvn = c.dcs * d1;
          gvp = mf_fma(gvn, c.nrm, gvp);
          gn = mf_fma(gvn, c.vp, gn);
          const S gcw = mf_fma(c.Aic, d1, sum_points(c.Acc * d1));
          const S gdh = mf_fma(gcw, c.dcj, c.kcs * d1);
          const S gzq = -gdh;
          // the terrain / friction gradient of the footprint cell: n = u / |u|, u = (-gx, -gy, 1)
          const S dotn = dot3(gn, c.nrm);
          const S gg = (gn - dotn * c.nrm) * c.inlr;                // lane 0: ggx, lane 1: ggy
          const S ggp = dpp<0x51>(gg);                               // quad_perm [1,0,1,1]
          const S nz = mf_fma(gzq, c.wq, cgA * gg + cgB * ggp);
          const S nm = gmuq * c.wq;
          {   // this lane's cell accumulator
            const unsigned ni = (unsigned)c.idx;
            const bool same = !act | (ni == acc_idx);            // absent points contribute exact zeros: never flushed
            st_pending = !same;
            st_idx = acc_idx; st_z = acc_z; st_m = acc_m;
            acc_idx = act ? ni : acc_idx;
            acc_z = same ? acc_z + nz : nz;
            acc_m = same ? acc_m + nm : nm;
          }
          const S vq = gzq * c.zc + gmuq * c.mcv;
          const S gpx = dot4(vq, c.wsb), gpy = dot4(vq, c.wsa);
          const S gp = mask_or(mask_or(mask_or(zero, gpx, lane0), gpy, lane1), gdh, lane2);
          const S gvp1 = dpp<kRot1>(gvp), gvp2 = dpp<kRot2>(gvp);
          gr += gvp1 * c.w2 - gvp2 * c.w1;                       // dr += gvp x w
          const S gw_p = c.r1 * gvp2 - c.r2 * gvp1;          // dw += r x gvp
          const S qa = gp + gr;
          lx += gp; lxd += gvp; lw += gw_p;
          lR0 = mf_fma(qa, P0, lR0); lR1 = mf_fma(qa, P1, lR1); lR2 = mf_fma(qa, P2, lR2);
          S gv = zero, gwc = zero;
          if constexpr (GCTRL) { gv = sum_points(gv_p); gwc = sum_points(gwc_p); }
          lR0 = mf_fma(-dot3(ge_p, c.e), c.eci, mf_fma(ge_p, c.il, lR0));      // e = col0(R) / max(|col0|, eps)
          gctrl_pending = __builtin_amdgcn_readfirstlane((unsigned)n * kC); gv_pending = gv; gwc_pending = gwc;
        };
        auto crunch = [&](int n, const Coef& c, const UpIn& up, Coef& c_next, UpIn& up_next, auto more, auto paired) {
          add_upstream_masked(up);
          if constexpr (decltype(paired)::value == 1) grab(c_next, up_next);                     // first of a pair: ensured by the loop
          else if constexpr (decltype(paired)::value == 2) { grab(c_next, up_next); release(); } // second of a pair
          else if constexpr (decltype(more)::value) take(c_next, up_next);
          flush_stash();
          if constexpr (GCTRL) bstore2(rGctrl, v_ctrl, gctrl_pending, gv_pending, gwc_pending);
#ifdef MF_STREAM_NO_VJP      // A/B build (tools/build_variant.sh): the computing wave only takes the steps -- times the fetching wave alone
          asm volatile("" :: "v"(c.R0), "v"(c.w1), "v"(c.f1), "v"(c.NnG), "v"(c.cmdv), "v"(c.csm), "v"(c.dcs), "v"(c.vp), "v"(c.zc), "v"(c.e));
#else
          chain(n, c, up);
#endif
        };
        using std::true_type;
        using std::false_type;
        Coef cA, cB;
        if (n_steps > 0) {
          take(cA, uA);
          using single = std::integral_constant<int, 0>;
          for (; n >= 2; n -= 2) {
            ensure(2);
            crunch(n, cA, uA, cB, uB, true_type{}, std::integral_constant<int, 1>{});
            crunch(n - 1, cB, uB, cA, uA, true_type{}, std::integral_constant<int, 2>{});
          }
          if (n == 1) { crunch(1, cA, uA, cB, uB, true_type{}, single{}); crunch(0, cB, uB, cA, uA, false_type{}, single{}); }
          else crunch(0, cA, uA, cB, uB, false_type{}, single{});
        }
        MF_PROF_ADD(9, t_compute);
        MF_PROF_OUT(8, acc_wait);
        loss_value_finish(l_row0);
        uA = uZ; uB = uZ;                             // (the epilogue reads whichever the last iteration would have requested into)
    } else {
      // MODE = kCpSaved: ONE wave reads the record itself (either integrator; launches the streaming form does not cover, and the
      // A/B leg of the parity tests).  Three stages in one instruction stream: rows, record and upstream row are requested THREE / TWO
      // iterations before they are used, the cells of step n - 1 gathered at the top of the iteration; the vector-Jacobian chain of
      // step n (~1000 cycles) runs; then step n - 1 is rebuilt from what has arrived meanwhile.  (Round 3 requested one iteration
      // ahead: with every SIMD of the chip holding such a wave -- 4096 rollouts -- an HBM round trip is longer than an iteration,
      // and the gather at the top, which needs the record's cell coordinate, waited for it: 268 instructions at ~9.8 cycles each.)
      // Three register sets, rotated by iteration count: iteration i (step n0 - i) runs on K[i % 3] / U[i % 3], gathers from and
      // rebuilds S[(i + 1) % 3] into K[(i + 1) % 3], requests step n - 3 into S[i % 3] and the upstream row of step n - 2 into
      // U[(i + 2) % 3]; unrolled by three, so every index is a constant.
      struct Raw { StateIn st; Saved sv; };
      Raw S_[3];
      Rec K_[3];
      UpIn U_[3];
      auto up_row = [&](int m) { return ODE ? max(m + 1, 0) : max(m, 0); };      // the output row step m produced (ODEINT's "step -1": row 0)
      auto run = [&](auto rtag, int n) {
        constexpr int R = decltype(rtag)::value, R1 = (R + 1) % 3, R2 = (R + 2) % 3;
        add_upstream_state(U_[R]);
        // (MF_SAVED_NO_*: A/B builds of tools/ab_saved_variants.sh -- what each stage costs where every SIMD holds such a wave; wrong results)
        gather_cells(S_[R1].sv);                                // step n - 1 (n = 0: a harmless repeat of step 0; MF_STREAM_NO_GATHER: constants)
#ifndef MF_SAVED_NO_STATE
        load_state(max(n - 3, 0), S_[R].st);
#endif
#ifndef MF_SAVED_NO_REC
        load_saved(max(n - 3, 0), S_[R].sv);
#endif
#ifndef MF_SAVED_NO_UP
        load_upstream(min(up_row(n - 2), a.T - 1), U_[R2]);
#endif
#ifndef MF_SAVED_NO_ATOMIC
        flush_stash();
#endif
        if constexpr (GCTRL) bstore2(rGctrl, v_ctrl, gctrl_pending, gv_pending, gwc_pending);
#ifndef MF_SAVED_NO_VJP
        vjp(n, K_[R], U_[R]);
#endif
        // (the gathered values pass through an empty asm that also reads the adjoint the chain ends in: their consumers stay behind it)
#ifndef MF_SAVED_NO_VJP
        asm("" : "+v"(S_[R1].sv.zc), "+v"(S_[R1].sv.mc) : "v"(lR0));
#endif
#ifndef MF_SAVED_NO_REBUILD
        rebuild(S_[R1].st, S_[R1].sv, K_[R1]);
#endif
      };
      const int n0 = max(n, 0);
      load_state(n0, S_[0].st);
      load_saved(n0, S_[0].sv);
      load_upstream(min(up_row(n0), a.T - 1), U_[0]);
      load_state(max(n0 - 1, 0), S_[1].st);
      load_saved(max(n0 - 1, 0), S_[1].sv);
      load_upstream(min(up_row(n0 - 1), a.T - 1), U_[1]);
      load_state(max(n0 - 2, 0), S_[2].st);
      load_saved(max(n0 - 2, 0), S_[2].sv);
      gather_cells(S_[0].sv);
      rebuild(S_[0].st, S_[0].sv, K_[0]);
      __builtin_amdgcn_s_waitcnt(0);
      using std::integral_constant;
      int i = 0;
      for (; i + 2 < n_steps; i += 3) {
        run(integral_constant<int, 0>{}, n0 - i);
        run(integral_constant<int, 1>{}, n0 - i - 1);
        run(integral_constant<int, 2>{}, n0 - i - 2);
      }
      if (i < n_steps) run(integral_constant<int, 0>{}, n0 - i);
      if (i + 1 < n_steps) run(integral_constant<int, 1>{}, n0 - i - 1);
      // the upstream gradient of output row 0 ("step -1") went into U[(n0 + 1) % 3]; the epilogue below reads it from uA / uB
      {
        const int r0 = (n0 + 1) % 3;
        uA = r0 == 0 ? U_[0] : (r0 == 1 ? U_[1] : U_[2]);
        uB = uA;
      }
      n = -1;
    }
  } else {
  load_state(max(n, 0), sA);
  load_upstream(min(max(n, 0) + (ODE ? 1 : 0), a.T - 1), uA);
  load_state(max(n - 1, 0), sB);
  recompute_gather(sA, recA);
  recompute(recA);
  __builtin_amdgcn_s_waitcnt(0);
  for (; n >= 1; n -= 2) {
    body(n, recA, recB, sB, sA, uA, uB);
    body(n - 1, recB, recA, sA, sB, uB, uA);
  }
  if (n == 0) body(0, recA, recB, sB, sA, uA, uB);
  }
  // the upstream gradient of output row 0 sits in the buffer the last iteration prefetched into
  UpIn up = (n_steps & 1) ? uB : uA;
  flush_stash();
  if constexpr (GCTRL) bstore2(rGctrl, v_ctrl, gctrl_pending, gv_pending, gwc_pending);
  if (act) emit(acc_idx, acc_z, acc_m);      // what is still accumulated in registers
  if constexpr (ODE) {                         // output 0 is the initial state itself (its forces are constant zeros)
    if (n_steps == 0) load_upstream(0, up);    // T == 1: the loop never ran
    add_upstream_state(up);
  }
  lx = sum_points(lx); lxd = sum_points(lxd); lw = sum_points(lw);      // the adjoint of the initial state proper
  lR0 = sum_points(lR0); lR1 = sum_points(lR1); lR2 = sum_points(lR2);

  // terrain snap of the initial height: x.z = mean_i blend(z; cell((R0 P_i + x0).xy))   (dphysics.py:567-571)
  S gx0 = lx;
  if (!a.skip_snap) {
    const S Ra = a.R0[b * 9 + cc * 3 + 0], Rb = a.R0[b * 9 + cc * 3 + 1], Rc = a.R0[b * 9 + cc * 3 + 2];
    const S x0c = a.x_init[b * 3 + cc];
    const S g = dpp<kB2>(lx) / (S)a.N;
    const S pc = (P0 * Ra + P1 * Rb + P2 * Rc) + x0c;
    const S lim = S(262144.0);
    const S uq = M::cell_coord(pc, a.d_max, a.res, a.inv_res);
    const int ui = (int)M::clamp(uq, -lim, lim);
    const S fr = uq - (S)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
    const int idx = min(max(base + cell_off, 0), last);
    const S wa = mf_fma(wa_s, dpp<kB0>(fr), wa_o), wb = mf_fma(wb_s, dpp<kB1>(fr), wb_o);
    const S zc = ld32(zmap, moff + (unsigned)idx);
    if (act) atomic_add(at32(gzmap, goff + (unsigned)idx), g * (wa * wb));
    const S gpx = dot4(zc, wa_s * wb) * g * a.inv_res, gpy = dot4(zc, wb_s * wa) * g * a.inv_res;
    S gpxy = q == 0 ? gpx : (q == 1 ? gpy : zero);
    gpxy = act ? gpxy : zero;
    gx0 = (q < 2 ? lx : zero) + sum_points(gpxy);        // the caller's x0.z is overwritten, so nothing flows to it
    lR0 += sum_points(gpxy * P0); lR1 += sum_points(gpxy * P1); lR2 += sum_points(gpxy * P2);
  }
  if (p == 0) {
    if (a.gx0) a.gx0[b * 3 + cc] = gx0;
    a.gxd0[b * 3 + cc] = lxd;
    a.gw0[b * 3 + cc] = lw;
    a.gR0[b * 9 + cc * 3 + 0] = lR0; a.gR0[b * 9 + cc * 3 + 1] = lR1; a.gR0[b * 9 + cc * 3 + 2] = lR2;
  }
  if constexpr (WIN) {
    __syncthreads();
    win_close(a, win, wx0, wy0);
  }
}

bool use_component_parallel_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, int scalar_bytes = 4);
long long cp_record_bytes(const MfRolloutDesc* d, int scalar_bytes);      // bytes of the forward's per-step record for this launch shape (0: none)
bool cp_loss_fusable(const MfRolloutDesc* d);           // both directions of this launch can carry the fused physics loss
bool cp_bwd_wants_zmu(const MfRolloutDesc* d, bool has_rec, bool has_mu);      // the launch will run the record-reading kernel on interleaved maps if it gets them
int launch_rollout_bwd_cp_f32(const RolloutBwdArgs<float>& a, int integ, bool xs_only, hipStream_t st);   // a.gcontrols may be NULL
int launch_rollout_bwd_cp_dynamics_f32(const RolloutBwdArgs<float>& a, bool xs_only, hipStream_t st);      // rollout_bwd_dyn_cp_fast.hip
void launch_rollout_bwd_cp_stream_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st);   // rollout_bwd_cp_stream_fast.hip
void launch_rollout_bwd_cp_stream_dynamics_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st);   // rollout_bwd_dyn_cp_stream_fast.hip

// Largest grid (workgroups = waves of rollouts) the streaming form takes: its LDS ring allows two workgroups per CU with six slots
// (2 x 60 / 72 KB of the CU's 160 KB), one with twelve; dynamics() carries six more planes per slot (96 / 108 KB with six slots: one
// workgroup per CU).  MF_CP_STREAM_MAX_GRID overrides the default integrator's limit (A/B runs).
inline unsigned cp_stream_max_grid(int integ = MF_INTEG_ODEINT_EULER) {
  static const int env = getenv("MF_CP_STREAM_MAX_GRID") ? atoi(getenv("MF_CP_STREAM_MAX_GRID")) : -1;
  const unsigned cus = (unsigned)device_cus();
  const unsigned v = env >= 0 ? (unsigned)env : 2u * cus;      // two workgroups per CU (MI355X: 512); dynamics(): one
  return integ == MF_INTEG_ODEINT_EULER ? v : (v < cus ? v : cus);
}

// one launch of the variant (positions-only loss?, control gradient?, late recompute?) the arguments call for
void launch_rollout_bwd_cp_stream_f64(const RolloutBwdArgs<double>& a, bool xs_only, unsigned grid, hipStream_t st);   // rollout_cp_f64.hip (the validation build)
int launch_rollout_bwd_cp_f64(const RolloutBwdArgs<double>& a, int integ, bool xs_only, hipStream_t st);
inline void launch_rollout_bwd_cp_stream_any(const RolloutBwdArgs<float>& a, int integ, bool xs_only, unsigned grid, hipStream_t st) {
  if (integ == MF_INTEG_ODEINT_EULER) launch_rollout_bwd_cp_stream_f32(a, xs_only, grid, st); else launch_rollout_bwd_cp_stream_dynamics_f32(a, xs_only, grid, st);
}
inline void launch_rollout_bwd_cp_stream_any(const RolloutBwdArgs<double>& a, int, bool xs_only, unsigned grid, hipStream_t st) {
  launch_rollout_bwd_cp_stream_f64(a, xs_only, grid, st);      // (default integrator only: cp_stream_max_grid_of<double>)
}
// float64 (validation build): a ring slot is twice the bytes -- six slots of the default integrator's planes fit a CU's LDS (123 / 147 KB,
// one workgroup per CU), dynamics()' sixteen / eighteen planes do not: its record is read by the computing wave itself (kCpSaved)
template <typename S>
inline unsigned cp_stream_max_grid_of(int integ) {
  if (sizeof(S) == 8) return integ == MF_INTEG_ODEINT_EULER ? cp_stream_max_grid(integ) : 0u;
  return cp_stream_max_grid(integ);
}
template <typename S, int INTEG>
int launch_rollout_bwd_cp_variant(const RolloutBwdArgs<S>& a, bool xs_only, hipStream_t st) {
  const long long threads = (long long)a.B * 16;
  const unsigned grid = (unsigned)((threads + 63) / 64);      // waves of rollouts
  const unsigned block = wave_unit_block(grid), wgs = (unsigned)((threads + block - 1) / block);      // (the streaming form: 192 threads, its own)
  const bool gc = a.gcontrols != nullptr;
  static const int forced = getenv("MF_CP_BWD_MODE") ? atoi(getenv("MF_CP_BWD_MODE")) : -1;      // A/B (tools/ab_cp.py): 0 early, 1 late, 2 record read by one wave
  // the forward's record when there is one: a second wave per workgroup streams it through LDS (default integrator, while the
  // rings fit the CUs' LDS), else one wave reads it itself; without a record at most one wave per SIMD: late recompute
  const int saved_mode = grid <= cp_stream_max_grid_of<S>(INTEG) && forced != kCpSaved ? kCpStream : kCpSaved;
  const int mode = a.rec ? saved_mode : (forced >= 0 && forced < kCpSaved ? forced : ((long long)grid <= device_simds() ? kCpLate : kCpEarly));
#define MF_BCP(XS_, GC_, M_) MF_KLAUNCH((rollout_bwd_cp_kernel<S, INTEG, XS_, GC_, M_>), dim3(wgs), dim3(block), 0, st, a)
  // (the record-reading mode on the interleaved maps the host staged: cp_bwd_wants_zmu)
  constexpr bool kZmu = std::is_same<S, float>::value && INTEG == MF_INTEG_ODEINT_EULER;
  // (measured and dropped, round 5: the LDS gradient window in this record-reading form -- 3072 / 4096 rollouts: 0.382 / 0.406 ms with or without)
#define MF_BCP_Z(XS_, GC_) do { if constexpr (kZmu) { if (a.zmu) { MF_KLAUNCH((rollout_bwd_cp_kernel<S, INTEG, XS_, GC_, kCpSaved, 6, 3, kZmu>), dim3(wgs), dim3(block), 0, st, a); break; } } MF_BCP(XS_, GC_, kCpSaved); } while (0)
  // early recompute beyond one wave per SIMD on ONE shared power-of-two map pair: the cell writes through an LDS window per workgroup of
  // eight waves (one workgroup per CU: 128 KB); every workgroup must be full (no early exit in front of its barriers).  MF_BWD_WIN=0: A/B.
  static const bool win_off = getenv("MF_BWD_WIN") && atoi(getenv("MF_BWD_WIN")) == 0;
  constexpr bool kWin = std::is_same<S, float>::value;
  const bool win = kWin && !win_off && mode == kCpEarly && a.map_shared && a.H == a.W && (a.H & (a.H - 1)) == 0 && threads % 512 == 0;
#define MF_BCP_W(XS_, GC_) do { if constexpr (kWin) { if (win) { MF_KLAUNCH((rollout_bwd_cp_kernel<S, INTEG, XS_, GC_, kCpEarly, 6, 3, false, kWin>), dim3((unsigned)(threads / 512)), dim3(512), 0, st, a); break; } } MF_BCP(XS_, GC_, kCpEarly); } while (0)
#define MF_BCP_L(XS_, GC_) do { if (mode == kCpStream) launch_rollout_bwd_cp_stream_any(a, INTEG, xs_only, grid, st); else if (mode == kCpSaved) MF_BCP_Z(XS_, GC_); else if (mode == kCpLate) MF_BCP(XS_, GC_, kCpLate); else MF_BCP_W(XS_, GC_); } while (0)
  // ONE1: the fused physics loss of the one-wave forms (float32; a.loss_gt set by the host for such a launch) -- instantiations of their own
  bool one1_done = false;
  if constexpr (std::is_same<S, float>::value) {
    if (xs_only && a.loss_gt != nullptr && mode != kCpStream) {
      // (the early-recompute form only -- more than one wave per SIMD: in the record-reading and the late-recompute forms the kernel loses what
      //  the loss's own two small launches cost, profiles/r6_ab_one_wave_loss.txt; cp_loss_one_wave in rollout_bwd.hip offers the fusion to
      //  exactly the shapes that run early recompute)
      MF_REQUIRE(mode == kCpEarly, MF_ERR_UNSUPPORTED, "rollout_bwd: only the early-recompute one-wave form carries a fused loss");
      one1_done = true;
#define MF_BCP1(GC_, M_, Z_, W_, G_, B_) MF_KLAUNCH((rollout_bwd_cp_kernel<S, INTEG, true, GC_, M_, 6, 3, Z_, W_, true>), dim3(G_), dim3(B_), 0, st, a)
#define MF_BCP1_L(GC_) do {                                                                                                   \
        if (win) MF_BCP1(GC_, kCpEarly, false, kWin, (unsigned)(threads / 512), 512);                                    \
        else MF_BCP1(GC_, kCpEarly, false, false, wgs, block);                                                                \
      } while (0)
      if (gc) MF_BCP1_L(true); else MF_BCP1_L(false);
#undef MF_BCP1_L
#undef MF_BCP1
    }
  }
  if (!one1_done) {
    MF_REQUIRE(a.loss_gt == nullptr || mode == kCpStream, MF_ERR_UNSUPPORTED, "rollout_bwd: the one-wave fused physics loss exists for float32 positions-only launches");
    if (xs_only) { if (gc) MF_BCP_L(true, true); else MF_BCP_L(true, false); }
    else         { if (gc) MF_BCP_L(false, true); else MF_BCP_L(false, false); }
  }
#undef MF_BCP_L
#undef MF_BCP_W
#undef MF_BCP_Z
#undef MF_BCP
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (component-parallel) launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf
