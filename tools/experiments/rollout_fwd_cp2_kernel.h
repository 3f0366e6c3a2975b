// EXPERIMENT (round 3; built with `make EXPERIMENTS=1`, run with MF_CP2_MAX_WGS=512): measured, correct, NOT faster -- kept as the
// evidence behind DESIGN.md 9's "why the forward stays one wave per rollout group".
//
// Fused DPhysics rollout, forward pass, component-parallel lane mapping, default integrator, TWO waves per workgroup.
//
// rollout_fwd_cp_kernel runs at its ISSUE bound: 151 instructions per step = 757 cycles for one wave, where the step's dependences
// would allow 287 (tools/critical_path.py).  Its loop already interleaves two instruction streams -- the contact chain
// of step n and the pose / footprint / gathers of step n + 1 -- and those are independent for a whole step: the explicit scheme
// gives pose n + 1 from the velocities step n STARTS with.  Here each stream gets a wave of its own (different SIMDs of the CU):
//   wave G ("geometry"): owns x, R.  Iteration n: pose n + 1 = pose n + h f(velocities n), its footprint and gathers, the state
//        rows of output row n, the blends under the point; hands geometry n + 1 to wave C (two 16-byte LDS writes per lane), then
//        takes velocities n + 1.
//   wave C ("contact"):  owns xd, w.  Step n: takes geometry n, runs the contact model and the wrench (the forward's `contact`),
//        stores the record quad and the force rows, hands velocities n + 1 over (one 8-byte LDS write per lane).
// Same formulas as the one-wave kernel (rollout_cp_common.h cp_*), same bits: the whole GPU suite passes with it selected.
//
// Result (MI355X, T = 500, N = 4, states + record; one-wave kernel 0.157 / 0.161 / 0.171 ms at B = 256 / 1024 / 2048):
//   hand-over by LDS counters, polled (default here)          0.178 / 0.189 / 0.203 ms
//   hand-over by s_barrier (-DMF_CP2_BARRIER), lock step        0.157 / 0.160 / 0.181 ms
// Each wave's loop is ~93 instructions (issue bound ~460 cycles), so the split itself works; what it buys is spent on the
// hand-overs.  The dependence cycle C(n) -> G(n + 1) -> C(n + 2) holds two hand-overs AND the gather's round trip (the one-wave
// kernel requests cells a whole step ahead; here the exact pose n + 1 exists only ~200 cycles before its cells are needed), and a
// polled LDS hand-over costs ~250-340 cycles (write lands, the poll that sees it, the payload read); with the barrier the step is
// max(C, G) + both waves' LDS round trips + G's exposed gather wait = the one-wave kernel's 757 cycles again.  Hiding the gather
// needs a speculative footprint a step ahead (~35 more instructions on wave G, which then bounds the step at ~650 cycles):
// not worth its complexity for <= 13 %.
#pragma once
#include "rollout_fwd_cp_kernel.h"

namespace mf {

// -DMF_CP2_SPIN_LIMIT=n (development builds): a wave gives up waiting after n polls in total instead of hanging the GPU on a protocol bug
#ifdef MF_CP2_SPIN_LIMIT
#define MF_CP2_SPIN_GUARD && ++spins < MF_CP2_SPIN_LIMIT
#else
#define MF_CP2_SPIN_GUARD
#endif

template <bool FORCES, bool ZMU, bool REC>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) rollout_fwd_cp2_kernel(const RolloutArgs<float> a) {
  using namespace cp;
  using M = Mth<float, true>;
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0: contact chain, 1: geometry and rows
  const int tid = blockIdx.x * 64 + lane;
  const int b = (tid >> 4) + a.b0;
  if (b >= a.B) return;                  // whole 16-lane rows leave together, the same ones in both waves
  const int p = (tid >> 2) & 3, q = tid & 3, cc = q < 3 ? q : 2;
  const float one = 1.0f, zero = 0.0f;
  const int HW = a.H * a.W, last = HW - 1;
  const bool has_mu = a.mu != nullptr;
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const float* zmap = a.z;
  const float* mumap = has_mu ? a.mu : a.z;

  const bool act = p < a.N;
  const int pi = act ? p : 0;
  const float P0 = a.points[pi * 3 + 0], P1 = a.points[pi * 3 + 1], P2 = a.points[pi * 3 + 2];
  const int part = act ? a.part[pi] : -1;
  const float tv_v = part < 0 ? zero : one;
  const float tv_w = part < 0 ? zero : ((part & 1) ? a.half_ly : -a.half_ly);
  const float I0 = a.Iinv[cc * 3 + 0], I1 = a.Iinv[cc * 3 + 1], I2 = a.Iinv[cc * 3 + 2];
  const float grav_c = cc == 2 ? a.mg * a.inv_mass : zero;
  const int cell_off = ((q & 1) ? a.H : 0) + ((q & 2) ? 1 : 0);
  const float wa_s = (q & 2) ? one : -one, wa_o = (q & 2) ? zero : one;
  const float wb_s = (q & 1) ? one : -one, wb_o = (q & 1) ? zero : one;
  const float n_mul = q < 2 ? -a.inv_res : zero, n_add = q < 2 ? zero : one;

  // ---- the two hand-over rings ----
  constexpr int kSlots = 4;
  __shared__ f4v ringG[kSlots * 2 * 64];       // geometry: (r, pc, zc, zq) | (mub, e, u, h)
  __shared__ f2v ringV[kSlots * 64];           // velocities: (xd, w)
  __shared__ int flags[2];                     // [0] geometries published, [1] velocity sets published
  typedef __attribute__((address_space(3))) volatile int LdsCounter;
  LdsCounter* vflags = (LdsCounter*)flags;
  if (threadIdx.x == 0) { flags[0] = 0; flags[1] = 0; }
  __syncthreads();

  // ---- start state (both waves, identically) ----
  float x, xd, w, R0, R1, R2;
  if (a.default_state) {
    const float v0 = a.controls[(size_t)b * a.ctrl_sb + 0], w0 = a.controls[(size_t)b * a.ctrl_sb + 1];
    x = zero; xd = cc == 0 ? v0 : zero; w = cc == 2 ? w0 : zero;
    R0 = cc == 0 ? one : zero; R1 = cc == 1 ? one : zero; R2 = cc == 2 ? one : zero;
    if (p == 0 && wave == 1) {
      float* oxd = const_cast<float*>(a.xd0); float* oR = const_cast<float*>(a.R0); float* ow = const_cast<float*>(a.w0);
      a.x0[b * 3 + cc] = x; oxd[b * 3 + cc] = xd; ow[b * 3 + cc] = w;
      oR[b * 9 + cc * 3 + 0] = R0; oR[b * 9 + cc * 3 + 1] = R1; oR[b * 9 + cc * 3 + 2] = R2;
    }
  } else {
    x = a.x0[b * 3 + cc]; xd = a.xd0[b * 3 + cc]; w = a.w0[b * 3 + cc];
    R0 = a.R0[b * 9 + cc * 3 + 0]; R1 = a.R0[b * 9 + cc * 3 + 1]; R2 = a.R0[b * 9 + cc * 3 + 2];
  }
  auto footprint = [&](float pc, int* idx, float* wq, float* ou) {
    const float lim = 262144.0f;
    const float u = M::cell_coord(pc, a.d_max, a.res, a.inv_res);
    const int ui = (int)M::clamp(u, -lim, lim);
    const float fr = u - (float)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
    *idx = min(max(base + cell_off, 0), last);
    const float wa = fmaf(wa_s, dpp<kB0>(fr), wa_o), wb = fmaf(wb_s, dpp<kB1>(fr), wb_o);
    *wq = wa * wb;
    *ou = u;
  };
  if (!a.skip_snap) {      // (both waves: the same value; wave 1 writes it back)
    const float pc = (P0 * R0 + P1 * R1 + P2 * R2) + x;
    int idx; float wq, u;
    footprint(pc, &idx, &wq, &u);
    const float zq = dot4(wq, ld32(zmap, moff + (unsigned)idx));
    const float acc = sum_points(act ? zq : zero);
    const float xz = acc / (float)a.N;
    x = cc == 2 ? xz : x;
    if (p == 0 && q == 2 && wave == 1) a.x0[b * 3 + 2] = xz;
  }
  const int n_steps = a.T - 1;
  int spins = 0;
  const unsigned row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)a.B : 1u;
  const unsigned row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)b : (unsigned)b * (unsigned)a.T;
  const unsigned ts_last = (unsigned)(a.T - 1) * 4u;

  if (wave == 1) {
    // =============================== geometry wave ===============================
    float* v3base = p == 0 ? a.Xs : p == 1 ? a.Xds : p == 2 ? a.Om : (a.Xraw ? a.Xraw : a.Xs);
    const float sink_l = (p == 3 && a.Xraw) ? zero : a.sink;
    const unsigned m_xd = p == 1 ? ~0u : 0u, m_w = p == 2 ? ~0u : 0u, m_x = ~(m_xd | m_w);
    char* p3 = reinterpret_cast<char*>(v3base) + (size_t)(row0 * 3u + (unsigned)cc) * 4u;
    unsigned o9 = (row0 * 9u + (unsigned)cc * 3u) * 4u;
    const unsigned d3 = row_stride * 12u, d9 = row_stride * 36u;
    const char* pRs = reinterpret_cast<const char*>(a.Rs);
    auto emit_row = [&](float ex, float e0, float e1, float e2) {
      const float vx = fmaf(e2, sink_l, ex);
      const float v3 = mask_or(mask_or(mask_or(zero, vx, m_x), xd, m_xd), w, m_w);
      __builtin_nontemporal_store(v3, reinterpret_cast<float*>(p3));
      bstore3(pRs, o9, 0u, e0, e1, e2);
      p3 += d3; o9 += d9;
    };
    struct Geo { float r, pc, wq, zc, mc, e, u; };
    auto geometry = [&]() {      // of the pose in (x, R0..2): footprint, gathers, thrust direction
      Geo g;
      g.r = cp_body_r(P0, P1, P2, R0, R1, R2);
      g.pc = g.r + x;
      int idx;
      footprint(g.pc, &idx, &g.wq, &g.u);
      if constexpr (ZMU) {
        const float2 zm = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.zmu) + (size_t)((unsigned)idx * 8u));
        g.zc = zm.x; g.mc = zm.y;
      } else {
        g.zc = ld32(zmap, moff + (unsigned)idx);
        g.mc = ld32(mumap, moff + (unsigned)idx);
      }
      g.e = R0 * M::inv_len(dot3(R0, R0));
      return g;
    };
    // geometry of step k and its size: the blends under the point are formed here (this wave has waited for the cells anyway)
    auto publish_geo = [&](const Geo& g, float h, int k) {
      const float zq = dot4(g.wq, g.zc);                             // height under the point (:211)
      const float mub = dot4(g.wq, has_mu ? g.mc : one);             // friction (:216); no map = a map of ones (:562)
      f4v* o = ringG + (unsigned)(k & (kSlots - 1)) * 128u + lane;
      o[0] = f4v{g.r, g.pc, g.zc, zq};
      o[64] = f4v{mub, g.e, g.u, h};
      asm volatile("" ::: "memory");
#ifndef MF_CP2_BARRIER
      vflags[0] = k + 1;
#endif
    };
    // time stamps two ahead: the size of step n + 1 is known while step n's pose is formed
    const Rsrc rTs = make_rsrc(a.ts);
    unsigned v_ts = min(4u, ts_last);
    float t_b = a.ts[min(1, a.T - 1)];
    float h = t_b - a.ts[0];
    publish_geo(geometry(), h, 0);
    __builtin_amdgcn_s_waitcnt(0);
#ifdef MF_CP2_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    for (int n = 0; n < n_steps; ++n) {
      v_ts = min(v_ts + 4u, ts_last);
      const float t_c = bload1(rTs, v_ts + 0u * (unsigned)lane, 0u);
      // pose n + 1 from the velocities step n starts with (torchdiffeq fixed-grid euler)
      const float d0 = h * cross_pre(w, R0), d1 = h * cross_pre(w, R1), d2 = h * cross_pre(w, R2);
      const float ox = x, o0 = R0, o1 = R1, o2 = R2;
      x = fmaf(h, xd, x);
      R0 = o0 + unrot(d0); R1 = o1 + unrot(d1); R2 = o2 + unrot(d2);
      const Geo g = geometry();                     // gathers requested ...
      emit_row(ox, o0, o1, o2);                     // ... row n stored after them in program order: the wait for the cells covers no store
      h = t_c - t_b; t_b = t_c;                     // (after the last step: 0, unused)
      publish_geo(g, h, n + 1);                     // (after the last step: the final pose's -- unread, its slot is free)
      // velocities n + 1, from the contact wave
      {
        const f2v* src = ringV + (unsigned)((n + 1) & (kSlots - 1)) * 64u + lane;
        int have; f2v v;
#ifdef MF_CP2_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        v = *src;
#else
        do {
          have = vflags[1];
          asm volatile("" ::: "memory");
          v = *src;
          asm volatile("" ::: "memory");
        } while (__builtin_amdgcn_readfirstlane(have) < n + 1 MF_CP2_SPIN_GUARD);
#endif
        xd = v.x; w = v.y;
      }
    }
    emit_row(x, R0, R1, R2);                        // the last row
    return;
  }

  // =============================== contact wave ===============================
  const Rsrc rCtrl = make_rsrc(a.controls);
  unsigned v_ctrl = (unsigned)b * (unsigned)a.ctrl_sb * 4u;
  const unsigned ctrl_step = (unsigned)a.ctrl_st * 4u, v_ctrl_last = v_ctrl + (unsigned)(a.T - 1) * ctrl_step;
  float cv, cw;
  bload2(rCtrl, v_ctrl, 0u, &cv, &cw);
  const unsigned frow = (unsigned)a.fstride * 3u;
  unsigned ofs = (row0 * frow + (unsigned)p * 3u + (unsigned)cc) * 4u;
  const unsigned df = row_stride * frow * 4u;
  const char* pFs = reinterpret_cast<const char*>(a.Fs);
  const char* pFf = reinterpret_cast<const char*>(a.Ff);
  char* const pRec0 = reinterpret_cast<char*>(a.rec);
  unsigned rec_off = (unsigned)tid * kRecBytesPerLane;
  const unsigned rec_step = (unsigned)a.B * 16u * kRecBytesPerLane;
  float oFs = zero, oFf = zero;
  __builtin_amdgcn_s_waitcnt(0);
#ifdef MF_CP2_BARRIER
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
  // one step; the controls of step n come in, those of step n + 1 go out into the OTHER register pair (the loop alternates two:
  // a copy at the end of the step would wait for the load -- and, behind it in the queue, for the record's store)
  auto step = [&](int n, float cv, float cw, float& cv_next, float& cw_next) {
    // geometry of step n, from the geometry wave
    f4v g0, g1;
    {
      const f4v* src = ringG + (unsigned)(n & (kSlots - 1)) * 128u + lane;
#ifdef MF_CP2_BARRIER
      g0 = src[0]; g1 = src[64];
#else
      int have;
      do {
        have = vflags[0];
        asm volatile("" ::: "memory");
        g0 = src[0]; g1 = src[64];
        asm volatile("" ::: "memory");
      } while (__builtin_amdgcn_readfirstlane(have) < n + 1 MF_CP2_SPIN_GUARD);
#endif
    }
    // next step's controls, requested before anything is stored
    v_ctrl = min(v_ctrl + ctrl_step, v_ctrl_last);
    bload2(rCtrl, v_ctrl, 0u, &cv_next, &cw_next);
    if constexpr (FORCES) { bstore1(pFs, ofs, 0u, oFs); bstore1(pFf, ofs, 0u, oFf); ofs += df; }      // force rows of output row n
    const float gr = g0.x, gpc = g0.y, gzc = g0.z, zq = g0.w, mub = g1.x, ge = g1.y, gu = g1.z, h = g1.w;
    const float tv = cp_track(tv_v, tv_w, cv, cw);
    // ---- the contact model and the wrench: rollout_fwd_cp_kernel.h `contact`, line by line ----
    const float vp = cp_vel(xd, w, gr);
    const float dz = gzc - dpp<kB0>(gzc);
    const float u = fmaf(dpp<kN12>(dz), n_mul, n_add);
    const float inl = M::inv_len(dot3(u, u));
    const float nrm = u * inl;
    const float dh = dpp<kB2>(gpc) - zq;
    float cj = M::sigmoid_m10(dh);
    cj = act ? cj : zero;
    const float csum = sum_points(cj);
    const float inv_csum = M::div(one, csum);
    const float vn = dot3(vp, nrm);
    const float A = cp_normal_force(a.k, dh, a.damp, vn);
    const float Fr = M::clamp(cp_spring(A, nrm, cj, inv_csum), -a.mg, a.mg);
    const float Nn = M::sqrt(dot3(Fr, Fr));
    const float s = mub * cp_cmd(tv, ge, vp);
    const float sn = dot3(s, nrm);
    const float Ff = M::clamp(Nn * cp_tangent(s, sn, nrm), -a.mg, a.mg);
    const float f = Fr + Ff;
    const float tau = unrot(cross_pre(gr, f));
    const float Fsum = sum_points(f), Tsum = sum_points(tau);
    const float wraw = cp_wraw(I0, I1, I2, Tsum);
    const float wd = M::clamp(wraw, -a.omega_max, a.omega_max);
    const float xdd = Fsum * a.inv_mass - grav_c;
    xd = fmaf(h, xdd, xd);
    w = fmaf(h, wd, w);
    // velocities n + 1 to the geometry wave
    ringV[(unsigned)((n + 1) & (kSlots - 1)) * 64u + lane] = f2v{xd, w};
    asm volatile("" ::: "memory");
#ifndef MF_CP2_BARRIER
    vflags[1] = n + 1;
#endif
    if constexpr (REC) {
      const f4v rq = {gu, cj, wraw, A};
      __builtin_nontemporal_store(rq, reinterpret_cast<f4v*>(pRec0 + (size_t)rec_off));
      rec_off += rec_step;
    }
    oFs = fmaf(h, Fr, oFs);
    oFf = fmaf(h, Ff, oFf);
#ifdef MF_CP2_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
  };
  float cvB = zero, cwB = zero;
  int n = 0;
  for (; n + 1 < n_steps; n += 2) {
    step(n, cv, cw, cvB, cwB);
    step(n + 1, cvB, cwB, cv, cw);
  }
  if (n < n_steps) step(n, cv, cw, cvB, cwB);
  if constexpr (FORCES) { bstore1(pFs, ofs, 0u, oFs); bstore1(pFf, ofs, 0u, oFf); }      // the last row
}

int launch_rollout_fwd_cp2_f32(const RolloutArgs<float>& a, bool forces, bool zmu, hipStream_t st);      // rollout_fwd_cp2_fast.hip
bool use_two_wave_forward(const MfRolloutDesc* d);      // launches of few enough workgroups that both waves find a SIMD of their own

}  // namespace mf
