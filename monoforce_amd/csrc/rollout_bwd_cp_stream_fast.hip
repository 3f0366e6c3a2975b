// Backward rollout, component-parallel lane mapping, default integrator, from the forward's per-step record STREAMED through LDS by
// two more waves of the workgroup (MODE = kCpStream, rollout_bwd_cp_kernel.h): a translation unit of its own -- four kernels, and
// the only ones whose computing loop has no memory latency of its own to schedule around.
#include "rollout_bwd_cp_kernel.h"

#ifdef MF_STREAM_PROFILE
namespace mf { __device__ unsigned long long mf_stream_prof[16]; }
extern "C" int mf_debug_stream_profile(unsigned long long* out16, int reset) {
  if (out16) { if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mf::mf_stream_prof), sizeof(mf::mf_stream_prof)) != hipSuccess) return 1; }
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mf::mf_stream_prof), z, sizeof(z)) != hipSuccess) return 2; }
  return 0;
}
#endif
namespace mf {

// Workgroup = the computing wave + two fetching waves.  Ring: twelve slots while a CU holds one workgroup (B <= 1024: 120 / 144 KB
// of its 160 KB LDS -- the fetching waves run up to four batches ahead), six for two workgroups per CU (60 / 72 KB each; the
// positions-only variants, held to 256 registers there, fetch in batches of TWO steps: no scratch).
void launch_rollout_bwd_cp_stream_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st) {
  constexpr int I = MF_INTEG_ODEINT_EULER;
  const bool gc = a.gcontrols != nullptr;
  static const int big_ring_env = getenv("MF_CP_STREAM_BIG_RING_MAX_GRID") ? atoi(getenv("MF_CP_STREAM_BIG_RING_MAX_GRID")) : -1;
  const unsigned big_ring_max = big_ring_env >= 0 ? (unsigned)big_ring_env : (unsigned)device_cus();      // one workgroup per CU
#define MF_BCPS(XS_, GC_) do { if (grid <= big_ring_max) MF_KLAUNCH((rollout_bwd_cp_kernel<float, I, XS_, GC_, kCpStream, 12>), dim3(grid), dim3(192), 0, st, a); \
                               else MF_KLAUNCH((rollout_bwd_cp_kernel<float, I, XS_, GC_, kCpStream, 6, (XS_ ? 2 : 3)>), dim3(grid), dim3(192), 0, st, a); } while (0)
  if (xs_only) { if (gc) MF_BCPS(true, true); else MF_BCPS(true, false); }
  else         { if (gc) MF_BCPS(false, true); else MF_BCPS(false, false); }
#undef MF_BCPS
}

}  // namespace mf
