// Forward rollout, float32 fast-math instantiations: this TU is compiled with -ffp-contract=fast (FMA), and the
// kernels use the hardware reciprocal / rsqrt / exp2 (Mth<float, true> in rollout_fwd_kernel.h).
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, hipStream_t st) {
  if (a.rec != nullptr && m.G >= 8 && m.PPL == 1)      // the record of rollout_bwd_mw_kernel.h
    return forces ? launch_rollout_fwd_mw_rec<true>(a, m, integ, st) : launch_rollout_fwd_mw_rec<false>(a, m, integ, st);
  if (!forces) return launch_rollout_fwd<float, true, false, false>(a, m, integ, block, st);   // states only
  return launch_rollout_fwd<float, true>(a, m, integ, block, st);
}
}  // namespace mf
