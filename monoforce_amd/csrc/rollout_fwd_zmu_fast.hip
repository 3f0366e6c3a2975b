// Float32 fast-math forward kernels reading ONE shared (height, friction) map pair interleaved cell by cell (ZMU): a point's
// footprint in both maps is two 16-byte loads instead of eight 4-byte ones.  Built with FMA contraction like the other
// *_fast units; same arithmetic, same bits as the kernels of rollout_fwd_fast.hip / _split_fast.hip / _cost.hip.
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_zmu_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, bool split, int cost, hipStream_t st) {
  if (cost == 2 && integ == MF_INTEG_ODEINT_EULER) return launch_rollout_fwd<float, true, false, false, 2, false, true>(a, m, integ, block, st);
  if (cost) return launch_rollout_fwd<float, true, false, false, 1, false, true>(a, m, integ, block, st);
  if (split && a.rec != nullptr && m.G >= 8 && m.G <= 64 && m.PPL == 1)      // the record of rollout_bwd_mw_kernel.h, split stores
    return forces ? launch_rollout_fwd_mw_rec<true, true, true>(a, m, integ, st) : launch_rollout_fwd_mw_rec<false, true, true>(a, m, integ, st);
  if (split) {
    if (!forces) return launch_rollout_fwd<float, true, false, false, 0, true, true>(a, m, integ, block, st);
    return launch_rollout_fwd<float, true, false, true, 0, true, true>(a, m, integ, block, st);
  }
  if (a.rec != nullptr && m.G >= 8 && m.G <= 64 && m.PPL == 1)      // the record of rollout_bwd_mw_kernel.h
    return forces ? launch_rollout_fwd_mw_rec<true, true>(a, m, integ, st) : launch_rollout_fwd_mw_rec<false, true>(a, m, integ, st);
  if (!forces) return launch_rollout_fwd<float, true, false, false, 0, false, true>(a, m, integ, block, st);
  return launch_rollout_fwd<float, true, false, true, 0, false, true>(a, m, integ, block, st);
}
}  // namespace mf
