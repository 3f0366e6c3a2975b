// Backward rollout, float32 fast math: the positions-only (XS_ONLY) instantiations of the general kernel for ONE rollout per wave with 2 / 4 / 8
// contact points per lane -- bodies of 65 .. 512 points beyond the record-reading multi-wave range (> two waves per SIMD: 175 / 223 points from
// 513 rollouts up).  A `physics_loss` upstream then costs one 12-byte row per step instead of 72 + 24 N bytes of (mostly zero) rows, and the
// impulse adjoints of the N points are compiled out (round 6: bench.py's points_sweep put these launches at 3.8 x their forward's time).
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_ppl_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd_xs_ppl<float>(a, m, integ, block, st);
}
}  // namespace mf
