// rollout_wave_ref.hip -- the one-wave kernel in the reference's operation order (BN_FLAG_REFERENCE_ORDER, chain_step<..., REF>).
// Its own translation unit because it is compiled WITHOUT the SLP vectoriser (benchnav_amd/build.py): with it, hipcc 7.2 carries
// the position (x, y) of a rollout through the step loop as one <2 x float> value and, in the variants that load injected noise,
// allocates the packed add of the next step a register pair nothing has written (correct LLVM IR, garbage trajectories on the
// hardware; tests/test_gpu_census.py compares every variant with the oracle bit for bit).
#include "rollout_wave.inc"

namespace bn {

hipError_t launch_rollout_wave_ref(const SolveParams &p, EpsMode mode, hipStream_t s) { return launch_wave_r<true>(p, mode, s); }

}  // namespace bn
