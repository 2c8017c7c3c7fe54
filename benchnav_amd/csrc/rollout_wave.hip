// rollout_wave.hip -- the throughput kernel in the default arithmetic (rollout_wave.inc) and the public launcher.
#include "rollout_wave.inc"

namespace bn {

hipError_t launch_rollout_wave(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    return p.ref_order ? launch_rollout_wave_ref(p, mode, s) : launch_wave_r<false>(p, mode, s);
}

}  // namespace bn
