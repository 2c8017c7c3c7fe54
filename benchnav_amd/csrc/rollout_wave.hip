// rollout_wave.hip -- the throughput kernel in the default arithmetic (rollout_wave.inc) and the public launcher.
#include "rollout_wave.inc"

static_assert(bn::kParkSteps == bn::kWaveParkSteps, "wave_park.h and mppi_kernels.h disagree on the parked steps");

namespace bn {

hipError_t launch_rollout_wave(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    return p.ref_order ? launch_rollout_wave_ref(p, mode, s) : launch_wave_r<false>(p, mode, s);
}

}  // namespace bn
