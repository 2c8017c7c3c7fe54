// rollout_lat_host.hip -- the HOST-PACED latency kernel (rollout_lat.inc, HOSTP): launched one control step ahead, it waits on the
// device for the state the host posts (bn_mppi_forward_state_async in a loop).  Philox noise only; a translation unit of its own.
#define BN_ROLE_EPS kEpsPhilox
#define BN_ROLE_REF false
#define BN_LAT_MODE 2
#include "mppi_device.h"
#include "rollout_lat.inc"

namespace bn {
hipError_t launch_rollout_lat_host(const SolveParams &p, hipStream_t s, hipEvent_t stop)
{
    g_lat_stop_event = stop;                           // (recorded behind the kernel by the launch itself: rollout_lat.inc)
    const hipError_t e = launch_lat_e<kEpsPhilox>(p, s);
    g_lat_stop_event = nullptr;
    return e;
}
}  // namespace bn
