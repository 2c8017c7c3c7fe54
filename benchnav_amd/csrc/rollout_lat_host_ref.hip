// rollout_lat_host_ref.hip -- the host-paced latency kernel in the REFERENCE's operation order (BN_FLAG_REFERENCE_ORDER); see
// rollout_lat_host.hip.  Compiled without the SLP vectoriser like the other reference-order units (benchnav_amd/build.py).
#define BN_ROLE_EPS kEpsPhilox
#define BN_ROLE_REF true
#define BN_LAT_MODE 2
#include "mppi_device.h"
#include "rollout_lat.inc"

namespace bn {
hipError_t launch_rollout_lat_host_ref(const SolveParams &p, hipStream_t s, hipEvent_t stop)
{
    g_lat_stop_event = stop;                           // (recorded behind the kernel by the launch itself: rollout_lat.inc)
    const hipError_t e = launch_lat_e<kEpsPhilox>(p, s);
    g_lat_stop_event = nullptr;
    return e;
}
}  // namespace bn
