// rollout_lat_self_kt2.hip -- the ONE-LAUNCH latency kernel (rollout_lat.inc, mode 1) for noise source kEpsKT2: a synchronous
// forward() whose own tail -- merge, U*, first-action mailbox, X*, weights -- is a workgroup of the rollout launch.  A translation unit of its
// own: the kernel of the dependent-solve chains (mode 0, rollout_role_*.hip) stays what it was.
#define BN_ROLE_EPS kEpsKT2
#define BN_ROLE_REF false
#define BN_LAT_MODE 1
#include "mppi_device.h"
#include "rollout_lat.inc"

namespace bn {
hipError_t launch_rollout_lat_self_kt2(const SolveParams &p, hipStream_t s) { return launch_lat_e<kEpsKT2>(p, s); }
}  // namespace bn
