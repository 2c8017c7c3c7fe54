// rollout_lat_self_t2k.hip -- the ONE-LAUNCH latency kernel (rollout_lat.inc, mode 1) for noise source kEpsT2K: a synchronous
// forward() whose own tail -- merge, U*, first-action mailbox, X*, weights -- is a workgroup of the rollout launch.  A translation unit of its
// own: the kernel of the dependent-solve chains (mode 0, rollout_role_*.hip) stays what it was.
#define BN_ROLE_EPS kEpsT2K
#define BN_ROLE_REF false
#define BN_LAT_MODE 1
#include "mppi_device.h"
#include "rollout_lat.inc"

namespace bn {
hipError_t launch_rollout_lat_self_t2k(const SolveParams &p, hipStream_t s) { return launch_lat_e<kEpsT2K>(p, s); }
}  // namespace bn
