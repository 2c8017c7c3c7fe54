// rollout_role_ref_kt2.hip -- the role-split rollout kernel and its latency variant in the REFERENCE's operation order
// (BN_FLAG_REFERENCE_ORDER) for noise source kEpsKT2 (see rollout_role.inc).  Compiled without the SLP vectoriser like
// rollout_wave_ref.hip (benchnav_amd/build.py).
#define BN_ROLE_EPS kEpsKT2
#define BN_ROLE_LAUNCHER launch_rollout_role_ref_kt2
#define BN_ROLE_REF true
#include "rollout_role.inc"
