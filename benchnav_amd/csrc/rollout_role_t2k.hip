// rollout_role_t2k.hip -- the role-split rollout kernel for noise source kEpsT2K (see rollout_role.inc).
#define BN_ROLE_EPS kEpsT2K
#define BN_ROLE_LAUNCHER launch_rollout_role_t2k
#include "rollout_role.inc"
