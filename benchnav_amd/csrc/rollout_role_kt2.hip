// rollout_role_kt2.hip -- the role-split rollout kernel for noise source kEpsKT2 (see rollout_role.inc).
#define BN_ROLE_EPS kEpsKT2
#define BN_ROLE_LAUNCHER launch_rollout_role_kt2
#include "rollout_role.inc"
