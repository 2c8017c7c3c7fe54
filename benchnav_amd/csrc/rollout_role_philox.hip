// rollout_role_philox.hip -- the role-split rollout kernel for noise source kEpsPhilox (see rollout_role.inc).
#define BN_ROLE_EPS kEpsPhilox
#define BN_ROLE_LAUNCHER launch_rollout_role_philox
#define BN_ROLE_OCCUPANCY 1
#include "rollout_role.inc"
