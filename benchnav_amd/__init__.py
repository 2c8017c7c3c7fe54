"""benchnav_amd -- MI355X-native MPPI local planner for BenchNav (hot path only).

    from benchnav_amd import MPPI          # drop-in for src/planners/local_planners/mppi.py:MPPI
    from benchnav_amd import NativeMPPI    # numpy-level wrapper of the C ABI, B instances per call
    from benchnav_amd import BatchedPlanetaryEnv   # reset / step / collision_check of PlanetaryEnv for B environments on the GPU
"""
from .native import NativeMPPI  # noqa: F401


def __getattr__(name):
    if name == "MPPI":          # torch is imported only when the torch-facing classes are used
        from .mppi import MPPI
        return MPPI
    if name == "DWA":
        from .dwa import DWA
        return DWA
    if name == "BatchedPlanetaryEnv":
        from .env import BatchedPlanetaryEnv
        return BatchedPlanetaryEnv
    raise AttributeError(name)
