"""citylearn_amd -- MI355X-native vectorised CityLearn step engine.

The package holds only what the hot path needs: the HIP kernels + C-ABI (`csrc/`, `include/citylearn_amd.h`), the
schema loader that packs device tables, the ctypes engine, and the host-side mirror of the reference's
`CityLearnEnv` / `RewardFunction` / `CostFunction` interfaces.
"""
__version__ = '0.1.0'

from .schema import load_district, DistrictSpec  # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so that `import citylearn_amd` works in loader-only contexts
    if name == 'CityLearnEnv':
        from .citylearn import CityLearnEnv
        return CityLearnEnv
    if name == 'VectorCityLearnEnv':
        from .vector_env import VectorCityLearnEnv
        return VectorCityLearnEnv
    if name == 'StepEngine':
        from .engine import StepEngine
        return StepEngine
    raise AttributeError(name)
