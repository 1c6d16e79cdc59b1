"""beacon_amd -- MI355X-native batched stepper for the solver hot path of jviquerat/beacon.

Batched envs (tensors on the GPU, one HIP launch per step):
    VecRayleigh, VecMixing, VecBurgers, VecShkadov, VecSloshing
Drop-in single-env mirrors of the reference classes:  beacon_amd.envs.{rayleigh, mixing, ...}
Multi-GPU replica sharding:  beacon_amd.dist.ShardedVecEnv

Importing the package does not touch the GPU; constructing an env does, and raises if the HIP
library or a ROCm device is missing (there is no CPU fallback for the solver path)."""
from .vec import Box, Discrete, VecBurgers, VecEnv, VecMixing, VecRayleigh, VecShkadov, VecSloshing  # noqa: F401
from .lorenz import lorenz  # noqa: F401
from .vortex import vortex  # noqa: F401

__version__ = "0.1.0"

VEC_ENVS = {"rayleigh-v0": VecRayleigh, "mixing-v0": VecMixing, "burgers-v0": VecBurgers,
            "shkadov-v0": VecShkadov, "sloshing-v0": VecSloshing}


def make_vec(env_id, batch, **kwargs):
    """make_vec("rayleigh-v0", 512, L=2.56, H=1.28) -- env ids as listed in the reference README."""
    return VEC_ENVS[env_id](batch, **kwargs)
