"""mr_rl_amd -- MI355X-native MR_env.step()/MR_simulator hot path."""
from .config import MRConfig  # noqa: F401
from .env import MR_Env  # noqa: F401
from .vec_env import MRVecEnv  # noqa: F401

__all__ = ["MRConfig", "MR_Env", "MRVecEnv"]
