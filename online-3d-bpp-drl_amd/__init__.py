"""MI355X-native vectorised 3D bin-packing environment (hot path of alexfrom0815/Online-3D-BPP-DRL).

Import as `bpp_amd` (the loader module at the repository root); the directory name itself is not a
valid Python identifier."""
from . import _lib, sequences  # noqa: F401
from ._lib import build  # noqa: F401
from .factory import make_pool, make_vec_envs  # noqa: F401
from .masks import (batched_mask_from_hmap, batched_mask_from_obs, batched_window_masks,  # noqa: F401
                    get_possible_position, get_rotation_mask, masked_act, masked_evaluate)
from .spaces import Box, Discrete  # noqa: F401
from .stats import EpisodeStats, shard_range  # noqa: F401
from .vec_env import BppVecEnv, LazyInfos, StepTensors  # noqa: F401

__all__ = ["BppVecEnv", "LazyInfos", "StepTensors", "Box", "Discrete", "batched_mask_from_obs",
           "batched_mask_from_hmap", "batched_window_masks", "get_possible_position", "get_rotation_mask", "build", "sequences", "EpisodeStats", "shard_range", "make_vec_envs", "make_pool", "masked_act", "masked_evaluate"]
