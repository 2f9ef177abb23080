"""`shadowing.path_shadowing` import path of the reference, served by shadowing_amd."""
from shadowing_amd.path_distance import *  # noqa: F401,F403
from shadowing_amd.path_distance import PathDistance, RelativeMSE  # noqa: F401
from shadowing_amd.path_embedding import (ArrayType, ContextManagerBase, CrossChannelContext, Foveal,  # noqa: F401
                                          Identity, ImputationContext, PathEmbedding, PredictionContext)
from shadowing_amd.path_shadowing import PathShadowing, select_cartesian_product  # noqa: F401
