"""shadowing_amd -- the k-nearest-path scan of Path Shadowing Monte-Carlo on MI355X.

One hot path of RudyMorel/shadowing, rebuilt MI355X-first: PathShadowing.shadow()
with Identity + RelativeMSE + PredictionContext runs as hand-written gfx950 HIP
kernels behind the reference's own PathShadowing / PathEmbedding / PathDistance /
context plugin surface (a drop-in for that path and nothing else).
"""
from .averaging import DiscreteProba, Softmax, Uniform
from .path_distance import PathDistance, RelativeMSE
from .path_embedding import (ArrayType, ContextManagerBase, CrossChannelContext, Foveal, Identity,
                             ImputationContext, PathEmbedding, PredictionContext)
from .path_shadowing import PathShadowing, PendingShadow, select_cartesian_product
from .plotting import plot_closest, plot_shadow, plot_volatility
from .statistics import realized_variance

__all__ = [
    "ArrayType", "ContextManagerBase", "PredictionContext", "ImputationContext", "CrossChannelContext",
    "PathEmbedding", "Identity", "Foveal", "PathDistance", "RelativeMSE", "PathShadowing",
    "select_cartesian_product", "DiscreteProba", "Softmax", "Uniform", "realized_variance",
    "plot_closest", "plot_shadow", "plot_volatility",
]
__version__ = "0.1.0"
