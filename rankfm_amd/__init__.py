"""rankfm_amd -- MI355X-native RankFM training engine (drop-in for etlundquist/rankfm's hot path).

    from rankfm_amd import RankFM
    from rankfm_amd.evaluation import hit_rate

The BPR/WARP SGD loop, predict and recommend run as hand-written HIP kernels for gfx950 behind the C ABI in
include/rankfm_hip.h; there is no CPU fallback (importing works anywhere, training needs the GPU).
"""
from ._rankfm import DEFAULT_ENGINE, REFERENCE_ENGINE, EngineOptions, UserItemsCSR   # noqa: F401
from .rankfm import RankFM   # noqa: F401

__version__ = "0.1.0"
