"""optimaltextures_amd — MI355X-native sliced-optimal-transport inner loop of JCBrouwer/OptimalTextures.

Same call surface as the reference's hot path (optex.py::optimal_transport, random_rotation; histmatch.py::hist_match,
cdf_match, interp), implemented as hand-written HIP kernels for gfx950 behind the C ABI in include/optex.h."""
from .histmatch import cdf_match, hist_match, interp  # noqa: F401
from .optex import optimal_transport, random_rotation  # noqa: F401

__all__ = ["optimal_transport", "random_rotation", "hist_match", "cdf_match", "interp"]
