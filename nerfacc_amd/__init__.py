"""nerfacc_amd -- MI355X-native ray marching and volumetric rendering behind nerfacc's API.

Drop-in for the hot path of nerfacc 0.5.3 (reference: leejaeyong7/nerfacc,
nerfacc/__init__.py:23-46): the same 20 public names with the same signatures.  The native
ops are hand-written HIP kernels for gfx950 in ``libnerfacc_hip.so`` (C ABI:
``include/nerfacc_hip.h``); they are loaded lazily on first use and there is no CPU fallback.
"""
from .data_specs import RayIntervals, RaySamples
from .estimators.occ_grid import OccGridEstimator
from .estimators.prop_net import PropNetEstimator
from .graphs import CapturedStep  # extension (hipGraph replay of fixed-shape steps); not part of the reference's 20 names
from .grid import ray_aabb_intersect, traverse_grids
from .pack import pack_info
from .pdf import importance_sampling, searchsorted
from .scan import exclusive_prod, exclusive_sum, inclusive_prod, inclusive_sum
from .version import __version__
from .volrend import (
    accumulate_along_rays,
    render_transmittance_from_alpha,
    render_transmittance_from_density,
    render_visibility_from_alpha,
    render_visibility_from_density,
    render_weight_from_alpha,
    render_weight_from_density,
    rendering,
)

__all__ = [
    "__version__",
    "inclusive_prod",
    "exclusive_prod",
    "inclusive_sum",
    "exclusive_sum",
    "pack_info",
    "render_visibility_from_alpha",
    "render_visibility_from_density",
    "render_weight_from_alpha",
    "render_weight_from_density",
    "render_transmittance_from_alpha",
    "render_transmittance_from_density",
    "accumulate_along_rays",
    "rendering",
    "importance_sampling",
    "searchsorted",
    "RayIntervals",
    "RaySamples",
    "ray_aabb_intersect",
    "traverse_grids",
    "OccGridEstimator",
    "PropNetEstimator",
]
