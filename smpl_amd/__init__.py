"""smpl_amd -- MI355X-native ARA* state-expansion engine behind smpl's plugin interfaces.

The product is the C-ABI shared library (include/smpl_amd.h, smpl_amd/libsmpl_amd.so: hand-written
gfx950 kernels + host lattice).  `capi` marshals numpy arrays to it; `scenes` builds the seeded
synthetic inputs of SURVEY.md section 8d.
"""
from . import build, scenes  # noqa: F401

__all__ = ["build", "scenes", "capi"]
