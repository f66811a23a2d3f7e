"""voxgraph_amd -- voxgraph's registration-cost and TSDF-integration inner loops
as hand-written HIP kernels for MI355X (gfx950), behind a C ABI.

The package is a thin ctypes view of voxgraph_amd/lib/libvoxgraph_amd.so
(include/voxgraph_amd.h); see DESIGN.md and INTEGRATION.md.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
