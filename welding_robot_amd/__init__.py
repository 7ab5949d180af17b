"""welding_robot_amd -- MI355X (gfx950) implementation of the ACS planning hot path of
mhsitu/welding_robot: voxelise -> rank-based ant-colony path search -> ACS-TSP seam ordering.

    csrc/            hand-written HIP kernels (*_kernels.hpp) + the C ABI (weldacs.hip, host_*.inc) -> lib/libweldacs.so
    include/core/    drop-in C++ headers with the reference's class names, written on the C ABI
    api.py           ctypes/numpy marshalling used by tests/ and bench.py
    build.py         in-tree hipcc build

The library has no CPU compute path; see DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
