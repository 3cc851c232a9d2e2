"""spiral_amd -- MI355X-native Spiral server-answer path.

The compute lives in libspiral_gpu.so (hand-written HIP for gfx950 behind the C ABI of
include/spiral_gpu.h).  This package is the thin host-side mirror of the reference's function
interface for that path (names and argument meaning of src/spiral.cpp / src/poly.cpp / src/core.cpp),
used by the parity tests and the benchmark.  numpy uint64 arrays carry the reference layouts.
"""
from ._lib import PackShape, Params, Shape, SpiralGpuError, build, lib  # noqa: F401
from .ops import *  # noqa: F401,F403
from .pack import PackServer, fastMultiplyQueryByDatabaseDim1, get_pack_shape, pack  # noqa: F401
from .server import Server, first_dim_batch, run_query_batch, time_sweep_batch  # noqa: F401
