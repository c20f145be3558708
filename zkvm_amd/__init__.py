"""zkvm_amd -- MI355X (gfx950) back end for the multiscalar-multiplication tail of
ZkVM / Bulletproofs-R1CS verification.

The product is `lib/libzkgpu.so` (hand-written HIP, C ABI in include/zkgpu.h);
this package is the thin host-side mirror used by tests and bench.py.  There is
no CPU fallback: importing works anywhere, but creating a `Context` without the
built library or without a GPU raises.
"""
import os as _os

# batches in flight use one light stream each (zkgpu_ctx_fork): let the HIP runtime give them hardware
# queues of their own (default 4).  Only effective when set before the runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .native import Context, PointSet, ZkGpuError, lib_path, load_library  # noqa: F401

__all__ = ["Context", "PointSet", "ZkGpuError", "lib_path", "load_library"]
