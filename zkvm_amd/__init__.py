"""zkvm_amd -- MI355X (gfx950) back end for the multiscalar-multiplication tail of
ZkVM / Bulletproofs-R1CS verification.

The product is `lib/libzkgpu.so` (hand-written HIP, C ABI in include/zkgpu.h);
this package is the thin host-side mirror used by tests and bench.py.  There is
no CPU fallback: importing works anywhere, but creating a `Context` without the
built library or without a GPU raises.
"""
# GPU_MAX_HW_QUEUES (the HIP runtime's number of hardware queues, read once when it starts) is the library's business:
# zkgpu_init sets it when it is unset and the runtime has not started yet, and remembers when it came too late
# (Context.queue_info, BlockVerifier.queue_info).  A process that uses HIP before its first Context -- torch.cuda, say --
# exports the variable itself, before that (bench.py does, at its very top).

from .native import Context, PointSet, ZkGpuError, lib_path, load_library  # noqa: F401

__all__ = ["Context", "PointSet", "ZkGpuError", "lib_path", "load_library"]
