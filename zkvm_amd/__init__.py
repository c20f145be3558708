"""zkvm_amd -- MI355X (gfx950) back end for the multiscalar-multiplication tail of
ZkVM / Bulletproofs-R1CS verification.

The product is `lib/libzkgpu.so` (hand-written HIP, C ABI in include/zkgpu.h);
this package is the thin host-side mirror used by tests and bench.py.  There is
no CPU fallback: importing works anywhere, but creating a `Context` without the
built library or without a GPU raises.
"""
# GPU_MAX_HW_QUEUES (the HIP runtime's number of hardware queues, read once when it starts) is the HOST's business: the library
# never edits the environment, it only recommends (zkgpu_runtime_hint).  `runtime_hint()` below is the Python host applying the
# recommendation through os.environ; Context() calls it, and the library remembers when it came too late (Context.queue_info,
# BlockVerifier.queue_info).  A process that uses HIP before its first Context -- torch.cuda, say -- calls runtime_hint()
# itself, before that (bench.py does, at its very top).

from .native import Context, PointSet, ZkGpuError, lib_path, load_library, runtime_hint  # noqa: F401

__all__ = ["Context", "PointSet", "ZkGpuError", "lib_path", "load_library", "runtime_hint"]
