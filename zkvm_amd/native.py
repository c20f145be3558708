"""ctypes binding of libzkgpu.so (include/zkgpu.h).

Mirrors, on the host side, the two reference call sites this library replaces
(upstream Rust names; /root/reference holds no source):

  Context.msm(scalars, points)           ~ RistrettoPoint::vartime_multiscalar_mul(..).compress()
  Context.verify_batch(scalars, points,  ~ for each proof: optional_multiscalar_mul(..)
                       offsets)             .map(|p| p.is_identity())  (tail of r1cs::Verifier::verify)

Errors follow the reference's "Result, never panic on malformed input":
an undecodable point raises ZkGpuError(EINVALID_POINT, index) from `msm` and
clears the accept bit in `verify_batch`.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional, Sequence, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, EINVAL, EINVALID_POINT, EHIP, ENOMEM, ENODEVICE, ENOCOMM = 0, -1, -2, -3, -4, -5, -6


class ZkGpuError(RuntimeError):
    def __init__(self, code: int, msg: str, index: Optional[int] = None):
        super().__init__(f"zkgpu error {code}: {msg}" + (f" (index {index})" if index is not None else ""))
        self.code = code
        self.index = index


def lib_path() -> str:
    # ZKGPU_LIB: another build of the same library (A/B experiments); never a different implementation
    return os.environ.get("ZKGPU_LIB") or os.path.join(HERE, "lib", "libzkgpu.so")


def _bind_hooks(lib) -> None:
    """include/zkgpu_hooks.h: measurement, tuning and test hooks are NOT exports of the library; `zkgpu_hook(name)` hands out
    their addresses to a process that had ZKGPU_TEST_HOOKS=1 in its environment when it loaded the library (tests/conftest.py
    and bench.py set it), NULL to everybody else.  Bound here under their own names, so that `lib.zkgpu_set_...` reads as it
    would for an export; without the variable the attribute is a function that raises."""
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_char_p
    lib.zkgpu_hook.restype = C.c_void_p
    lib.zkgpu_hook.argtypes = [C.c_char_p]
    table = {
        "zkgpu_set_group_size": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_locate_mode": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_horner_mode": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_transcript_mode": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_static_parts": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_locate_parts": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_tail_mode": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_window_bits": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_prover_mode": (C.c_int, [vp, C.c_int]),
        "zkgpu_set_serial": (C.c_int, [vp, C.c_int]),
        "zkgpu_measure_hbm_copy": (C.c_int, [vp, sz, C.c_int, C.POINTER(C.c_double)]),
        "zkgpu_profile_enable": (C.c_int, [vp, C.c_int]),
        "zkgpu_profile_reset": (None, [vp]),
        "zkgpu_profile_count": (C.c_int, [vp]),
        "zkgpu_profile_get": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
        "zkgpu_last_window_bits": (C.c_int, [vp]),
        "zkgpu_last_bucket_adds": (C.c_uint64, [vp]),
        "zkgpu_verifier_lane": (vp, [vp, C.c_int]),
        "zkgpu_debug_arith": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
        "zkgpu_debug_coop_selftest": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), sz]),
        "zkgpu_debug_read": (C.c_longlong, [vp, C.c_char_p, vp, sz]),
        "zkgpu_cloak_plan_layout": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
        "zkgpu_debug_force_regroup": (C.c_longlong, [vp, C.c_int]),
        "zkgpu_debug_comm_mock": (C.c_longlong, [vp, C.c_int, u8p, sz]),
        "zkgpu_debug_fail_after": (C.c_longlong, [vp, C.c_longlong, C.POINTER(C.c_longlong)]),
    }
    for name, (restype, argtypes) in table.items():
        addr = lib.zkgpu_hook(name.encode())
        if addr:
            fn = C.CFUNCTYPE(restype, *argtypes)(addr)
        else:
            def fn(*_a, _name=name):
                raise ZkGpuError(EINVAL, "%s is a hook (include/zkgpu_hooks.h), not an export: set ZKGPU_TEST_HOOKS=1 in the environment "
                                         "before the library is loaded" % _name)
        setattr(lib, name, fn)


def load_library():
    """Load libzkgpu.so.  Fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ZkGpuError(ENODEVICE, f"{path} is missing: run `python -m zkvm_amd.build` (there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_char_p
    lib.zkgpu_abi_version.restype = C.c_int
    lib.zkgpu_strerror.restype = C.c_char_p
    lib.zkgpu_strerror.argtypes = [C.c_int]
    lib.zkgpu_last_error.restype = C.c_char_p
    lib.zkgpu_last_error.argtypes = [vp]
    lib.zkgpu_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.zkgpu_destroy.argtypes = [vp]
    lib.zkgpu_destroy.restype = None
    lib.zkgpu_msm.argtypes = [vp, u8p, u8p, sz, u8p, C.POINTER(sz)]
    lib.zkgpu_msm_dev.argtypes = [vp, vp, vp, sz, u8p, C.POINTER(sz)]
    lib.zkgpu_verify_batch.argtypes = [vp, u8p, u8p, C.POINTER(C.c_uint64), sz, u8p]
    lib.zkgpu_verify_batch_dev.argtypes = [vp, vp, vp, vp, sz, sz, u8p]
    lib.zkgpu_pointset_create.argtypes = [vp, u8p, sz, C.POINTER(vp)]
    lib.zkgpu_pointset_destroy.argtypes = [vp]
    lib.zkgpu_pointset_destroy.restype = None
    lib.zkgpu_pointset_size.argtypes = [vp]
    lib.zkgpu_pointset_size.restype = sz
    lib.zkgpu_verify_batch_ps.argtypes = [vp, vp, sz, u8p, u8p, C.POINTER(C.c_uint64), u8p, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint64), u8p]
    lib.zkgpu_verify_batch_ps_dev.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, sz, u8p]
    lib.zkgpu_pointset_build_tables.argtypes = [vp, vp, C.c_int]
    lib.zkgpu_pointset_table_bytes.argtypes = [vp]
    lib.zkgpu_choose_table_bits.argtypes = [vp, sz]
    lib.zkgpu_pointset_table_bits.argtypes = [vp]
    lib.zkgpu_pointset_table_bytes.restype = sz
    lib.zkgpu_cloak_prove_batch.argtypes = [vp, vp, sz, sz, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), u8p, u8p, C.c_int,
                                            u8p, u8p, sz, C.POINTER(sz)]
    lib.zkgpu_msm_ps_batch.argtypes = [vp, vp, sz, u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), u8p]
    lib.zkgpu_decode_check.argtypes = [vp, u8p, sz, u8p]
    lib.zkgpu_cloak_verify_batch.argtypes = [vp, vp, sz, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u8p, u8p,
                                             C.POINTER(C.c_uint64), u8p, u8p, C.c_int]
    lib.zkgpu_cloak_plan_create.argtypes = [vp, C.c_uint32, C.c_uint32, sz, C.POINTER(vp)]
    lib.zkgpu_cloak_plan_destroy.argtypes = [vp]
    lib.zkgpu_cloak_plan_destroy.restype = None
    lib.zkgpu_cloak_plan_info.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 5
    lib.zkgpu_cloak_verify_batch_gpu.argtypes = [vp, vp, vp, sz, u8p, u8p, sz, u8p, u8p]
    lib.zkgpu_cloak_verify_batch_gpu_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, u8p]
    lib.zkgpu_ctx_fork.argtypes = [vp, C.POINTER(vp)]
    lib.zkgpu_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.zkgpu_free.argtypes = [vp, vp]
    lib.zkgpu_upload.argtypes = [vp, vp, u8p, sz]
    lib.zkgpu_cloak_verify_submit_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp]
    lib.zkgpu_cloak_verify_submit.argtypes = [vp, vp, vp, sz, u8p, u8p, sz, u8p]
    lib.zkgpu_verify_batch_ps_submit_dev.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, sz]
    lib.zkgpu_verify_wait.argtypes = [vp, u8p]
    lib.zkgpu_cloak_prepare_batch.argtypes = [sz, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u8p, u8p,
                                              C.POINTER(C.c_uint64), u8p, C.c_int, u8p, u8p, C.POINTER(C.c_uint64), sz,
                                              u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), sz, u8p]
    lib.zkgpu_msm_batch.argtypes = [vp, u8p, u8p, C.POINTER(C.c_uint64), sz, u8p, u8p]
    lib.zkgpu_hash_to_points.argtypes = [vp, u8p, sz, u8p]
    lib.zkgpu_pedersen_gens.argtypes = [vp, u8p, u8p]
    lib.zkgpu_bulletproof_gens.argtypes = [vp, sz, C.c_uint32, u8p, u8p]
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.zkgpu_verifier_create.argtypes = [vp, vp, sz, C.c_int, C.POINTER(vp)]
    lib.zkgpu_verifier_destroy.argtypes = [vp]
    lib.zkgpu_verifier_destroy.restype = None
    lib.zkgpu_verifier_set_chunk.argtypes = [vp, sz]
    lib.zkgpu_verifier_lanes.argtypes = [vp]
    lib.zkgpu_verifier_set_merge.argtypes = [vp, sz]
    lib.zkgpu_verifier_reserve.argtypes = [vp, C.c_uint32, C.c_uint32, sz]
    lib.zkgpu_verifier_submit_dev.argtypes = [vp, C.c_uint32, C.c_uint32, sz, vp, vp, sz, vp, C.POINTER(C.c_uint64)]
    lib.zkgpu_verifier_wait.argtypes = [vp, C.c_uint64, u8p]
    lib.zkgpu_verifier_last_error.argtypes = [vp]
    lib.zkgpu_verifier_last_error.restype = C.c_char_p
    lib.zkgpu_verifier_verify.argtypes = [vp, sz, u32p, u32p, u8p, u8p, u64p, u8p, u8p]
    lib.zkgpu_txblock_create.argtypes = [vp, sz, u32p, u32p, u8p, u8p, u64p, u8p, C.POINTER(vp)]
    lib.zkgpu_txblock_destroy.argtypes = [vp]
    lib.zkgpu_txblock_destroy.restype = None
    lib.zkgpu_txblock_size.argtypes = [vp]
    lib.zkgpu_txblock_size.restype = sz
    lib.zkgpu_txblock_shapes.argtypes = [vp]
    lib.zkgpu_txblock_shapes.restype = sz
    lib.zkgpu_verifier_verify_block.argtypes = [vp, vp, u8p]
    lib.zkgpu_verifier_block_start.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    lib.zkgpu_verifier_block_finish.argtypes = [vp, C.c_uint64, u8p]
    lib.zkgpu_cloak_msm_terms.argtypes = [C.c_uint32, C.c_uint32]
    lib.zkgpu_cloak_msm_terms.restype = C.c_uint64
    lib.zkgpu_shard_cuts.argtypes = [sz, u32p, u32p, C.c_int, u64p]
    lib.zkgpu_comm_unique_id.argtypes = [u8p]
    lib.zkgpu_comm_create.argtypes = [vp, C.c_int, C.c_int, u8p, C.POINTER(vp)]
    lib.zkgpu_comm_destroy.argtypes = [vp]
    lib.zkgpu_comm_destroy.restype = None
    lib.zkgpu_comm_rank.argtypes = [vp]
    lib.zkgpu_comm_world.argtypes = [vp]
    lib.zkgpu_comm_allgather.argtypes = [vp, u8p, sz, u8p]
    lib.zkgpu_comm_allgather_bitmap.argtypes = [vp, u64p, u8p, C.c_int, u8p]
    lib.zkgpu_verifier_verify_sharded.argtypes = [vp, vp, sz, u32p, u32p, u8p, u8p, u64p, u8p, u8p]
    lib.zkgpu_tx_verify_batch.argtypes = [vp, sz, u8p, u64p, C.c_int, u8p, u8p]
    lib.zkgpu_verifier_set_tx_format.argtypes = [vp, C.c_int]
    lib.zkgpu_tx_verify_submit.argtypes = [vp, sz, u8p, u64p, C.c_int, C.POINTER(C.c_uint64)]
    lib.zkgpu_tx_verify_wait.argtypes = [vp, C.c_uint64, u8p, u8p]
    lib.zkgpu_tx_verify_stats.argtypes = [vp, u64p]
    lib.zkgpu_verifier_queue_info.argtypes = [vp, C.POINTER(C.c_int)]
    lib.zkgpu_ctx_queue_info.argtypes = [vp, C.POINTER(C.c_int)]
    lib.zkgpu_verifier_set_tx_chunk.argtypes = [vp, sz]
    lib.zkgpu_verifier_set_tx_statements_kept.argtypes = [vp, sz]
    lib.zkgpu_verifier_submit_many_dev.argtypes = [vp, C.c_uint32, C.c_uint32, sz, sz, vp, vp, sz, vp, vp]
    lib.zkgpu_verifier_submit.argtypes = [vp, C.c_uint32, C.c_uint32, sz, C.c_char_p, C.c_char_p, sz, C.c_char_p, C.POINTER(C.c_uint64)]
    lib.zkgpu_verifier_submit_many.argtypes = [vp, C.c_uint32, C.c_uint32, sz, sz, vp, vp, sz, vp, vp]
    lib.zkgpu_r1cs_plan_create.argtypes = [vp, vp, sz, C.POINTER(vp)]
    lib.zkgpu_r1cs_plan_destroy.argtypes = [vp]
    lib.zkgpu_r1cs_plan_destroy.restype = None
    lib.zkgpu_r1cs_verify_batch_gpu.argtypes = [vp, vp, vp, sz, u8p, u8p, sz, u8p, u8p]
    lib.zkgpu_r1cs_verify_submit.argtypes = [vp, vp, vp, sz, u8p, u8p, sz, u8p]
    lib.zkgpu_r1cs_verify_submit_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp]
    lib.zkgpu_r1cs_verify_batch.argtypes = [vp, vp, vp, sz, sz, u8p, u8p, sz, u8p, u8p, C.c_int]
    lib.zkgpu_r1cs_prove_batch.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), sz, sz, u8p, u8p, u8p, sz, u8p, C.c_int, u8p, u8p, sz,
                                           C.POINTER(sz)]
    lib.zkgpu_runtime_hint.argtypes = [C.c_char_p, sz]
    _bind_hooks(lib)
    _LIB = lib
    return lib


HINT_APPLY, HINT_PRESENT, HINT_LATE = 0, 1, 2


def runtime_hint(apply: bool = True) -> Tuple[int, str]:
    """zkgpu_runtime_hint: what the host should export before its first HIP call ("GPU_MAX_HW_QUEUES=18").  The library
    never edits the environment; THIS is the host side doing it (os.environ, i.e. the host language's own setenv) when the
    answer is HINT_APPLY.  Returns (answer, "NAME=value").  Context() calls it, so a Python host that creates its Context
    before it touches torch.cuda / HIP needs nothing else; one that touches HIP first calls it first (bench.py does)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.zkgpu_runtime_hint(buf, len(buf))
    text = buf.value.decode()
    if rc == HINT_APPLY and apply:
        for pair in text.split("\n"):
            name, _, value = pair.partition("=")
            if name:
                os.environ[name] = value
    return rc, text


def _ptr(x) -> int:
    """Device pointer of a torch tensor (or a raw int)."""
    return x if isinstance(x, int) else x.data_ptr()


def _u64arr(xs: Sequence[int]):
    return (C.c_uint64 * len(xs))(*xs)


class PointSet:
    """A set of points decompressed once and resident on the device (generators)."""

    def __init__(self, ctx: "Context", points: bytes):
        assert len(points) % 32 == 0
        self.ctx = ctx
        self.h = C.c_void_p()
        rc = ctx.lib.zkgpu_pointset_create(ctx.h, points, len(points) // 32, C.byref(self.h))
        ctx._check(rc)

    def __len__(self) -> int:
        return int(self.ctx.lib.zkgpu_pointset_size(self.h))

    def table_bits(self) -> int:
        """zkgpu_pointset_table_bits: the window width of the tables in use (0: none)"""
        return int(self.ctx.lib.zkgpu_pointset_table_bits(self.h))

    def build_tables(self, window_bits: int) -> int:
        """Fixed-base window tables (one-time; window_bits 0: the library chooses by capacity and free HBM); returns their size in bytes."""
        self.ctx._check(self.ctx.lib.zkgpu_pointset_build_tables(self.ctx.h, self.h, window_bits))
        return int(self.ctx.lib.zkgpu_pointset_table_bytes(self.h))

    def close(self) -> None:
        if self.h:
            self.ctx.lib.zkgpu_pointset_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU, one HIP stream.  Use one Context per process / per GPU."""

    def __init__(self, device: int = 0, _parent: Optional["Context"] = None, _borrowed: Optional[int] = None):
        self.lib = load_library()
        self.h = C.c_void_p()
        self.parent = _parent          # a fork shares its parent's chip-filling streams: keep the parent alive
        self._forks = weakref.WeakSet()
        self._pending_batch = 0
        self._borrowed = _borrowed is not None
        if self._borrowed:             # a context owned by a zkgpu_verifier: a view for the measurement hooks
            self.h = C.c_void_p(_borrowed)
            return
        if _parent is not None:
            rc = self.lib.zkgpu_ctx_fork(_parent.h, C.byref(self.h))
        else:
            runtime_hint()             # (the host's own setenv, before the first HIP call; a no-op afterwards)
            rc = self.lib.zkgpu_init(device, C.byref(self.h))
        if rc != OK:
            raise ZkGpuError(rc, self.lib.zkgpu_strerror(rc).decode() + " (libzkgpu needs a HIP device; no CPU fallback)")

    def to_device(self, data: bytes) -> int:
        """zkgpu_malloc + zkgpu_upload: a device buffer holding `data`; release with free_device."""
        p = C.c_void_p()
        self._check(self.lib.zkgpu_malloc(self.h, len(data), C.byref(p)))
        self._check(self.lib.zkgpu_upload(self.h, p, data, len(data)))
        return int(p.value)

    def free_device(self, d_ptr: int) -> None:
        self._check(self.lib.zkgpu_free(self.h, C.c_void_p(d_ptr)))

    def queue_info(self):
        """zkgpu_ctx_queue_info -> (the two pipeline streams run side by side: 1 / 0 / -1, GPU_MAX_HW_QUEUES or 0,
        1 when the HIP runtime had started before the variable was set)"""
        out = (C.c_int * 3)()
        self._check(self.lib.zkgpu_ctx_queue_info(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def measure_hbm_copy(self, nbytes: int = 1 << 30, iters: int = 10) -> float:
        """zkgpu_measure_hbm_copy: achievable HBM bandwidth of a streaming copy, GB/s (read + written)"""
        out = C.c_double(0.0)
        self._check(self.lib.zkgpu_measure_hbm_copy(self.h, nbytes, iters, C.byref(out)))
        return float(out.value)

    def set_group_size(self, group: int) -> None:
        """zkgpu_set_group_size: transactions per group check of the whole-proof paths (1 = none)."""
        self._check(self.lib.zkgpu_set_group_size(self.h, group))

    def set_transcript_mode(self, mode: int) -> None:
        """zkgpu_set_transcript_mode: 0 automatic, 1 one lane per transaction, 2 one wavefront per transaction."""
        self._check(self.lib.zkgpu_set_transcript_mode(self.h, mode))

    def coop_selftest(self, a, b, addr, states):
        """zkgpu_debug_coop_selftest -> (8 x 64 words of primitive outputs, permuted states)"""
        inp = (C.c_uint32 * 192)(*(list(a) + list(b) + list(addr)))
        out = (C.c_uint32 * 512)()
        flat = [w for st in states for w in st]
        st = (C.c_uint64 * max(len(flat), 1))(*flat)
        self._check(self.lib.zkgpu_debug_coop_selftest(self.h, inp, out, st, len(states)))
        return [list(out[64 * i: 64 * i + 64]) for i in range(8)], [list(st[25 * i: 25 * i + 25]) for i in range(len(states))]

    def debug_arith(self, op: int, a: bytes, b: bytes) -> bytes:
        """zkgpu_debug_arith: one operation of the field / scalar layers on len(a) / 32 elements"""
        n = len(a) // 32
        assert len(a) == len(b) == 32 * n and n > 0
        out = C.create_string_buffer(32 * n)
        self._check(self.lib.zkgpu_debug_arith(self.h, op, a, b, out, n))
        return out.raw

    def set_prover_mode(self, mode: int) -> None:
        """zkgpu_set_prover_mode: 0 the whole proof on the device, 1 host threads in lockstep, 16 + S: on the device in S slices."""
        self._check(self.lib.zkgpu_set_prover_mode(self.h, mode))

    def set_horner_mode(self, mode: int) -> None:
        """zkgpu_set_horner_mode: 0 automatic, 1 one Horner chain per transaction, 2 one per group (+ failed groups' transactions)."""
        self._check(self.lib.zkgpu_set_horner_mode(self.h, mode))

    def set_locate_mode(self, mode: int) -> None:
        """zkgpu_set_locate_mode: failed groups -- 0 automatic, 1 re-check every transaction, 2 locate the culprit, 3 locate with the locating sums formed up front."""
        self._check(self.lib.zkgpu_set_locate_mode(self.h, mode))

    def force_regroup(self, on: bool) -> int:
        """zkgpu_debug_force_regroup: test hook for the ungrouped re-run; returns the re-runs so far."""
        return int(self.lib.zkgpu_debug_force_regroup(self.h, 1 if on else 0))

    def set_serial(self, on: bool) -> None:
        """zkgpu_set_serial: one stream for the whole batch (measurement aid)."""
        self._check(self.lib.zkgpu_set_serial(self.h, 1 if on else 0))

    def fork(self) -> "Context":
        """zkgpu_ctx_fork: own workspace + light stream, the parent's heavy streams (batches in flight)."""
        f = Context(_parent=self)
        self._forks.add(f)
        return f

    def verify_batch_ps_submit_dev(self, ps: "PointSet", batch: int, d_dyn_scalars, d_dyn_points, d_dyn_offsets,
                                   n_dyn: int, d_static_scalars, d_static_index, d_static_offsets, n_static: int) -> None:
        self._check(self.lib.zkgpu_verify_batch_ps_submit_dev(
            self.h, ps.h, batch, _ptr(d_dyn_scalars), _ptr(d_dyn_points), _ptr(d_dyn_offsets), n_dyn,
            _ptr(d_static_scalars), _ptr(d_static_index) if d_static_index is not None else None,
            _ptr(d_static_offsets), n_static))
        self._pending_batch = batch

    def verify_wait(self) -> bytes:
        """zkgpu_verify_wait: the accept bitmap of the batch submitted last on this context."""
        n = (self._pending_batch + 7) // 8
        bm = C.create_string_buffer(max(n, 1))
        self._check(self.lib.zkgpu_verify_wait(self.h, bm))
        return bm.raw[:n]

    def close(self) -> None:
        if self.h and self._borrowed:
            self.h = C.c_void_p()
            return
        if self.h:
            for f in list(self._forks):     # forks borrow this context's streams: they go first
                f.close()
            self.lib.zkgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, index: Optional[int] = None) -> None:
        if rc != OK:
            detail = self.lib.zkgpu_last_error(self.h).decode() if rc in (EHIP, ENOMEM, EINVAL, ENOCOMM) else ""
            raise ZkGpuError(rc, self.lib.zkgpu_strerror(rc).decode() + (": " + detail if detail else ""), index)

    # ---- single MSM ---------------------------------------------------------
    def msm(self, scalars: bytes, points: bytes) -> bytes:
        n = len(scalars) // 32
        assert len(scalars) == 32 * n and len(points) == 32 * n
        out = C.create_string_buffer(32)
        bad = C.c_size_t(0)
        rc = self.lib.zkgpu_msm(self.h, scalars, points, n, out, C.byref(bad))
        self._check(rc, bad.value if rc == EINVALID_POINT else None)
        return out.raw

    def msm_dev(self, d_scalars, d_points, n: int) -> bytes:
        out = C.create_string_buffer(32)
        bad = C.c_size_t(0)
        rc = self.lib.zkgpu_msm_dev(self.h, _ptr(d_scalars), _ptr(d_points), n, out, C.byref(bad))
        self._check(rc, bad.value if rc == EINVALID_POINT else None)
        return out.raw

    # ---- batch of MSM == identity checks ---------------------------------------
    def verify_batch(self, scalars: bytes, points: bytes, offsets: Sequence[int]) -> bytes:
        batch = len(offsets) - 1
        bm = C.create_string_buffer(max((batch + 7) // 8, 1))
        rc = self.lib.zkgpu_verify_batch(self.h, scalars, points, _u64arr(offsets), batch, bm)
        self._check(rc)
        return bm.raw[: (batch + 7) // 8]

    def verify_batch_dev(self, d_scalars, d_points, d_offsets, batch: int, n_terms: int) -> bytes:
        bm = C.create_string_buffer(max((batch + 7) // 8, 1))
        rc = self.lib.zkgpu_verify_batch_dev(self.h, _ptr(d_scalars), _ptr(d_points), _ptr(d_offsets), batch, n_terms, bm)
        self._check(rc)
        return bm.raw[: (batch + 7) // 8]

    def verify_batch_ps(self, ps: PointSet, dyn_scalars: bytes, dyn_points: bytes, dyn_offsets: Sequence[int],
                        static_scalars: bytes, static_offsets: Sequence[int],
                        static_index: Optional[Sequence[int]] = None) -> bytes:
        batch = len(dyn_offsets) - 1
        assert len(static_offsets) == batch + 1
        bm = C.create_string_buffer(max((batch + 7) // 8, 1))
        idx = (C.c_uint32 * len(static_index))(*static_index) if static_index is not None else None
        rc = self.lib.zkgpu_verify_batch_ps(self.h, ps.h, batch, dyn_scalars, dyn_points, _u64arr(dyn_offsets),
                                            static_scalars, idx, _u64arr(static_offsets), bm)
        self._check(rc)
        return bm.raw[: (batch + 7) // 8]

    def verify_batch_ps_dev(self, ps: PointSet, batch: int, d_dyn_scalars, d_dyn_points, d_dyn_offsets, n_dyn: int,
                            d_static_scalars, d_static_index, d_static_offsets, n_static: int) -> bytes:
        bm = C.create_string_buffer(max((batch + 7) // 8, 1))
        rc = self.lib.zkgpu_verify_batch_ps_dev(self.h, ps.h, batch, _ptr(d_dyn_scalars), _ptr(d_dyn_points),
                                                _ptr(d_dyn_offsets), n_dyn, _ptr(d_static_scalars),
                                                _ptr(d_static_index) if d_static_index is not None else None,
                                                _ptr(d_static_offsets), n_static, bm)
        self._check(rc)
        return bm.raw[: (batch + 7) // 8]

    def msm_ps_batch(self, ps: PointSet, scalars: bytes, offsets: Sequence[int],
                     index: Optional[Sequence[int]] = None) -> bytes:
        """zkgpu_msm_ps_batch: values (32-byte encodings) of MSMs over a resident set with tables."""
        batch = len(offsets) - 1
        out = C.create_string_buffer(max(32 * batch, 1))
        idx = (C.c_uint32 * max(len(index), 1))(*index) if index is not None else None
        self._check(self.lib.zkgpu_msm_ps_batch(self.h, ps.h, batch, scalars, idx, _u64arr(list(offsets)), out))
        return out.raw[: 32 * batch]

    def msm_batch(self, scalars: bytes, points: bytes, offsets: Sequence[int]) -> Tuple[bytes, bytes]:
        """-> (batch x 32-byte encodings, ok bitmap)"""
        batch = len(offsets) - 1
        out = C.create_string_buffer(max(32 * batch, 1))
        bm = C.create_string_buffer(max((batch + 7) // 8, 1))
        self._check(self.lib.zkgpu_msm_batch(self.h, scalars, points, _u64arr(offsets), batch, out, bm))
        return out.raw[: 32 * batch], bm.raw[: (batch + 7) // 8]

    def hash_to_points(self, uniform: bytes) -> bytes:
        n = len(uniform) // 64
        out = C.create_string_buffer(max(32 * n, 1))
        self._check(self.lib.zkgpu_hash_to_points(self.h, uniform, n, out))
        return out.raw[: 32 * n]

    def pedersen_gens(self) -> Tuple[bytes, bytes]:
        b, bb = C.create_string_buffer(32), C.create_string_buffer(32)
        self._check(self.lib.zkgpu_pedersen_gens(self.h, b, bb))
        return b.raw, bb.raw

    def bulletproof_gens(self, capacity: int, party: int = 0) -> Tuple[bytes, bytes]:
        g, h = C.create_string_buffer(max(32 * capacity, 1)), C.create_string_buffer(max(32 * capacity, 1))
        self._check(self.lib.zkgpu_bulletproof_gens(self.h, capacity, party, g, h))
        return g.raw[: 32 * capacity], h.raw[: 32 * capacity]

    def decode_check(self, points: bytes) -> bytes:
        n = len(points) // 32
        ok = C.create_string_buffer(max(n, 1))
        self._check(self.lib.zkgpu_decode_check(self.h, points, n, ok))
        return ok.raw[:n]

    # ---- measurement hooks -------------------------------------------------------
    def profile(self, on: bool) -> None:
        self.lib.zkgpu_profile_enable(self.h, 1 if on else 0)

    def profile_reset(self) -> None:
        self.lib.zkgpu_profile_reset(self.h)

    def profile_read(self) -> Dict[str, Tuple[int, float]]:
        out = {}
        for i in range(self.lib.zkgpu_profile_count(self.h)):
            name, n, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self.lib.zkgpu_profile_get(self.h, i, C.byref(name), C.byref(n), C.byref(ms))
            out[name.value.decode()] = (int(n.value), float(ms.value))
        return out

    def debug_read(self, what: str, nbytes: int) -> bytes:
        """zkgpu_debug_read: an intermediate buffer of the last device-side preparation on this context."""
        buf = C.create_string_buffer(max(nbytes, 1))
        n = int(self.lib.zkgpu_debug_read(self.h, what.encode(), buf, nbytes))
        if n < 0:
            raise ZkGpuError(n, "zkgpu_debug_read(%s)" % what)
        return buf.raw[:n]

    def last_window_bits(self) -> int:
        return int(self.lib.zkgpu_last_window_bits(self.h))

    def set_static_parts(self, parts: int) -> None:
        self._check(self.lib.zkgpu_set_static_parts(self.h, parts))

    def set_locate_parts(self, parts: int) -> None:
        self._check(self.lib.zkgpu_set_locate_parts(self.h, parts))

    def set_tail_mode(self, mode: int) -> None:
        self._check(self.lib.zkgpu_set_tail_mode(self.h, mode))

    def set_window_bits(self, w: int) -> None:
        self._check(self.lib.zkgpu_set_window_bits(self.h, w))


class R1csDescStruct(C.Structure):
    """zkgpu_r1cs_desc"""
    _fields_ = [("transcript_label", C.c_char_p), ("n_commitments", C.c_uint32), ("n_multipliers_phase1", C.c_uint32),
                ("n_multipliers", C.c_uint32), ("n_challenges", C.c_uint32), ("challenge_labels", C.POINTER(C.c_char_p)),
                ("n_constraints", C.c_uint32), ("term_offsets", C.POINTER(C.c_uint64)), ("term_var_kind", C.POINTER(C.c_uint8)),
                ("term_var_index", C.POINTER(C.c_uint32)), ("term_coeff", C.c_char_p), ("term_challenge", C.POINTER(C.c_int32)),
                ("term_power", C.POINTER(C.c_uint32))]


class R1csDescription:
    """A constraint system as data (include/zkgpu.h, zkgpu_r1cs_desc).  constraints: list of constraints, each a list of
    terms (kind, index, coefficient, challenge, power): kind 0 committed, 1 / 2 / 3 left / right / out of a multiplier,
    4 the constant one; coefficient an int mod l; challenge -1 or the index of a second-phase challenge."""
    KIND_COMMITTED, KIND_LEFT, KIND_RIGHT, KIND_OUT, KIND_ONE = range(5)
    L = 2**252 + 27742317777372353535851937790883648493

    def __init__(self, label: bytes, n_commitments: int, n_phase1: int, n_multipliers: int, challenge_labels, constraints):
        self.label, self.m, self.n1, self.n = label, n_commitments, n_phase1, n_multipliers
        self.challenge_labels = list(challenge_labels)
        self.constraints = constraints
        offs, kinds, idx, coeff, chal, power = [0], [], [], bytearray(), [], []
        for con in constraints:
            for (k, i, c, ch, pw) in con:
                kinds.append(k); idx.append(i); coeff += int(c % self.L).to_bytes(32, "little"); chal.append(ch); power.append(pw)
            offs.append(len(kinds))
        nt = max(len(kinds), 1)
        self._keep = ((C.c_char_p * max(len(self.challenge_labels), 1))(*self.challenge_labels), (C.c_uint64 * len(offs))(*offs),
                      (C.c_uint8 * nt)(*kinds), (C.c_uint32 * nt)(*idx), bytes(coeff), (C.c_int32 * nt)(*chal), (C.c_uint32 * nt)(*power))
        k = self._keep
        self.struct = R1csDescStruct(label, self.m, self.n1, self.n, len(self.challenge_labels), k[0], len(constraints), k[1], k[2], k[3],
                                     k[4], k[5], k[6])


def shard_cuts(shapes: Sequence[Tuple[int, int]], world: int):
    """zkgpu_shard_cuts: contiguous shards of a block, balanced by multiscalar-multiplication terms (no GPU needed)."""
    lib = load_library()
    n = len(shapes)
    a = (C.c_uint32 * max(n, 1))(*[s[0] for s in shapes])
    b = (C.c_uint32 * max(n, 1))(*[s[1] for s in shapes])
    cuts = (C.c_uint64 * (world + 1))()
    rc = lib.zkgpu_shard_cuts(n, a, b, world, cuts)
    if rc != OK:
        raise ZkGpuError(rc, lib.zkgpu_strerror(rc).decode())
    return list(cuts)


class Comm:
    """zkgpu_comm: the RCCL communicator of this process's GPU (one process per GPU)."""

    def __init__(self, ctx: Context, rank: int, world: int, unique_id: Optional[bytes]):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(ctx.lib.zkgpu_comm_create(ctx.h, rank, world, unique_id, C.byref(self.h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.zkgpu_comm_unique_id(buf)
        if rc != OK:
            raise ZkGpuError(rc, lib.zkgpu_strerror(rc).decode())
        return buf.raw

    def allgather(self, local: bytes) -> bytes:
        out = C.create_string_buffer(max(len(local) * self.world, 1))
        self.ctx._check(self.ctx.lib.zkgpu_comm_allgather(self.h, local, len(local), out))
        return out.raw[: len(local) * self.world]

    def allgather_bitmap(self, cuts: Sequence[int], local_bitmap: bytes, local_status: int = 0) -> bytes:
        n = cuts[-1]
        out = C.create_string_buffer(max((n + 7) // 8, 1))
        self.ctx._check(self.ctx.lib.zkgpu_comm_allgather_bitmap(self.h, _u64arr(list(cuts)), local_bitmap, local_status, out))
        return out.raw[: (n + 7) // 8]

    def close(self) -> None:
        if self.h:
            self.ctx.lib.zkgpu_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
