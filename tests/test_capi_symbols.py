"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol
include/zkgpu.h declares, and refuses to work without a GPU (no silent fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkgpu_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from zkvm_amd import build, native
    build.build()
    return native.load_library()


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_abi_version_and_strerror(lib):
    assert lib.zkgpu_abi_version() == 3
    assert lib.zkgpu_strerror(0) == b"ok"
    assert b"ristretto" in lib.zkgpu_strerror(-2)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zkvm_amd import Context, ZkGpuError
    with pytest.raises(ZkGpuError) as e:
        Context(0)
    assert e.value.code == -5


def test_product_never_touches_oracle():
    # the shipped package must not import, link or open anything under oracle/
    pkg = os.path.join(ROOT, "zkvm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("liboracle", "from oracle", "import oracle", "oracle/_build", "oracle.h"):
                    assert needle not in text, (f, needle)


def _hooks_declared():
    src = open(os.path.join(ROOT, "include", "zkgpu_hooks.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkgpu_[a-z0-9_]+)\s*\(", src)))


def test_hooks_are_not_exports_and_answer_only_to_a_process_that_asked_for_them():
    """VERDICT r04 "ABI sprawl" / ADVICE r03: the 25 measurement, tuning and test hooks (include/zkgpu_hooks.h) are not in
    the library's symbol table and not in zkgpu.h; `zkgpu_hook(name)` answers NULL unless ZKGPU_TEST_HOOKS=1 was in the
    environment BEFORE the library was loaded (fresh interpreters: the answer is read once per process), and the Python
    binding's attribute then raises instead of calling anything."""
    import subprocess
    import sys
    from zkvm_amd import build
    build.build()
    hooks = _hooks_declared()
    assert len(hooks) == 25 and "zkgpu_debug_fail_after" in hooks and "zkgpu_debug_comm_mock" in hooks and "zkgpu_profile_get" in hooks
    assert not set(hooks) & set(_declared())
    assert len(_declared()) <= 90
    nm = subprocess.run(["nm", "-D", "--defined-only", build.OUT], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines()}
    assert not exported & set(hooks)
    assert not [n for n in exported if "debug" in n or n.startswith("_ZN") and "zkgpu" in n], "internals leak into the symbol table"
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from zkvm_amd import native; lib = native.load_library(); "
            "names = %r; print(sum(1 for n in names if lib.zkgpu_hook(n.encode()))); print(lib.zkgpu_hook(b'zkgpu_init') or 0)\n"
            "try:\n    print(int(lib.zkgpu_debug_comm_mock(None, 0, None, 0)))\nexcept native.ZkGpuError as e:\n    print('raised', e.code)" % (ROOT, hooks))
    out = {}
    for on in ("", "1"):
        env = dict(os.environ)
        env.pop("ZKGPU_TEST_HOOKS", None)
        if on:
            env["ZKGPU_TEST_HOOKS"] = on
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out[on] = r.stdout.strip().splitlines()[-3:]
    assert out[""] == ["0", "0", "raised -1"]                     # nothing answers; the binding raises ZKGPU_EINVAL
    assert out["1"] == [str(len(hooks)), "0", "0"]                # every hook answers (an export is not a hook); the mock switch works


def test_the_library_never_edits_the_environment_it_only_recommends(lib):
    """no setenv / putenv anywhere under zkvm_amd/csrc; zkgpu_runtime_hint writes its recommendation into
    the caller's buffer, answers APPLY (unset, runtime not started) or PRESENT (the caller set it) without a GPU, and
    leaves the environment exactly as it found it -- exporting is the host's act (zkvm_amd.runtime_hint, os.environ)."""
    import ctypes
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "zkvm_amd", "csrc")
    for f in os.listdir(csrc):
        text = re.sub(r"//[^\n]*|/\*.*?\*/", "", open(os.path.join(csrc, f), errors="ignore").read(), flags=re.S)
        assert not re.search(r"\b(setenv|putenv|unsetenv)\s*\(", text), f
    child = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ.pop("GPU_MAX_HW_QUEUES", None)
from zkvm_amd import native
lib = native.load_library()
buf = ctypes.create_string_buffer(64)
before = dict(os.environ)
rc = lib.zkgpu_runtime_hint(buf, 64)
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
assert rc == 0 and buf.value == b"GPU_MAX_HW_QUEUES=18" and libc.getenv(b"GPU_MAX_HW_QUEUES") is None, (rc, buf.value)
small = ctypes.create_string_buffer(8)
assert lib.zkgpu_runtime_hint(small, 8) == 0 and small.value == b"GPU_MAX"          # truncated, terminated
assert lib.zkgpu_runtime_hint(None, 0) == 0
rc2, text = native.runtime_hint()                                                     # the HOST applies it
assert rc2 == 0 and os.environ["GPU_MAX_HW_QUEUES"] == "18" and libc.getenv(b"GPU_MAX_HW_QUEUES") == b"18"
assert lib.zkgpu_runtime_hint(buf, 64) == 1                                          # present now
os.environ["GPU_MAX_HW_QUEUES"] = "12"
assert native.runtime_hint() == (1, "GPU_MAX_HW_QUEUES=18") and os.environ["GPU_MAX_HW_QUEUES"] == "12"   # the caller's choice stays
print("HINT OK")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=120)
    assert "HINT OK" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])
