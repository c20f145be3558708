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


def test_collective_mock_hook_is_refused_unless_the_process_asked_for_test_hooks():
    """zkgpu_debug_comm_mock is in the shipped ABI; it must do nothing in a process that did not set
    ZKGPU_TEST_HOOKS=1 before loading the library (fresh interpreters: the answer is read once per process)."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from zkvm_amd import native; lib = native.load_library(); "
            "print(int(lib.zkgpu_debug_comm_mock(None, 0, None, 0)))" % ROOT)
    out = {}
    for hooks in ("", "1"):
        env = dict(os.environ)
        env.pop("ZKGPU_TEST_HOOKS", None)
        if hooks:
            env["ZKGPU_TEST_HOOKS"] = hooks
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out[hooks] = int(r.stdout.strip().splitlines()[-1])
    assert out[""] == -1          # ZKGPU_EINVAL: refused
    assert out["1"] == 0          # switched (to RCCL, which it already was): no all-gather served yet
