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
