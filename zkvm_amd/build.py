"""Build lib/libzkgpu.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "zkgpu.hip")
OUT = os.path.join(HERE, "lib", "libzkgpu.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))
        if f.endswith((".hip", ".hpp", ".cpp", ".inc", ".h", ".map"))]   # every source: a stale library is a silent wrong answer
HOST_SRC = os.path.join(HERE, "csrc", "hostlib.cpp")
HOST_OUT = os.path.join(HERE, "lib", "libzkhost.so")
DEPS.append(os.path.join(HERE, "..", "include", "zkgpu.h"))
DEPS.append(os.path.join(HERE, "..", "include", "zkgpu_hooks.h"))


def stale() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(HOST_OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not (force or stale()):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fvisibility=hidden: the library exports what include/zkgpu.h declares and nothing else (no C++ internals, no hooks)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-Wl,--version-script=" + os.path.join(HERE, "csrc", "zkgpu.map"), "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    # host-only logic (scalars, Merlin, R1CS verifier preparation) for the CPU test tier
    cxx = os.environ.get("CXX", "g++")
    subprocess.run([cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-pthread", "-o", HOST_OUT, HOST_SRC], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
