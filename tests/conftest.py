import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the collective mock of the exchange tests (zkgpu_debug_comm_mock) is refused by the library unless the process asked for
# the test hooks before loading it: a deployed verifier never does
os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the library never edits the environment; the host exports what zkgpu_runtime_hint recommends before its first HIP call
# (GPU_MAX_HW_QUEUES).  Here, before any test has touched the GPU.  (Not built yet: the tests that need it say so.)
try:
    from zkvm_amd import runtime_hint
    runtime_hint()
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "ristretto255.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def pyref():
    from oracle import pyref as R
    return R
