"""2^20-term MSM microbench (BASELINE configs[2]) with per-kernel HIP-event times, checked against the committed
expected value (tests/golden/msm_2p20.json).  GPU box only; what tools/profile_bench.sh profiles for the MSM pipeline."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from zkvm_amd import Context
ctx = Context(0)
bench.emit(bench.msm_microbench(ctx, torch, torch.device("cuda", 0)))
ctx.close()
