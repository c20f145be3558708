"""2^20-term MSM microbench (BASELINE configs[2]) with per-kernel HIP-event times.  GPU box only."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from zkvm_amd import Context
ctx = Context(0)
r = bench.msm_microbench(ctx, torch, torch.device("cuda", 0))
print(json.dumps(r))
