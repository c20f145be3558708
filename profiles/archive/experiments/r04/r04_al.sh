#!/bin/bash
# timing experiment: k_bucket_accumulate reading 64 of a row's 128 bytes and rebuilding the rest with two products (WRONG sums:
# the golden check is switched off by catching the assertion) -- what would 64-byte rows buy?
cd "$(dirname "$0")/../../.."
for g in 0 1; do
  ZKGPU_ACC_HALF=$g python3 - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from gpu_util import msm_2p20_inputs
from zkvm_amd import Context
ctx = Context(0)
n = 1 << 20
sc, uniform = msm_2p20_inputs(n)
pts = ctx.hash_to_points(uniform)
dev = torch.device("cuda", 0)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev); d_pt = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
for _ in range(3): ctx.msm_dev(d_sc, d_pt, n)
t0 = time.perf_counter()
for _ in range(5): ctx.msm_dev(d_sc, d_pt, n)
dt = (time.perf_counter() - t0) / 5
ctx.profile_reset(); ctx.profile(True)
for _ in range(5): ctx.msm_dev(d_sc, d_pt, n)
ctx.profile(False)
prof = ctx.profile_read()
print("half=%s: %.3f ms per call; k_bucket_accumulate %.3f ms" % (os.environ["ZKGPU_ACC_HALF"], dt * 1e3, prof["k_bucket_accumulate"][1] / prof["k_bucket_accumulate"][0]))
PY
done
