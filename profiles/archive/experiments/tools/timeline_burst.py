"""Timeline of a burst: two device batches of 10240 transactions submitted together on two contexts (what the bench's
20-step run is).  ZKGPU_TIMELINE is set here; prints every launch of both batches: start, end (ms since the first)."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "18")
path = os.environ.setdefault("ZKGPU_TIMELINE", "/tmp/zk_timeline_burst.txt")
import time
import torch
import bench
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Verifier
reps = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,10").split(",")]   # batches of 1024 per context
nctx = len(reps)
rep = max(reps)
ctx = Context(0)
txs, expected = bench.workload_2x2(1024, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
gens = BulletproofGens(ctx, 256, table_bits=16)
r = bench.shake(b"verifier-r|0", 64 * 1024)
dev = torch.device("cuda", 0)
to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_com, d_pr, d_r = (to_dev(b"".join(t[2] for t in txs)).repeat(rep), to_dev(b"".join(t[3] for t in txs)).repeat(rep), to_dev(r).repeat(rep))
ctx.set_group_size(16)
ctxs = [ctx] + [ctx.fork() for _ in range(nctx - 1)]
v = Verifier(ctx, gens)
plen = len(txs[0][3])
sub_ms = []
def burst():
    for c in ctxs:
        ts = time.perf_counter()
        v.submit_packed_gpu_dev(2, 2, 1024 * reps[ctxs.index(c)], d_com, d_pr, plen, d_r, ctx=c)
        sub_ms.append((time.perf_counter() - ts) * 1e3)
    for c in ctxs:
        c.verify_wait()
for _ in range(3):
    burst()
torch.cuda.synchronize()
t0 = time.perf_counter()
burst()
dt = time.perf_counter() - t0
print("burst of %s x 1024 tx: %.3f ms (%.0f tx/s); host time of the submits: %s ms" % (reps, dt * 1e3, 1024 * sum(reps) / dt, ["%.3f" % x for x in sub_ms[-nctx:]]))
for c in ctxs:
    c.profile(True)
if os.path.exists(path):
    os.remove(path)
burst()
for c in ctxs:
    c.profile(False)
rows = []
for line in open(path):
    c, name, a, b = line.split()
    rows.append((float(a), float(b), c, name))
rows.sort()
t_first = rows[0][0]
names = {}
for a, b, c, name in rows:
    names.setdefault(c, len(names))
    print("ctx%d %-22s %7.3f -> %7.3f  (%.3f)" % (names[c], name, a - t_first, b - t_first, b - a))
print("wall (profiled) %.3f ms" % (max(r_[1] for r_ in rows) - t_first))
