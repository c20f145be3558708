"""Timeline of the batches in flight (GPU box): ZKGPU_TIMELINE=<file> python tools/timeline_bench.py [inflight] [steps]
Every context is profiled (HIP events around each launch, one device-wide clock); prints, per batch of the middle of the
run, each kernel's start (relative to the batch's first kernel), duration and the gap since the kernel it depends on
in the same stream finished, plus the totals: sum of durations, sum of gaps, wall time of the batch."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "18")
path = os.environ.setdefault("ZKGPU_TIMELINE", "/tmp/zk_timeline.txt")
import torch
import bench
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Verifier
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = Context(0)
txs, expected = bench.workload_2x2(1024, 0)
gens = BulletproofGens(ctx, 256, table_bits=16)
r = bench.shake(b"verifier-r|0", 64 * 1024)
dev = torch.device("cuda", 0)
to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_com, d_pr, d_r = to_dev(b"".join(t[2] for t in txs)), to_dev(b"".join(t[3] for t in txs)), to_dev(r)
ctx.set_group_size(16)
ctxs = [ctx] + [ctx.fork() for _ in range(inflight - 1)]
v = Verifier(ctx, gens)
plen = len(txs[0][3])
def run(n):
    for i in range(n):
        c = ctxs[i % inflight]
        if i >= inflight:
            c.verify_wait()
        v.submit_packed_gpu_dev(2, 2, 1024, d_com, d_pr, plen, d_r, ctx=c)
    for i in range(max(n - inflight, 0), n):
        ctxs[i % inflight].verify_wait()
run(3 * inflight)
for c in ctxs:
    c.profile(True)
import time
t0 = time.perf_counter()
run(steps)
dt = time.perf_counter() - t0
for c in ctxs:
    c.profile(False)
print("inflight %d: %.4f ms/step" % (inflight, dt / steps * 1e3))
rows = []
for line in open(path):
    c, name, a, b = line.split()
    rows.append((float(a), float(b), c, name))
rows.sort()
by_ctx = collections.defaultdict(list)
for a, b, c, name in rows:
    by_ctx[c].append((a, b, name))
# split each context's launches into batches at k_batch_init
tot = collections.Counter(); cnt = collections.Counter()
shown = 0
for c, ev in by_ctx.items():
    batches, cur = [], []
    for a, b, name in ev:
        if name == "k_batch_init" and cur:
            batches.append(cur); cur = []
        cur.append((a, b, name))
    batches.append(cur)
    for bt in batches[2:-2]:
        t_first, t_last = bt[0][0], max(x[1] for x in bt)
        busy = sum(x[1] - x[0] for x in bt)
        tot["wall"] += t_last - t_first; tot["busy"] += busy; cnt["n"] += 1
        for a, b, name in bt:
            tot[name] += b - a; cnt[name] += 1
        if shown < 2:
            shown += 1
            print("--- batch on", c)
            for a, b, name in bt:
                print("   %8.3f  %7.3f  %s" % (a - t_first, b - a, name))
print("batches analysed:", cnt["n"], " mean wall %.3f ms, mean sum of kernel durations %.3f ms" % (tot["wall"] / cnt["n"], tot["busy"] / cnt["n"]))
for k in sorted(tot, key=lambda k: -tot[k]):
    if k not in ("wall", "busy"):
        print("   %-22s mean %.4f ms x %.1f per batch" % (k, tot[k] / cnt[k], cnt[k] / cnt["n"]))
