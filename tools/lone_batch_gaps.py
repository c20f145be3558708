"""What a HIP graph could win on a LONE batch: one 1024-transaction batch at a time through the ticket
path (zkgpu_verifier_submit_dev + _wait, nothing else in flight), the library's per-launch events on a device-wide clock
(ZKGPU_TIMELINE).  Per batch: the span from the first launch's start to the last launch's end, the time inside it during which
NO kernel of the batch runs (launch gaps, event waits between streams: all a graph could remove), and the host's time in the
submit call.  usage: lone_batch_gaps.py [batch=1024] [reps=12]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "18")
path = os.environ.setdefault("ZKGPU_TIMELINE", "/tmp/zk_lone_batch_timeline.txt")
from gpu_util import benched_randomness, benched_step
from zkvm_amd import Context
from zkvm_amd.verifier import BlockVerifier, BulletproofGens
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=-1)
bv = BlockVerifier(ctx, gens, batches_in_flight=5)
bv.set_merge(10240)
sets = []
for s in range(reps + 3):
    txs, expected = benched_step(1024, 0, 64, s)
    k = (batch + 1023) // 1024
    com, proofs = b"".join(t[2] for t in txs) * k, b"".join(t[3] for t in txs) * k
    r = benched_randomness(0, s, 1024) * k
    sets.append([ctx.to_device(x[: n * batch]) for x, n in ((com, 256), (proofs, len(txs[0][3])), (r, 64))] + [len(txs[0][3])])
for d in sets[:3]:                                           # warm: workspaces, plans
    bv.wait(bv.submit_dev(2, 2, batch, d[0], d[1], d[3], d[2]))
for i in range(bv.lanes()):
    ctx.lib.zkgpu_profile_enable(ctx.lib.zkgpu_verifier_lane(bv.h, i), 1)
wall, sub = [], []
marks = []
for d in sets[3:]:
    t0 = time.perf_counter()
    tk = bv.submit_dev(2, 2, batch, d[0], d[1], d[3], d[2])
    t1 = time.perf_counter()
    bv.wait(tk)
    t2 = time.perf_counter()
    wall.append((t2 - t0) * 1e3); sub.append((t1 - t0) * 1e3)
bv.close(); gens.close(); ctx.close()
rows = []
for line in open(path):
    c, name, a, b = line.split()
    rows.append((float(a), float(b), name))
rows.sort()
# a batch begins with its k_batch_init
groups, cur = [], []
for a, b, n in rows:
    if n == "k_batch_init" and cur:
        groups.append(cur); cur = []
    cur.append((a, b, n))
if cur:
    groups.append(cur)
groups = [g for g in groups if len(g) > 8][-reps:]
spans, idles = [], []
for g in groups:
    s0, s1 = g[0][0], max(x[1] for x in g)
    ev = sorted([(a, 1) for a, b, n in g] + [(b, -1) for a, b, n in g])
    depth, last, idle = 0, s0, 0.0
    for t, dlt in ev:
        if depth == 0:
            idle += t - last
        depth += dlt; last = t
    spans.append(s1 - s0); idles.append(idle)
print("lone batches of %d transactions, %d measured: wall (submit + wait) median %.3f ms; host time in submit median %.3f ms" % (batch, len(wall), statistics.median(wall), statistics.median(sub)))
print("device span first launch -> last launch end: median %.3f ms; inside it NO kernel of the batch running: median %.3f ms (min %.3f, max %.3f); launches per batch %d"
      % (statistics.median(spans), statistics.median(idles), min(idles), max(idles), statistics.median([len(g) for g in groups])))
g = groups[-1]
print("last batch, launches (ms since its first):")
for a, b, n in g:
    print("  %7.3f %7.3f  %s" % (a - g[0][0], b - g[0][0], n))
