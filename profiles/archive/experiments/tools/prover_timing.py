import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys, json, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
ht = bench.usable_cores(os.cpu_count())
os.environ["ZKGPU_PROVER_TIMING"] = "1"
print(json.dumps(bench.prover_microbench(ctx, gens, ht)), file=sys.stderr)
gens.close()
print(json.dumps(bench.prover_program_microbench(ctx, ht)), file=sys.stderr)
