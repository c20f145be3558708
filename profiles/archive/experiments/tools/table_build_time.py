import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import time, sys
sys.path.insert(0, '.')
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens
ctx = Context(0)
for tb in (13, 14, 15):
    t0 = time.perf_counter()
    g = BulletproofGens(ctx, 256, table_bits=tb)
    print(tb, "build s", round(time.perf_counter() - t0, 3), flush=True)
    g.close()
