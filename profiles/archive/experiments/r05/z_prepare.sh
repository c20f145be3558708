#!/bin/bash
# round 5: k_prepare's power tables as outer products of small tables -- parity, per-section clocks, solo times, the driver's line
O=gpurun_out/r05z; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
ZKGPU_LIB=build/ab/stamps/libzkgpu.so timeout 300 python3 tools/prep_stamps.py 1 > $O/stamps_serial_new.txt 2>&1
timeout 600 python3 bench.py --solo --steps 20 > $O/solo.json 2> $O/solo.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.log; cat $O/stamps_serial_new.txt
python3 - <<'PY'
import json
for f in ("gpurun_out/r05z/solo.json","gpurun_out/r05z/bench.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"))
        r=d.get("roofline",{})
        bk=(r.get("valu") or {}).get("by_kernel") or {}
        for k,v in bk.items(): print("  ",k,v)
        for k in ("steady_state","latency_one_batch_ms"): print("  ",k,d.get(k) or (d.get("config") or {}).get(k))
    except Exception as e: print(f,"ERR",e)
PY
