#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q > $O/tests3.log 2>&1; echo "tests rc=$?" >> $O/tests3.log
ZKGPU_LIB=build/ab/stamps/libzkgpu.so python3 tools/prep_stamps.py 1 > $O/stamps_serial_rot.txt 2>&1
timeout 600 python3 bench.py --solo --steps 20 > $O/solo_rot.json 2> $O/solo_rot.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_rot.json 2> $O/bench_rot.err
tail -2 $O/tests3.log; cat $O/stamps_serial_rot.txt
python3 -c "
import json
d=json.loads(open('$O/solo_rot.json').read().strip().splitlines()[-1]);print({k:round(v,4) for k,v in d['solo_kernel_ms'].items()})
d=json.loads(open('$O/bench_rot.json').read().strip().splitlines()[-1]);print(d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'])"
