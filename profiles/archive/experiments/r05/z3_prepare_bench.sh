O=gpurun_out/r05z; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q > $O/tests2.log 2>&1; echo "tests rc=$?" >> $O/tests2.log
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench2_$i.json 2> $O/bench2_$i.err; done
tail -3 $O/tests2.log
python3 - <<'PY'
import json
for i in (1,2,3):
    d=json.loads(open("gpurun_out/r05z/bench2_%d.json"%i).read().strip().splitlines()[-1])
    print(d["value"], d["steady_state"]["tx_per_s"], d["latency_one_batch_ms"], d["roofline"]["valu"]["by_kernel"]["k_prepare"])
PY
