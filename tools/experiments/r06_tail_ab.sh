# round 6: the Horner chain with the next window fetched ahead (main) -- lone-batch latency, value; the bucket reduction in chunks of 32 (variant)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do python3 tools/msm_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('main', round(d['pairs_per_s']/1e6,1), d['ms'], {k:round(x,3) for k,x in d['kernel_ms'].items() if 'reduce' in k or 'partials' in k})"; done
export ZKGPU_LIB=$R/build/ab/chunk32/libzkgpu.so
for i in 1 2; do python3 tools/msm_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('chunk32', round(d['pairs_per_s']/1e6,1), d['ms'], d['equals_committed_expected_value'], {k:round(x,3) for k,x in d['kernel_ms'].items() if 'reduce' in k or 'partials' in k})"; done
unset ZKGPU_LIB
python3 tools/lone_batch_gaps.py 1024 12 2>/dev/null | tail -22
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-msm --no-cpu --no-sweep 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value', d['value'], 'latency', d.get('latency_one_batch_ms'), 'steady', d.get('steady_state',{}).get('tx_per_s'), d.get('value_by_steps'))"; done
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
