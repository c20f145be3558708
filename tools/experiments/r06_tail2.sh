# round 6: re-check / locate tails folded per wavefront (k_recheck_fused on 1024 shares): lone-batch latency, value, the fused kernels' solo times
R=$GRAFT_REPO_ROOT; cd $R
python3 tools/lone_batch_gaps.py 1024 12 2>/dev/null | tail -19
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-msm --no-cpu --no-sweep 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['kernel_ms_solo']; print('value', d['value'], 'latency', d.get('latency_one_batch_ms'), 'steady', d.get('steady_state',{}).get('tx_per_s'), {k:round(v/1e6,2) for k,v in d.get('value_by_steps',{}).items() if k!='note'}, 'recheck', s.get('k_recheck_fused'), 'locate', s.get('k_locate_fused'))"; done
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py tests/test_zkvm_tx.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
