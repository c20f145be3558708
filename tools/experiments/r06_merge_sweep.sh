# round 6: the merge target again under the equal-parts policy: value (20 steps), value_by_steps, steady state per target
R=$GRAFT_REPO_ROOT; cd $R
for m in 6144 7168 8192 10240 12288; do
  for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --merge $m --no-msm --no-cpu --no-sweep 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d.get('value_by_steps',{}); print('merge', $m, 'value', round(d['value']/1e6,3), 'latency', d.get('latency_one_batch_ms'), 'steady', round(d.get('steady_state',{}).get('tx_per_s',0)/1e6,3), {k:round(v/1e6,2) for k,v in b.items() if k!='note'}, 'host', round(d.get('host_memory',{}).get('tickets',{}).get('tx_per_s',0)/1e6,3))"; done
done
