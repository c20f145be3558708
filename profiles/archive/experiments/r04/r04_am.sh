#!/bin/bash
# k_merge_inputs on aligned words + a bounded grid: ticket tests, then the driver's run three times, and a timeline of its region
cd "$(dirname "$0")/../../.."
python3 -m pytest tests/test_gpu_block.py -m gpu -x -q -k "ticket or merged or benched" 2>&1 | tail -2
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('rep $rep: value %.0f steady %.0f latency %s host %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s']))"
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_am -- python3 bench.py --lean --steps 20 --warmup 5 > /dev/null 2> gpurun_out/tl_am.err
python3 tools/trace_region.py $(ls gpurun_out/tl_am/*/*_kernel_trace.csv | head -1) > gpurun_out/r04am_timeline.txt 2>&1; rm -rf gpurun_out/tl_am; head -12 gpurun_out/r04am_timeline.txt; tail -2 gpurun_out/r04am_timeline.txt
