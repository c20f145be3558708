cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in main pow4; do
  if [ $v = pow4 ]; then export ZKGPU_LIB=$R/build/ab/pow4/libzkgpu.so; else unset ZKGPU_LIB; fi
  for i in 1 2 3; do python3 $R/tools/msm_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['pairs_per_s']/1e6,1), d['ms'], d['equals_committed_expected_value'], {k:round(x,3) for k,x in d['kernel_ms'].items()})"; done
  rocprofv3 --kernel-trace -d $R/gpurun_out/msmtl_$v -o t --output-format csv -- python3 $R/tools/msm_bench.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/msmtl_$v -name "*kernel_trace.csv" | head -1); python3 $R/tools/msm_timeline.py $f > $R/gpurun_out/r06c_msm_timeline_$v.txt; cat $R/gpurun_out/r06c_msm_timeline_$v.txt; rm -rf $R/gpurun_out/msmtl_$v
done
unset ZKGPU_LIB
cd $R; timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
