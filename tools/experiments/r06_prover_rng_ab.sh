# round 6: the TranscriptRng's draws one lane per proof (ZKGPU_PV_RNG=lanes) against one wavefront per proof (coop), fresh process per setting
R=$GRAFT_REPO_ROOT; cd $R
for mode in coop lanes coop lanes; do
  for spec in "cloak 16384" "program 8192" "cloak 2048" "program 1024"; do
    ZKGPU_PV_RNG=$mode python3 tools/prover_sweep.py child $spec 16 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$mode', d['kind'], d['batch'], d['proofs_per_s'], d['ms_all'], d['kernel_ms'])"
  done
done
ZKGPU_PV_RNG=lanes timeout 900 python -m pytest tests/test_gpu_verifier.py tests/test_gpu_block.py -m gpu -x -q -k "prov" 2>&1 | grep -E "passed|failed|error" | tail -3
