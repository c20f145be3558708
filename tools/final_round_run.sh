#!/bin/bash
# Run on the GPU box:  bash tools/final_round_run.sh <tag>   -- the artifacts of a round: GPU test log, smoke, the bench with the
# driver's flags and with the defaults, rocprofv3 stats / traces / PMC passes (tools/profile_bench.sh, tools/pmc_breakdown.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
timeout 600 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driverflags.json 2> gpurun_out/${TAG}_bench_driverflags.err; echo "driver command wall: $(( $(date +%s) - S )) s"
S=$(date +%s); ZKGPU_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n2_selflaunched_shared_gpu.json 2> gpurun_out/${TAG}_bench_n2_selflaunched_shared_gpu.err; echo "N=2 self-launched (both ranks on this GPU): rc $? wall $(( $(date +%s) - S )) s"
S=$(date +%s); python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default command wall: $(( $(date +%s) - S )) s"
S=$(date +%s); python bench.py --config 4 > gpurun_out/${TAG}_bench_config4.json 2> gpurun_out/${TAG}_bench_config4.err; echo "config 4: rc $? wall $(( $(date +%s) - S )) s; $(python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench_config4.json').readline()); print(d['value'], d['config'].get('workload'))")"
python - <<PY
import json
for name in ("driverflags", "default"):
    d=json.loads(open("gpurun_out/${TAG}_bench_%s.json" % name).readline())
    print(name, "value", d["value"], "steps", d["steps"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"), "hbm", d.get("hbm_copy",{}).get("measured_copy_GBps"))
    r=d["roofline"]; print("  roofline", r["kernel"], r["frac"], r["traffic"], "valu", r["step"].get("valu_issue_frac"), r["step"].get("valu_wave_instructions"))
    m=d.get("msm_2p20",{}); print("  msm", m.get("pairs_per_s"), m.get("ms"), m.get("roofline",{}).get("kernel"))
    t=d.get("tx_verify",{}); print("  tx", {k:v for k,v in t.items() if k!="note"})
    print("  host tickets", {k:v for k,v in d.get("host_memory",{}).items() if k!="note"})
    print("  cpu", d.get("cpu_baseline",{}).get("value"), "prover", d.get("prover",{}).get("proofs_per_s"), d.get("prover_1024_constraints",{}).get("proofs_per_s"), "sweep", {k:v.get("tx_per_s") for k,v in d["setup"].get("table_bits_sweep",{}).items() if isinstance(v,dict)})
PY
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
bash tools/pmc_breakdown.sh $TAG > gpurun_out/${TAG}_pmcx.log 2>&1
ls gpurun_out | grep $TAG | wc -l
