#!/bin/bash
# how many of a wavefront's 64 lanes work, per kernel: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for K in "solo bench.py --solo --steps 6" "prover tools/prover_profile.py" "proverprog tools/prover_program_profile.py"; do
  N=$(echo $K | cut -d" " -f1); S=$(echo $K | cut -d" " -f2-)
  D=$R/gpurun_out/r05z/lanes_$N
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $D -- python3 $R/$S > /dev/null 2> $D.err
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for n in ("solo","prover","proverprog"):
    print("==",n)
    for f in glob.glob("gpurun_out/r05z/lanes_%s/*/*counter_collection.csv"%n):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("zk::","").replace("void ","")
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        rows=[]
        for k,c in acc.items():
            if c.get("SQ_ACTIVE_INST_VALU",0)>0 and c.get("SQ_INSTS_VALU",0)>1e6:
                rows.append((c["SQ_INSTS_VALU"],k,c["SQ_THREAD_CYCLES_VALU"]/(c["SQ_ACTIVE_INST_VALU"]*64)))
        for v,k,u in sorted(rows,reverse=True)[:16]:
            print("  %-34s insts %9.1f M  lanes busy %.2f" % (k[:34], v/1e6, u))
PY
