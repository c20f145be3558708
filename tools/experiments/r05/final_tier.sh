O=gpurun_out/r05fin; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/gpu_tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
bash tools/profile_bench.sh r05 > $O/profile.log 2>&1
tail -3 $O/gpu_tests.log; cat $O/smoke.log | tail -1; tail -c 600 $O/bench_driverflags.json
