#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
timeout 600 python3 tools/lone_batch_gaps.py 1024 12 > $O/lone_batch_gaps_1024.txt 2> $O/gaps.err; echo "rc=$?"
timeout 600 python3 tools/lone_batch_gaps.py 10240 8 > $O/lone_batch_gaps_10240.txt 2>> $O/gaps.err; echo "rc=$?"
head -4 $O/lone_batch_gaps_1024.txt; head -3 $O/lone_batch_gaps_10240.txt; tail -5 $O/gaps.err
