#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03e}
mkdir -p $out
cd $GRAFT_REPO_ROOT
CRASS_GROUP_DEBUG=1 timeout 90 python -m pytest tests/test_gpu_group.py -m gpu -q -x -s -k "exchange_overflow" > $out/t_overflow.txt 2>&1; tail -40 $out/t_overflow.txt
timeout 300 python -m pytest tests/test_gpu_device_merge.py -m gpu -q -x --timeout=200 -k "out_of_memory" > $out/t_dm.txt 2>&1; tail -5 $out/t_dm.txt
