#!/bin/bash
# short GPU session: the tests that failed / hung in r03b, with per-test time-outs and the group trace
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03c}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_device_merge.py -m gpu -q -x --timeout=200 -k "out_of_memory or first_call_bound or n_kmers" > $out/t_dm.txt 2>&1; tail -5 $out/t_dm.txt
CRASS_GROUP_DEBUG=1 timeout 200 python -m pytest tests/test_gpu_group.py -m gpu -q -x --timeout=120 -k "exchange_overflow" > $out/t_overflow.txt 2>&1; tail -40 $out/t_overflow.txt
timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -q --timeout=200 -k "not exchange_overflow" > $out/t_group.txt 2>&1; tail -15 $out/t_group.txt
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q --timeout=300 > $out/t_multirank.txt 2>&1; tail -15 $out/t_multirank.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=200 -x -k "reference or uniform or ragged" > $out/t_parity.txt 2>&1; tail -5 $out/t_parity.txt
