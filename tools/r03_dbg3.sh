#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1 CRASS_SDMA_DEBUG=1
CRASS_GUARD_PAGES=1 timeout 300 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x -s --timeout=200 -k "device_resident_exchange_two_shards_one_process" > $out/g1.txt 2>&1; echo "guard rc=$?"; grep -n "crass_sdma\|Fatal\|File \"/root\|Segmentation" $out/g1.txt | head -30
timeout 300 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x -s --timeout=200 -k "device_resident_exchange_two_shards_one_process" > $out/g2.txt 2>&1; echo "plain rc=$?"; grep -n "crass_sdma" $out/g2.txt | head -5
