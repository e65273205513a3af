#!/bin/bash
# what `crass-hip` spends outside main (spawn -> main, _exit -> reaped) on a 50 M-read FASTA, with the tear-down left to the
# process's end (default) and handed to a thread beside the output stage (CRASS_TEARDOWN=async)
mkdir -p gpurun_out
N=${1:-50000000}
python3 tools/e2e_big.py $N auto > gpurun_out/e2e_outside_default.txt 2>&1
CRASS_TEARDOWN=async python3 tools/e2e_big.py $N auto > gpurun_out/e2e_outside_async.txt 2>&1
grep -h "wall\|spawn\|_exit\|cli:" gpurun_out/e2e_outside_default.txt gpurun_out/e2e_outside_async.txt
