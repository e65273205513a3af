#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
env | grep -E "^(HSA|HIP|ROC|GPU_|AMD|NCCL|RCCL)" | sort > $out/env.txt; cat $out/env.txt
