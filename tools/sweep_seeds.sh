#!/bin/bash
# several parity sweeps with different seeds (tools/parity_sweep.py), the last one under guard pages: sweep_seeds.sh <tag> <seed> ...
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT
last=""
for seed in "$@"; do
  timeout 900 python tools/parity_sweep.py 300 $seed > $out/sweep_$seed.txt 2>&1; echo "seed $seed rc=$? $(tail -1 $out/sweep_$seed.txt)"
  last=$seed
done
CRASS_GUARD_PAGES=1 CRASS_POISON=1 timeout 900 python tools/parity_sweep.py 300 $((last + 1)) > $out/sweep_guard_$((last + 1)).txt 2>&1; echo "guard+poison seed $((last + 1)) rc=$? $(tail -1 $out/sweep_guard_$((last + 1)).txt)"
