#!/bin/bash
# long-read parity sweeps (tools/parity_sweep_long.py): plain seeds, then one sweep each with the once-per-process switches and
# under guard pages + poison: sweep_seeds_long.sh <tag> <cases> <seed> ...
set -u
tag=$1; n=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT
last=0
for seed in "$@"; do
  timeout 1500 python tools/parity_sweep_long.py $n $seed > $out/long_$seed.txt 2>$out/long_$seed.err; echo "seed $seed rc=$? $(tail -1 $out/long_$seed.txt)"
  last=$seed
done
CRASS_SINK_EAGER=1 timeout 1500 python tools/parity_sweep_long.py $n $((last + 1)) > $out/long_eager.txt 2>$out/long_eager.err; echo "CRASS_SINK_EAGER seed $((last + 1)) rc=$? $(tail -1 $out/long_eager.txt)"
CRASS_HL_HOST_DEDUPE=1 timeout 1500 python tools/parity_sweep_long.py $n $((last + 2)) > $out/long_hostdedupe.txt 2>$out/long_hostdedupe.err; echo "CRASS_HL_HOST_DEDUPE seed $((last + 2)) rc=$? $(tail -1 $out/long_hostdedupe.txt)"
CRASS_GUARD_PAGES=1 CRASS_POISON=1 timeout 1500 python tools/parity_sweep_long.py $n $((last + 3)) > $out/long_guard.txt 2>$out/long_guard.err; echo "guard+poison seed $((last + 3)) rc=$? $(tail -1 $out/long_guard.txt)"
grep -h "^FAIL" $out/long_*.txt | head -20
