#!/bin/bash
# kernel timeline of one long-read step (BASELINE configs[3]) with the environment given as VAR=VALUE arguments after the tag:
#   tools/tl_long.sh <tag> [VAR=VALUE ...]
tag=${1:-tl_c3}; shift
for kv in "$@"; do export "$kv"; done
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --steps 4 --warmup 2 --cpu-sample 0 --e2e-reads 0 > $out/bench_prof.json 2> $out/prof.err
python3 $GRAFT_REPO_ROOT/tools/timeline_long.py $out/rp ${HINT_LAUNCHES:-2} > $out/timeline.txt
rm -rf $out/rp
grep -v "fillBuffer\|k_dm_\|k_dx_\|mask_\|block_scan\|COPY\|copyBuffer" $out/timeline.txt
