#!/bin/bash
# kernel timeline of one bench step (rocprofv3 kernel trace): tools/timeline_only.sh <tag> [extra bench args]
tag=${1:-tl}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 "$@" > $out/bench_prof.json 2> $out/prof.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 0 > $out/timeline.txt
rm -rf $out/rp
cat $out/timeline.txt
