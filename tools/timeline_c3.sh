#!/bin/bash
# kernel timeline of one long-read step (BASELINE configs[3]): tools/timeline_c3.sh <tag>
tag=${1:-tl_c3}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --steps 4 --warmup 2 --cpu-sample 0 > $out/bench_prof.json 2> $out/prof.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 1 k_hint_positions k_recruit_finish > $out/timeline.txt
rm -rf $out/rp
cat $out/timeline.txt
