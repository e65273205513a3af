#!/bin/bash
# Runs on the GPU box: the N-rank emulation (tools/scaling_projection.py) under rocprofv3 --kernel-trace --stats;
# leaves the kernel stats CSV under gpurun_out/$1/.   usage: tools/prof_sp.sh <tag> <world>
set -u
tag=${1:-prof_sp}; W=${2:-8}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/tools/scaling_projection.py 100000000 $W > $out/sp.txt 2> $out/prof.err
f=$(find $out/rp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/rp
head -40 $out/kernel_stats.csv | cut -c1-150
