#!/bin/bash
# kernel timeline of ONE step of rank 0 in the N-rank emulation (tools/scaling_projection.py):  tools/tl_sp.sh <tag> <world> ["VAR=1 ..."]
set -u
tag=$1; W=${2:-8}; setting=${3:-}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for kv in $setting; do export $kv; done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/tools/scaling_projection.py 100000000 $W > $out/sp.txt 2> $out/tl.err
echo "== world $W [$setting]" > $out/timeline.txt
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 1 >> $out/timeline.txt 2>&1
rm -rf $out/rp
cat $out/timeline.txt
