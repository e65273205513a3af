#!/bin/bash
# kernel timelines of one bench step under a list of environment settings: r03_tl.sh <tag> "VAR=1" "VAR2=1 VAR3=1" ...
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for setting in "" "$@"; do
  i=$((i+1))
  ( for kv in $setting; do export $kv; done
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp_$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_$i.json 2> $out/tl_$i.err )
  echo "== [$setting]" > $out/timeline_$i.txt
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp_$i 0 >> $out/timeline_$i.txt 2>&1
  rm -rf $out/rp_$i
  cat $out/timeline_$i.txt
done
