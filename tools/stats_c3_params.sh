#!/bin/bash
# per-kernel totals (rocprofv3 --kernel-trace --stats) of configs[3] with other options:  tools/stats_c3_params.sh <tag> "<params>" [config]
tag=${1:-st_c3p}; params=${2:-d=20,D=40}; cfg=${3:-3}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --params "$params" --steps 5 --warmup 2 --cpu-sample 0 --e2e-reads 0 --single-shots 0 > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/rp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > $out/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %6s  total %10.3f ms  avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat $out/kernel_stats.txt
rm -rf $out/rp
