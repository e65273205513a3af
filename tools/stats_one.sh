#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench command: tools/stats_one.sh <tag> <bench args...>  -> gpurun_out/<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.json 2> $out/prof.err
f=$(find $out/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/rp
python3 - $out/kernel_stats.csv $out/bench.json <<'PY'
import csv, json, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("  %-46s calls %4s avg %9.1f us  %5s%%" % (r["Name"].replace("crass::","").replace("void ","").split("(")[0][:46], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["stages_ms_scouting_steps"])
PY
