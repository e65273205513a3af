#!/bin/bash
# Runs on the GPU box: one rocprofv3 counter pass for the LDS side of pass 2's probe (k_anchor_filter_dev) — bank-conflict cycles,
# LDS-array cycles, LDS instructions — beside the VALU / wait counters, to tell "LDS array saturated" from "latency not covered".
# usage: pmc_lds.sh <tag>      (counters this gfx950 build of rocprofv3 does not know are dropped from the list)
set -u
tag=${1:-pmc_lds}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $out/avail.txt 2>&1
sel=""
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES; do
  grep -qw "$c" $out/avail.txt && sel="$sel $c"
done
echo "counters:$sel"
B="--config 2 --steps 3 --warmup 1 --cpu-sample 0 --single-shots 0 --e2e-reads 0"
# (few counters per pass: the SQ block has eight slots, derived metrics take several)
set -- $sel
while [ $# -gt 0 ]; do
  grp="$1 ${2:-} ${3:-} ${4:-}"; shift; shift 2>/dev/null; shift 2>/dev/null; shift 2>/dev/null
  n=$(echo $grp | tr ' ' '_')
  timeout 420 rocprofv3 --pmc $grp --output-format csv -d $out/rp_$n -o r -- python3 $GRAFT_REPO_ROOT/bench.py $B > $out/bench_$n.json 2> $out/$n.err
  f=$(find $out/rp_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_$n.csv; rm -rf $out/rp_$n
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/pmc_*.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "anchor_filter" in k or "filter_fast" in k or "dm_verify" in k:
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]): print("   %-24s %16.0f per launch (%d launches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
