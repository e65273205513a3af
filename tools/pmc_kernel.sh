#!/bin/bash
# SQ counters of the kernels matching a pattern: tools/pmc_kernel.sh <tag> <kernel regex> <bench args...>
tag=$1; pat=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.json 2> $out/err.txt
f=$(find $out/rp -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc.csv; rm -rf $out/rp
python3 - $out/pmc.csv "$pat" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("crass::", "")
    if re.search(sys.argv[2], k): acc[k + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
