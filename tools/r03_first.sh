#!/bin/bash
# round-3 first GPU session: the whole GPU suite on the first-call-bounds build, the headline bench (configs[2], 100 M reads)
# with single-shot figures, A/B of the first call without the load-time pool sizing, kernel stats + PMC passes at 100 M.
set -u
tag=${1:-r03a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.err
CRASS_NO_PRESIZE=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $out/bench_nopresize.json 2> $out/bench_nopresize.err
timeout 600 python bench.py --config 1 --cpu-sample 0 > $out/bench_c1.json 2> $out/bench_c1.err
CRASS_NO_PRESIZE=1 timeout 600 python bench.py --config 1 --cpu-sample 0 > $out/bench_c1_nopresize.json 2> $out/bench_c1_nopresize.err
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_c2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --single-shots 0 > $out/bench_c2_prof.json 2> $out/c2_prof.err
f=$(find $out/rp_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
rm -rf $out/rp_c2
cd $GRAFT_REPO_ROOT
bash tools/pmc_round.sh $tag/pmc 2 100000000 150 > $out/pmc_summary.txt 2>&1
for f in $out/bench_default.json $out/bench_nopresize.json $out/bench_c1.json $out/bench_c1_nopresize.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print({k:d.get(k) for k in ("value","ms_per_step","first_call_ms","single_shot_ms","scaling","n_gpus")}, d.get("single_shot",{}).get("ms_all"), r["kernel"], "frac", r["frac"], "path_frac", r["path_frac"], d["config"]["workload"][:40], d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("ERR", e)
PY
done
head -30 $out/c2_kernel_stats.csv
