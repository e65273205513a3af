#!/bin/bash
# round-2 final evidence run (GPU box): whole GPU suite, parity sweep, every bench mode, kernel-trace stats of the default
# bench and of configs 2-4, a one-step timeline, the PMC passes, the consensus stage.  Outputs under gpurun_out/<tag>/.
set -u
tag=${1:-r02final}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
timeout 600 python tools/parity_sweep.py 400 ${SEED:-41} > $out/sweep.txt 2>&1; tail -1 $out/sweep.txt
cd /tmp; export TMPDIR=/tmp
stats() {   # stats <name> <bench args...>: rocprofv3 --kernel-trace --stats of one bench command -> <name>_kernel_stats.csv
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_$name -o r -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench_${name}_prof.json 2> $out/${name}_prof.err
  f=$(find $out/rp_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
  rm -rf $out/rp_$name
}
stats default --cpu-sample 0
stats c2 --config 2 --gpus 1 --cpu-sample 0 --steps 5 --warmup 2
stats c3 --config 3 --cpu-sample 0 --steps 5 --warmup 2
stats c4 --config 4 --cpu-sample 0 --steps 5 --warmup 2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $out/bench_tl.json 2> $out/tl.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 0 > $out/timeline.txt; rm -rf $out/rp
cd $GRAFT_REPO_ROOT
bash tools/pmc_round.sh $tag/pmc > $out/pmc_summary.txt 2>&1
cp $out/pmc/pmc_traffic.json profiles/r02_pmc_traffic.json 2>/dev/null      # (on the box only: so that the bench lines below quote it)
timeout 300 python bench.py --cpu-sample 0 --steps 100 > /dev/null 2>&1      # (throw-away: the first run after the counter passes shows many slow steps, step_ms.p90 0.80 instead of 0.71)
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --alternate --cpu-sample 0 > $out/bench_alt.json 2> $out/bench_alt.err
timeout 600 python bench.py --gpus 1 --config 2 --cpu-sample 0 > $out/bench_c2.json 2> $out/bench_c2.err
timeout 600 python bench.py --config 3 --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_c3.json 2> $out/bench_c3.err
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_c4.json 2> $out/bench_c4.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --total-reads 20000000 --cpu-sample 0 > $out/bench_2rank_shared.json 2> $out/bench_2rank_shared.err
timeout 600 python tools/consensus_timing.py 10000000 > $out/consensus_timing.json 2> $out/consensus_timing.err
for f in $out/bench_default.json $out/bench_alt.json $out/bench_c2.json $out/bench_c3.json $out/bench_c4.json $out/bench_2rank_shared.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print({k:d[k] for k in ("value","ms_per_step","first_call_ms","scaling","n_gpus")}, r["kernel"], "frac", r["frac"], "path_frac", r["path_frac"], "traffic", r["traffic"], "valu", (r.get("valu_issue") or {}).get("frac"), d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("ERR", e)
PY
done
cat $out/timeline.txt; tail -3 $out/consensus_timing.json
