#!/bin/bash
# round-3 GPU session (staged: a failing stage stops the session before the long ones run).
# usage: r03_session.sh <tag> [stages]   stages: smoke,ubench,tests,bench,prof,e2e,sweep,hygiene,other,scale,pmc3,pmc (default: the first row)
set -u
tag=${1:-r03b}
stages=${2:-smoke,ubench,tests,bench,prof,pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
has() { [[ ",$stages," == *",$1,"* ]]; }
summ() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f.split("/")[-1], {k:d.get(k) for k in ("value","ms_per_step","first_call_ms","single_shot_ms","scaling","n_gpus")}, (d.get("single_shot") or {}).get("ms_all"), r["kernel"], "frac", r["frac"], "path_frac", r["path_frac"], "traffic", r.get("traffic"), "valu", (r.get("valu_issue") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["config"].get("launcher"), "ovf?", d["config"].get("merge_fallbacks"))
    except Exception as e: print(f, "ERR", e)
PY
}
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { tail -20 $out/smoke.txt; echo SMOKE FAILED; exit 1; }
  tail -1 $out/smoke.txt
  timeout 300 python bench.py --config 1 --steps 20 --warmup 3 --cpu-sample 0 > $out/bench_c1_quick.json 2> $out/bench_c1_quick.err || { tail -5 $out/bench_c1_quick.err; echo QUICK BENCH FAILED; exit 1; }
  summ $out/bench_c1_quick.json
fi
if has ubench; then
  (cd profiles/ubench && mkdir -p bin && hipcc --offload-arch=gfx950 -O2 -o bin/vgpr_edge3 vgpr_edge3.hip && timeout 120 ./bin/vgpr_edge3) > $out/vgpr_edge3.txt 2>&1; cat $out/vgpr_edge3.txt
fi
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -rf > $out/pytest.txt 2>&1; tail -25 $out/pytest.txt
  grep -q " failed" $out/pytest.txt && { echo TESTS FAILED; exit 1; }
fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || tail -5 $out/bench_default.err
  CRASS_NO_PRESIZE=1 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $out/bench_nopresize.json 2> $out/bench_nopresize.err
  timeout 400 python bench.py --config 1 --cpu-sample 0 > $out/bench_c1.json 2> $out/bench_c1.err
  CRASS_NO_PRESIZE=1 timeout 400 python bench.py --config 1 --cpu-sample 0 > $out/bench_c1_nopresize.json 2> $out/bench_c1_nopresize.err
  timeout 400 python bench.py --gpus 2 --local-copies --cpu-sample 0 --steps 20 > $out/bench_group2_shared.json 2> $out/bench_group2_shared.err
  summ $out/bench_default.json $out/bench_nopresize.json $out/bench_c1.json $out/bench_c1_nopresize.json $out/bench_group2_shared.json
fi
cd /tmp; export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_c2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --single-shots 0 > $out/bench_c2_prof.json 2> $out/c2_prof.err
  f=$(find $out/rp_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp_tl -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 > $out/bench_tl.json 2> $out/tl.err
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp_tl 0 > $out/timeline_c2.txt 2>&1
  rm -rf $out/rp_c2 $out/rp_tl
  head -32 $out/c2_kernel_stats.csv
fi
if has e2e; then
  cd $GRAFT_REPO_ROOT
  timeout 600 bash tools/e2e_cli.sh 5000000 > $out/e2e_cli.txt 2>&1; tail -40 $out/e2e_cli.txt
  timeout 300 python tools/consensus_timing.py 10000000 > $out/consensus_timing.json 2> $out/consensus_timing.err; tail -3 $out/consensus_timing.json
fi
if has sweep; then
  cd $GRAFT_REPO_ROOT
  timeout 900 python tools/parity_sweep.py 300 ${SEED:-43} > $out/sweep.txt 2>&1; tail -2 $out/sweep.txt
fi
if has hygiene; then
  cd $GRAFT_REPO_ROOT
  POISON=1 bash tools/mem_hygiene.sh $tag/hygiene > $out/hygiene.txt 2>&1; cat $out/hygiene.txt
fi
if has other; then
  cd $GRAFT_REPO_ROOT
  timeout 400 python bench.py --config 3 --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_c3.json 2> $out/bench_c3.err
  timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_c4.json 2> $out/bench_c4.err
  CRASS_DV_ONE=1 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --single-shots 0 > $out/bench_c2_dv_one.json 2> $out/bench_c2_dv_one.err
  summ $out/bench_c3.json $out/bench_c4.json $out/bench_c2_dv_one.json
fi
if has scale; then
  cd $GRAFT_REPO_ROOT
  timeout 600 python tools/scaling_projection.py > $out/scaling_projection.txt 2> $out/scaling_projection.err; tail -12 $out/scaling_projection.txt
fi
if has pmc3; then
  cd $GRAFT_REPO_ROOT
  bash tools/pmc_round.sh $tag/pmc_c3 3 1000000 10000 > $out/pmc_c3_summary.txt 2>&1; tail -30 $out/pmc_c3_summary.txt
fi
if has pmc; then
  cd $GRAFT_REPO_ROOT
  bash tools/pmc_round.sh $tag/pmc 2 100000000 150 > $out/pmc_summary.txt 2>&1; tail -40 $out/pmc_summary.txt
fi
