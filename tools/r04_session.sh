#!/bin/bash
# round-4 GPU session (staged): r04_session.sh <tag> [stages]
# stages: smoke,tests,bench,prof,e2e,sweep,hygiene,other,scale,pmc3,pmc4,pmc
set -u
tag=${1:-r04}
stages=${2:-smoke,tests,bench,prof}
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
        print(f.split("/")[-1], {k:d.get(k) for k in ("value","ms_per_step","first_call_ms","single_shot_ms","abi_fetch_ms","e2e_reads_per_s","scaling","n_gpus")}, r["kernel"], "frac", r["frac"], "path_frac", r["path_frac"], "traffic", r.get("traffic"), "valu", (r.get("valu_issue") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["config"].get("launcher"), "fallbacks", d["config"].get("merge_fallbacks"))
    except Exception as e: print(f, "ERR", e)
PY
}
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { tail -20 $out/smoke.txt; echo SMOKE FAILED; exit 1; }
  tail -1 $out/smoke.txt
fi
if has tests; then
  timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -rf > $out/pytest.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $out/pytest.txt | tail -12
  grep -q " failed" $out/pytest.txt && { echo TESTS FAILED; exit 1; }
fi
if has bench; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c2_driver_style.json 2> $out/bench_c2_driver_style.err || tail -5 $out/bench_c2_driver_style.err
  timeout 400 python bench.py --config 1 --cpu-sample 0 --e2e-reads 0 > $out/bench_c1.json 2> $out/bench_c1.err
  timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-reads 0 --alternate > $out/bench_c2_alternate.json 2> $out/bench_c2_alternate.err
  timeout 400 python bench.py --gpus 2 --local-copies --cpu-sample 0 --steps 20 > $out/bench_group2_shared.json 2> $out/bench_group2_shared.err
  CRASS_DEVICE_VIEW=1 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-reads 0 --single-shots 0 > $out/bench_c2_device_view.json 2> $out/bench_c2_device_view.err
  summ $out/bench_c2_driver_style.json $out/bench_c1.json $out/bench_c2_alternate.json $out/bench_group2_shared.json $out/bench_c2_device_view.json
fi
if has prof; then
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_c2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_c2_prof.json 2> $out/c2_prof.err
  f=$(find $out/rp_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp_tl -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_tl.json 2> $out/tl.err
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp_tl 0 > $out/timeline_c2.txt 2>&1
  rm -rf $out/rp_c2 $out/rp_tl
  head -40 $out/c2_kernel_stats.csv | cut -c1-160
  cat $out/timeline_c2.txt
  cd $GRAFT_REPO_ROOT
fi
if has e2e; then
  timeout 600 bash tools/e2e_cli.sh 5000000 > $out/e2e_cli.txt 2>&1; grep "wall\|searchAndRecruit" $out/e2e_cli.txt | head -12
  CRASS_INGEST=stream timeout 600 bash tools/e2e_cli.sh 5000000 > $out/e2e_cli_streamed.txt 2>&1; grep "wall\|searchAndRecruit" $out/e2e_cli_streamed.txt | head -12
  timeout 300 python tools/consensus_timing.py 10000000 > $out/consensus_timing.json 2> $out/consensus_timing.err; tail -1 $out/consensus_timing.json
fi
if has sweep; then
  bash tools/sweep_seeds.sh $tag/sweeps ${SEEDS:-401 402 403} > $out/sweeps.txt 2>&1; cat $out/sweeps.txt
fi
if has hygiene; then
  POISON=1 bash tools/mem_hygiene.sh $tag/hygiene > $out/hygiene.txt 2>&1; cat $out/hygiene.txt
fi
if has other; then
  timeout 400 python bench.py --config 3 --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_c3.json 2> $out/bench_c3.err
  timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --cpu-sample 0 --e2e-reads 0 > $out/bench_c4.json 2> $out/bench_c4.err
  summ $out/bench_c3.json $out/bench_c4.json
  for d in 0 1 2 3; do CRASS_SURV_DEBUG=$d python tools/longread_ab.py 2>&1 | tail -1; done > $out/longread_breakdown.txt; cat $out/longread_breakdown.txt
fi
if has scale; then
  timeout 600 python tools/scaling_projection.py 100000000 2 4 8 > $out/scaling_projection.txt 2> $out/scaling_projection.err; tail -4 $out/scaling_projection.txt
  CRASS_MERGE_PROFILE=1 timeout 600 python tools/scaling_projection.py 100000000 8 2>&1 | grep "helper\|recruit:" | tail -6 > $out/sp8_merge_profile.txt; cat $out/sp8_merge_profile.txt
  bash tools/tl_sp.sh $tag/tl8 8 > $out/timeline_sp8.txt 2>&1; tail -45 $out/timeline_sp8.txt
  python tools/group_first_call.py 100000000 2 > $out/group_first_call.txt 2>&1; tail -8 $out/group_first_call.txt
fi
if has pmc3; then
  bash tools/pmc_round.sh $tag/pmc_c3 3 1000000 10000 > $out/pmc_c3_summary.txt 2>&1; tail -30 $out/pmc_c3_summary.txt
fi
if has pmc4; then
  bash tools/pmc_round.sh $tag/pmc_c4 4 200000000 150 > $out/pmc_c4_summary.txt 2>&1; tail -30 $out/pmc_c4_summary.txt
fi
if has pmc; then
  bash tools/pmc_round.sh $tag/pmc 2 100000000 150 > $out/pmc_summary.txt 2>&1; tail -40 $out/pmc_summary.txt
fi
