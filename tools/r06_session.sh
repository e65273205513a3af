#!/bin/bash
# round-6 GPU evidence session (staged): r06_session.sh <tag> [stages]
# stages: smoke,tests,bench,params,prof,long,e2e,sweep,hygiene,other,scale,lens,pmc3,pmc4,pmc
# Every stage that gates the ones behind it is judged by its EXIT STATUS (a collection error, a timeout kill or a
# crashed interpreter is not "no ' failed' line"); rocprofv3 runs of bench.py carry --e2e-reads 0 (no child process inside a profile).
set -u
tag=${1:-r06}
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
        print(f.split("/")[-1], {k:d.get(k) for k in ("value","ms_per_step","first_call_ms","single_shot_ms","abi_fetch_ms","e2e_reads_per_s","scaling","n_gpus")}, r["kernel"], "frac", r["frac"], "path_frac", r["path_frac"], "traffic", r.get("traffic"), "valu", (r.get("valu_issue") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", (d.get("parity_checked") or {}).get("equal"), d["config"].get("launcher"), "fast_filter", d["config"].get("fast_filter"), "fallbacks", d["config"].get("merge_fallbacks"))
    except Exception as e: print(f, "ERR", e)
PY
}
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; rc=$?
  tail -1 $out/smoke.txt
  [ $rc -ne 0 ] && { tail -20 $out/smoke.txt; echo "SMOKE FAILED (exit $rc)"; exit 1; }
fi
if has tests; then
  timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -rf > $out/pytest.txt 2>&1; rc=$?
  grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $out/pytest.txt | tail -12
  [ $rc -ne 0 ] && { echo "TESTS FAILED (pytest exit $rc)"; exit 1; }
fi
if has bench; then
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c2_driver_style.json 2> $out/bench_c2_driver_style.err; rc=$?
  [ $rc -ne 0 ] && { tail -5 $out/bench_c2_driver_style.err; echo "DRIVER-STYLE BENCH FAILED (exit $rc)"; exit 1; }
  timeout 400 python bench.py --config 1 --e2e-reads 0 > $out/bench_c1.json 2> $out/bench_c1.err
  timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-reads 0 --alternate > $out/bench_c2_alternate.json 2> $out/bench_c2_alternate.err
  timeout 400 python bench.py --gpus 2 --local-copies --cpu-sample 0 --steps 20 > $out/bench_group2_shared.json 2> $out/bench_group2_shared.err
  summ $out/bench_c2_driver_style.json $out/bench_c1.json $out/bench_c2_alternate.json $out/bench_group2_shared.json
fi
if has params; then
  # what a user of non-default options gets: 10 M x 150 bp, the kernel that ran is in config.fast_filter
  for p in "w=7" "d=20,D=40" "s=20,S=60" "n=3"; do
    f=$out/bench_c1_params_$(echo $p | tr '=,' '__').json
    timeout 400 python bench.py --config 1 --params "$p" --steps 20 --warmup 3 --cpu-sample 200000 --e2e-reads 0 --single-shots 0 > $f 2> ${f%.json}.err
    summ $f
  done
  # (with its CPU leg: the line carries cpu_baseline and parity_checked — VERDICT r05 weak #11)
  timeout 600 python bench.py --config 3 --params "d=20,D=40" --steps 5 --warmup 2 --e2e-reads 0 --single-shots 0 > $out/bench_c3_params_d_20_D_40.json 2> $out/bench_c3_params.err
  summ $out/bench_c3_params_d_20_D_40.json
fi
if has prof; then
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_c2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_c2_prof.json 2> $out/c2_prof.err
  f=$(find $out/rp_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp_tl -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_tl.json 2> $out/tl.err
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp_tl 0 > $out/timeline_c2.txt 2>&1
  for c in 3 4; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_c$c -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_c${c}_prof.json 2> $out/c${c}_prof.err
    f=$(find $out/rp_c$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c${c}_kernel_stats.csv
  done
  rm -rf $out/rp_c2 $out/rp_tl $out/rp_c3 $out/rp_c4
  head -24 $out/c2_kernel_stats.csv | cut -c1-160
  head -14 $out/c3_kernel_stats.csv | cut -c1-160
  cat $out/timeline_c2.txt
  cd $GRAFT_REPO_ROOT
fi
if has long; then
  bash tools/tl_long.sh $tag/tl_c3 > /dev/null 2>&1; cp $out/tl_c3/timeline.txt $out/timeline_c3.txt; cat $out/timeline_c3.txt
  CRASS_NO_LIGHT=1 bash tools/tl_long.sh $tag/tl_c3_nolight > /dev/null 2>&1; cp $out/tl_c3_nolight/timeline.txt $out/timeline_c3_nolight.txt; tail -3 $out/timeline_c3_nolight.txt
  CRASS_HINT_PARTS=1 timeout 300 python tools/longread_phases.py 1000000 3000 > $out/longread_phases.txt 2>&1; cat $out/longread_phases.txt
  CRASS_NO_LIGHT=1 CRASS_HINT_PARTS=1 timeout 300 python tools/longread_phases.py 1000000 10 2>&1 | grep surv_prof > $out/longread_phases_nolight.txt; cat $out/longread_phases_nolight.txt
fi
if has e2e; then
  timeout 1200 python tools/e2e_big.py 50000000 auto whole stream > $out/e2e_big.txt 2>&1; grep -v "^{" $out/e2e_big.txt
  timeout 600 bash tools/e2e_cli.sh 5000000 > $out/e2e_cli.txt 2>&1; grep "wall\|searchAndRecruit\|fastx" $out/e2e_cli.txt | head -12
  timeout 300 python tools/consensus_timing.py 10000000 > $out/consensus_timing.json 2> $out/consensus_timing.err; tail -1 $out/consensus_timing.json
fi
if has sweep; then
  bash tools/sweep_seeds.sh $tag/sweeps ${SEEDS:-501 502 503} > $out/sweeps.txt 2>&1; cat $out/sweeps.txt
  bash tools/sweep_seeds_long.sh $tag/sweeps_long 150 ${LSEEDS:-511 512} > $out/sweeps_long.txt 2>&1; cat $out/sweeps_long.txt
fi
if has hygiene; then
  POISON=1 bash tools/mem_hygiene.sh $tag/hygiene > $out/hygiene.txt 2>&1; cat $out/hygiene.txt
fi
if has other; then
  timeout 400 python bench.py --config 3 --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
  timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --cpu-sample 200000 --e2e-reads 0 > $out/bench_c4.json 2> $out/bench_c4.err
  summ $out/bench_c3.json $out/bench_c4.json
fi
if has scale; then
  timeout 900 python tools/scaling_projection.py 100000000 2 4 8 > $out/scaling_projection.txt 2> $out/scaling_projection.err; tail -5 $out/scaling_projection.txt
  bash tools/tl_sp.sh $tag/tl8 8 > $out/timeline_sp8.txt 2>&1; tail -45 $out/timeline_sp8.txt
fi
if has lens; then
  # ms per 1.5 Gbases against the read length: lane-per-read filter (<= 256 bases), position hints as the filter (257 .. 2 048), long-read path
  bash tools/len_sweep.sh $tag/len_sweep > $out/len_sweep.txt 2>&1; cat $out/len_sweep.txt
  LENS="300 500 1000" CRASS_NO_HINT_FILTER=1 bash tools/len_sweep.sh $tag/len_sweep_general > $out/len_sweep_general_filter.txt 2>&1; cat $out/len_sweep_general_filter.txt
  # ... the same lengths under another window / seed lattice (the hint bits' every-position form), with and without it
  : > $out/len_sweep_params.txt
  for L in 300 1000; do n=$((1500000000 / L)); for p in w=7 d=20,D=40; do for e in CRASS_X=1 CRASS_NO_HINT_FILTER=1; do
    env $e python bench.py --read-len $L --total-reads $n --params $p --steps 5 --warmup 2 --cpu-sample 20000 --single-shots 0 --e2e-reads 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('L=$L $p $e', 'ms/step', d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['roofline']['per_kernel'].items()}, 'fast_filter', d['config'].get('fast_filter'), 'parity', (d.get('parity_checked') or {}).get('equal'))" >> $out/len_sweep_params.txt
  done; done; done
  cat $out/len_sweep_params.txt
fi
if has extra; then
  # round 6: what an idle device costs a step; a fresh context's first step against its fourth; the lane kernel's stages; host reads of pinned memory
  timeout 300 python tools/idle_effect.py > $out/idle_effect.txt 2>&1; tail -7 $out/idle_effect.txt
  bash tools/single_shot_tl.sh $tag/ss > /dev/null 2>&1; cp $out/ss/timeline.txt $out/single_shot_timeline.txt; cat $out/ss/steps_plain.txt >> $out/single_shot_timeline.txt; head -9 $out/ss/steps_plain.txt
  : > $out/survivor_stages.txt
  for n in 100000000 12500000; do for d in 1 2 3 4 0; do CRASS_SURV_DEBUG=$d python tools/survivor_ab.py $n 2>&1 | tail -1 >> $out/survivor_stages.txt; done; done; cat $out/survivor_stages.txt
  (cd profiles/ubench && hipcc -O2 pinned_read.cpp -o /tmp/pinned_read 2>/dev/null && /tmp/pinned_read) > $out/pinned_read.txt 2>&1; tail -4 $out/pinned_read.txt
  CRASS_HINT_PARTS=1 timeout 300 python tools/longread_phases.py 1000000 1000 > $out/longread_phases.txt 2>&1; grep "cat 2 total\|oracle" $out/longread_phases.txt | tail -3
  bash tools/r06_lens2.sh > $out/len_sweep_params_1000.txt 2>&1; tail -3 $out/len_sweep_params_1000.txt
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
