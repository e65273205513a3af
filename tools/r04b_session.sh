#!/bin/bash
# round-4 second half: quick A/B session.  r04b_session.sh <tag> [stages]   stages: tests,ab,tl,ablate
set -u
tag=${1:-r04b}
stages=${2:-tests,ab,tl,ablate}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
has() { [[ ",$stages," == *",$1,"* ]]; }
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], d.get("single_shot_ms"), {k:v["avg_launch_ms"] for k,v in r["per_kernel"].items()}, r.get("stages_ms_scouting_steps"), "fallbacks", d["config"].get("merge_fallbacks"))
    except Exception as e: print(f, "ERR", e)
PY
}
if has tests; then
  timeout 1500 python -m pytest ${TESTS:-tests/test_gpu_device_merge.py tests/test_gpu_parity.py tests/test_gpu_device_view.py} -m gpu -q -x --timeout=900 > $out/pytest.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $out/pytest.txt | tail -8
fi
B="--steps 20 --warmup 5 --cpu-sample 0 --e2e-reads 0 --single-shots 0"
if has ab; then
  for rep in 1 2; do
    timeout 300 python bench.py $B > $out/ab_new_$rep.json 2> $out/ab.err
    for sw in ${SWITCHES:-CRASS_DV_PAIR CRASS_DD_FULL_TABLE}; do
      env $sw=1 timeout 300 python bench.py $B > $out/ab_${sw}_$rep.json 2>> $out/ab.err
    done
  done
  line $out/ab_*.json
fi
if has tl; then
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp_tl -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/bench_tl.json 2> $out/tl.err
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp_tl 0 > $out/timeline_c2.txt 2>&1
  rm -rf $out/rp_tl
  cat $out/timeline_c2.txt
  cd $GRAFT_REPO_ROOT
fi
if has ablate; then
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/rp_ab -o r -- python3 $GRAFT_REPO_ROOT/tools/dm_ablate.py run > $out/ablate_run.txt 2> $out/ablate.err
  tail -3 $out/ablate_run.txt
  python3 $GRAFT_REPO_ROOT/tools/dm_ablate.py parse $out/rp_ab > $out/dm_ablate.txt 2>&1
  rm -rf $out/rp_ab
  cat $out/dm_ablate.txt
  cd $GRAFT_REPO_ROOT
fi
