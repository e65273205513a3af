#!/bin/bash
# round-2 iteration: GPU tests (-x), short parity sweep, bench, timeline of one step
set -u
tag=${1:-it}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -8 $out/pytest.txt
timeout 900 python tools/parity_sweep.py ${SWEEP:-120} ${SEED:-7} > $out/sweep.txt 2>&1; tail -3 $out/sweep.txt
timeout 400 python bench.py --cpu-sample 0 > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","first_call_ms")}, d["roofline"]["per_kernel"], d["roofline"]["stages_ms_scouting_steps"], d["config"]["pass1_found"], d["config"]["pass2_found"])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $out/bench_prof.json 2> $out/prof.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 0 > $out/timeline.txt
rm -rf $out/rp
cat $out/timeline.txt
