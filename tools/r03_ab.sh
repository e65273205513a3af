#!/bin/bash
# quick A/B session: GPU suite, then the headline bench with and without a switch (same box, back to back)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03h}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -rf -x > $out/pytest.txt 2>&1; tail -6 $out/pytest.txt
B="--steps 40 --warmup 5 --cpu-sample 0 --single-shots 0"
timeout 300 python bench.py --config 1 --cpu-sample 0 --single-shots 0 > $out/bench_c1.json 2>> $out/err.txt
python - $out/bench_*.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], d["step_ms"]["median"], r["stages_ms_scouting_steps"], d["config"]["pass2_found"])
    except Exception as e: print(f, "ERR", e)
PY
