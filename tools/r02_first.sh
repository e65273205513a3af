#!/bin/bash
# round-2 first GPU call: new tests, default bench, the other configs at full size
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "known_answer or bench" > $out/pytest_new.txt 2>&1; tail -5 $out/pytest_new.txt
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.err
timeout 400 python bench.py --gpus 1 --config 2 > $out/bench_c2.json 2> $out/bench_c2.err; tail -c 300 $out/bench_c2.err
timeout 600 python bench.py --config 3 --steps 10 --warmup 2 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 300 $out/bench_c3.err
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 > $out/bench_c4.json 2> $out/bench_c4.err; tail -c 300 $out/bench_c4.err
timeout 300 python bench.py --alternate --cpu-sample 0 > $out/bench_alt.json 2> $out/bench_alt.err
for f in $out/bench_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","first_call_ms","scaling")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["path_frac"], d["roofline"]["per_kernel"], d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("calibration"))
except Exception as e: print("ERR", e)
PY
done
