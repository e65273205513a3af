#!/bin/bash
# same-box A/B of one environment switch: r03_ab.sh <tag> <ENVVAR> [bench args]
set -u
tag=$1; var=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 400 python bench.py --steps 40 --warmup 5 --cpu-sample 0 --single-shots 0 "$@" > $out/a$rep.json 2> $out/a$rep.err
  env $var=1 timeout 400 python bench.py --steps 40 --warmup 5 --cpu-sample 0 --single-shots 0 "$@" > $out/b$rep.json 2> $out/b$rep.err
done
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/[ab]?.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], d["step_ms"], d["roofline"]["host_ms"])
PY
