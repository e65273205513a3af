#!/bin/bash
# A/B of an environment switch on the default bench workload: tools/ab.sh CRASS_SOME_SWITCH [steps]
sw=$1; steps=${2:-200}
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $sw=1; else unset $sw; fi
    python bench.py --cpu-sample 0 --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sw=$v', 'ms_per_step', d['ms_per_step'], 'value %.3e' % d['value'], d['roofline'].get('stages_ms_scouting_steps'))"
  done
done
