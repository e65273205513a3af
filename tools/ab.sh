#!/bin/bash
# A/B of environment switches on the default bench: prints ms/step and the stage times
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  for rep in 1 2; do
    env $v python bench.py --cpu-sample 0 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s' % '$v', d['ms_per_step'], d['roofline']['stages_ms_scouting_steps'], d['roofline']['host_ms'])"
  done
done
