#!/bin/bash
# same-box A/B of library builds WITHOUT the profiler (rocprofv3 turns the D2H copies into blit kernels that stretch
# whatever runs beside them): tools/ab_plain.sh <bench args...> -- lib1.so lib2.so ...
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for l in "$@"; do
    cp $l crass_amd/libcrass_hip.so
    python bench.py --cpu-sample 0 "${args[@]}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $l .so)', 'ms_per_step', d['ms_per_step'], d.get('step_ms'), d['roofline'].get('stages_ms_scouting_steps'))"
  done
done
