#!/bin/bash
# quick GPU check after a survivor-kernel change: parity tests + sweep + the pass-1 kernel times
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_merge.py -x -q 2>&1 | tail -3
timeout 600 python tools/parity_sweep.py ${SWEEP:-200} ${SEED:-11} 2>&1 | tail -2
python tools/survivor_ab.py 2>&1 | grep -v amdgpu | tail -1
