#!/bin/bash
# memory-hygiene run (csrc/devmem.h): the GPU suite with every device buffer between unmapped ranges, one pytest process
# per file so that a fault in one does not hide the others; the parity sweep the same way; then the list of tests that
# depend on the zero fill of fresh buffers (CRASS_POISON)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03e}
mkdir -p $out
cd $GRAFT_REPO_ROOT
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  CRASS_GUARD_PAGES=1 timeout 600 python -m pytest $f -m gpu -q --timeout=300 > $out/guard_$b.txt 2>&1; rc=$?
  echo "guard $b rc=$rc $(tail -1 $out/guard_$b.txt)"
  if [ $rc -ne 0 ]; then grep -n "Memory access\|^FAILED" $out/guard_$b.txt | head -8; fi
done
CRASS_GUARD_PAGES=1 timeout 900 python tools/parity_sweep.py 200 43 > $out/guard_sweep.txt 2>&1; echo "guard sweep rc=$?"; tail -2 $out/guard_sweep.txt
timeout 900 python tools/parity_sweep.py 300 43 > $out/sweep.txt 2>&1; echo "sweep rc=$?"; tail -2 $out/sweep.txt
if [ "${POISON:-0}" = 1 ]; then
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  CRASS_POISON=1 timeout 600 python -m pytest $f -m gpu -q --timeout=300 > $out/poison_$b.txt 2>&1; rc=$?
  echo "poison $b rc=$rc $(tail -1 $out/poison_$b.txt)"
done
fi
