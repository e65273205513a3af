#!/bin/bash
# kernel timelines of a FRESH context's first step and of its fourth (steady state), 100 M x 150 bp:  tools/single_shot_tl.sh <tag>
set -u
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cat > /tmp/ss_steps.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import crass_amd as ca
ca.load()
n, L = 100_000_000, 150
w = ca.synth_packed(ca.synth_spec(read_len=L), 0, n)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(w, n, L)
import torch; torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); t1 = time.perf_counter(); eng.merge(fetch=False); t2 = time.perf_counter(); eng.recruit(fetch=False); t3 = time.perf_counter()
    print("call%d total %.3f seed %.3f merge %.3f recruit %.3f" % ((i,) + tuple(1e3 * x for x in (t3 - t0, t1 - t0, t2 - t1, t3 - t2))), eng.counters()["n_bound_overflows"], flush=True)
    time.sleep(0.05)
eng.close()
PY
cd /tmp; export TMPDIR=/tmp
python3 /tmp/ss_steps.py > $out/steps_plain.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/rp -o r -- python3 /tmp/ss_steps.py > $out/steps.txt 2> $out/tl.err
echo "== first step of a fresh context" > $out/timeline.txt
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 3 >> $out/timeline.txt 2>&1
echo "== fourth step" >> $out/timeline.txt
python3 $GRAFT_REPO_ROOT/tools/timeline.py $out/rp 0 >> $out/timeline.txt 2>&1
rm -rf $out/rp
cat $out/steps_plain.txt $out/steps.txt $out/timeline.txt
