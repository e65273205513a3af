#!/bin/bash
# ms per step of the search path against the read length (uniform reads, 1.5 Gbases per batch): where the bit-parallel filter and the
# lane kernel stop and the general kernels take over.   bash tools/len_sweep.sh [outdir]
out=${1:-gpurun_out/len_sweep}; mkdir -p $out
for L in ${LENS:-100 150 250 256 300 400 500 700 1000 2000 2100}; do
  n=$((1500000000 / L))
  python bench.py --read-len $L --total-reads $n --steps 5 --warmup 2 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > $out/L$L.json 2> $out/L$L.err
  python - $out/L$L.json $L $n <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pk = j["roofline"]["per_kernel"]
    top = " ".join("%s %.3f" % (k, v["avg_launch_ms"]) for k, v in pk.items())
    print("L=%s n=%s ms/step %.3f  Gbases/s %.1f  %s" % (sys.argv[2], sys.argv[3], j["ms_per_step"], int(sys.argv[2]) * int(sys.argv[3]) / j["ms_per_step"] / 1e6, top))
except Exception as e:
    print("L=%s failed: %r" % (sys.argv[2], e))
P
done
