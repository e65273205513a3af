#!/bin/bash
# end-to-end run of the crass-hip command line on a generated FASTA (GPU box): stage times on stderr
n=${1:-5000000}
cd $GRAFT_REPO_ROOT
python - <<EOF
import numpy as np, sys, time
sys.path.insert(0, ".")
import crass_amd as ca
ca.load()
n, L = $n, 150
spec = ca.synth_spec(read_len=L)
w = ca.synth_packed(spec, 0, n)
asc = ca.unpack_ascii(w, 10, L, n).reshape(n, L)
rec = np.empty((n, 10 + L + 1), np.uint8)
ids = np.char.zfill(np.arange(n).astype("S8"), 8)
rec[:, 0] = ord(">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, 8); rec[:, 9] = 10
rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
open("/tmp/e2e.fa", "wb").write(rec.tobytes())
print("fasta MB", rec.size / 1e6)
EOF
mkdir -p /tmp/e2e_out
for i in 1 2; do ( s=$(date +%s.%N); CRASS_TIMING=1 crass_amd/crass-hip -o /tmp/e2e_out /tmp/e2e.fa 2>&1; e=$(date +%s.%N); python3 -c "print(\"wall %.3f s\" % ($e - $s))" ) | tr '\r' '\n' | grep "timing\|wall\|Found"; done
gzip -1 -k /tmp/e2e.fa
for i in 1 2; do ( s=$(date +%s.%N); CRASS_TIMING=1 crass_amd/crass-hip -o /tmp/e2e_out /tmp/e2e.fa.gz 2>&1; e=$(date +%s.%N); python3 -c "print(\"gz (libdeflate) wall %.3f s\" % ($e - $s))" ) | tr '\r' '\n' | grep "timing\|wall"; done
( s=$(date +%s.%N); CRASS_NO_LIBDEFLATE=1 CRASS_TIMING=1 crass_amd/crass-hip -o /tmp/e2e_out /tmp/e2e.fa.gz 2>&1; e=$(date +%s.%N); python3 -c "print(\"gz (zlib) wall %.3f s\" % ($e - $s))" ) | tr '\r' '\n' | grep "timing\|wall"
rm -f /tmp/e2e.fa /tmp/e2e.fa.gz
