#!/bin/bash
# end-to-end run of the crass-hip command line on a generated FASTA (GPU box): stage times, wall clock and peak RSS
n=${1:-5000000}
cd $GRAFT_REPO_ROOT
python - <<PY
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
PY
gzip -1 -k /tmp/e2e.fa
python - <<'PY'
import os, resource, subprocess, time
def run(label, args, env=None):
    e = dict(os.environ, CRASS_TIMING="1")
    e.update(env or {})
    os.makedirs("/tmp/e2e_out", exist_ok=True)
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t0 = time.time()
    p = subprocess.run(["crass_amd/crass-hip", "-g", "-o", "/tmp/e2e_out"] + args, capture_output=True, env=e)
    dt = time.time() - t0
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    for line in (p.stderr.decode() + p.stdout.decode().replace("\r", "\n")).splitlines():
        if "timing" in line or "Found" in line or "CRISPRs" in line or "true direct" in line:
            print("   ", line.strip())
    print("%-34s wall %.3f s, rc %d, peak RSS of the children so far %.0f MB" % (label, dt, p.returncode, rss / 1024.0))
for i in range(2):
    run(".fa (one call, device merge)", ["/tmp/e2e.fa"])
run(".fa --seam (crass's three calls)", ["--seam", "/tmp/e2e.fa"])
run(".fa, orderly tear-down", ["/tmp/e2e.fa"], {"CRASS_RELEASE_AT_EXIT": "1"})
for i in range(2):
    run(".fa.gz (libdeflate)", ["/tmp/e2e.fa.gz"])
run(".fa.gz (zlib)", ["/tmp/e2e.fa.gz"], {"CRASS_NO_LIBDEFLATE": "1"})
print(sorted(os.listdir("/tmp/e2e_out"))[:12])
PY
rm -f /tmp/e2e.fa /tmp/e2e.fa.gz
