#!/usr/bin/env python3
"""The indexed reader on ONE gzip member of N reads (gzip -1 of bench.py's stream), no GPU: the several-thread inflate's phase times
(CRASS_TIMING lines: find + decode, windows + narrowing, CRC-32).   python tools/gz_index_time.py [reads]"""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
td = tempfile.mkdtemp(prefix="crass_gzix_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    fa = os.path.join(td, "r.fa")
    spec = ca.synth_spec(read_len=L)
    with open(fa, "wb") as f:
        for first in range(0, n, 5_000_000):
            m = min(5_000_000, n - first)
            asc = ca.unpack_ascii(ca.synth_packed(spec, first, m), (L + 15) // 16, L, m).reshape(m, L)
            rec = np.empty((m, 10 + L + 1), np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("S8"), 8)
            rec[:, 0] = ord(">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
            rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
            f.write(rec.tobytes())
    subprocess.check_call("gzip -1 -c %s > %s.gz" % (fa, fa), shell=True)
    code = "import sys; sys.path.insert(0, %r); import crass_amd as ca, time; ca.load()\nfor _ in range(3):\n    t = time.perf_counter(); ix = ca.FastxIndex(%r); print('index %%.3f s' %% (time.perf_counter() - t), flush=True); del ix\n" % (
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), fa + ".gz")
    for env in ({}, {}):
        print("==", env or "default", flush=True)
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CRASS_TIMING="1", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for line in p.stdout.decode().splitlines():
            if "inflate:" in line or line.startswith("index"): print("  " + line[:220])
finally:
    shutil.rmtree(td, ignore_errors=True)
