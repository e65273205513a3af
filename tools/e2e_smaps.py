#!/usr/bin/env python3
"""the resident mappings `crass-hip` holds when it ends (CRASS_SMAPS_AT_EXIT=1), on a FASTA of N synthetic reads"""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
import bench
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = 150
spec = ca.synth_spec(read_len=L)
td = tempfile.mkdtemp(prefix="crass_smaps_", dir=bench._e2e_dir(n * (11 + L)))
try:
    fa = os.path.join(td, "e2e.fa")
    with open(fa, "wb") as f:
        for first in range(0, n, 5_000_000):
            m = min(5_000_000, n - first)
            w = ca.synth_packed(spec, first, m)
            asc = ca.unpack_ascii(w, (L + 15) // 16, L, m).reshape(m, L)
            rec = np.empty((m, 10 + L + 1), np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("S8"), 8)
            rec[:, 0] = ord(">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
            rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
            f.write(rec.tobytes())
    od = os.path.join(td, "out"); os.makedirs(od)
    env = dict(os.environ, CRASS_TIMING="1", CRASS_SMAPS_AT_EXIT="1")
    if os.environ.get("CLI_PRELOAD"):                  # (profiles/ubench/alloc_trace.c: who allocates the mappings of a given size)
        env["LD_PRELOAD"] = os.environ["CLI_PRELOAD"]
    p = subprocess.run([os.path.join(bench.ROOT, "crass_amd", "crass-hip"), "-g", "-o", od, fa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    if os.environ.get("CLI_RAW"):
        open(os.environ["CLI_RAW"], "wb").write(p.stdout)
    for line in p.stdout.decode().replace("\r", "\n").splitlines():
        if "smaps" in line or "peek" in line or "cli:" in line or "resident" in line or "fastx index" in line: print(line)
finally:
    shutil.rmtree(td, ignore_errors=True)
