#!/usr/bin/env python3
"""Config-4 shape (BASELINE.json configs[3]): 10 kbp reads, CRISPR arrays of 20-60 repeats in 2 % of them.
Timing of pass 1 + merge + pass 2 with the reads resident in HBM (no 2-bit filter at this length: every read
goes through the wave-per-read survivor kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 10000
rng = np.random.default_rng(42)
acgt = np.frombuffer(b"ACGT", np.uint8)
t0 = time.time()
reads = acgt[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
drs = [acgt[rng.integers(0, 4, size=int(rng.integers(28, 38)))] for _ in range(50)]
for i in np.nonzero(rng.random(n) < 0.02)[0]:
    dr = drs[int(rng.integers(0, 50))]
    pos = int(rng.integers(0, 4000))
    for _ in range(int(rng.integers(20, 61))):
        sp = acgt[rng.integers(0, 4, size=int(rng.integers(30, 39)))]
        unit = np.concatenate([dr, sp])
        if pos + len(unit) > L:
            break
        reads[i, pos:pos + len(unit)] = unit
        pos += len(unit)
print("generated %d x %d in %.1fs" % (n, L, time.time() - t0), flush=True)
seqs = [reads[i].tobytes() for i in range(n)]
del reads
t0 = time.time()
packed = ca.PackedReads(seqs)
print("packed in %.1fs" % (time.time() - t0), flush=True)
eng = ca.SearchEngine(device=0)
eng.load_reads(packed, None)
for it in range(3):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); t1 = time.perf_counter(); eng.merge(fetch=False); t2 = time.perf_counter(); eng.recruit(fetch=False); t3 = time.perf_counter()
    c = eng.counters()
    print("step %d: scan %.2f ms merge %.2f ms recruit %.2f ms total %.2f ms -> %.2f M reads/s, %.2f G bases/s; found %d + %d, patterns %d, device merge %d, survivor kernel %.2f ms" % (
        it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), n / (t3 - t0) / 1e6, n * L / (t3 - t0) / 1e9,
        c["n_pass1_found"], c["n_pass2_found"], c["n_patterns"], c["used_device_merge"], c["ms_survivor"]), flush=True)
eng.close()
