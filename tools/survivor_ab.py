#!/usr/bin/env python3
"""Pass-1 kernel times (HIP events) of repeated seed scans over 10 M x 150 bp reads, for A/B runs of the survivor kernel
(CRASS_SURV_DEBUG=1..4 cut it short for a timing breakdown: 1 load only, 2 seed finds only, 3 no QC, 4 no output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
eng.set_stage_timing(2)
f, s = [], []
for it in range(12):
    try:
        eng.seed_scan(fetch=False)
    except ca.CrassError as e:
        print("seed_scan:", e); break
    c = eng.counters()
    if it >= 2: f.append(c["ms_filter"]); s.append(c["ms_survivor"])
print("CRASS_SURV_DEBUG=%s  filter %.1f us  survivor %.1f us  (survivors %d, found %d)" % (os.environ.get("CRASS_SURV_DEBUG", "-"), 1e3 * np.mean(f), 1e3 * np.mean(s), c["n_filter_survivors"], c["n_pass1_found"]))
eng.close()
