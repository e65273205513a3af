#!/usr/bin/env python3
"""Pass-1 kernel times (HIP events) of repeated seed scans over BASELINE configs[3] (1 M x 10 kbp, arrays in 5 % of the reads) for a
timing breakdown of the wave-per-read kernel (CRASS_SURV_DEBUG=1 load only, 2 no seed finds, 3 no QC, 4 no output).
python tools/longread_ab.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 10000
spec = ca.synth_spec(read_len=L, n_dr=50, crispr_per_million=50000, array_min_repeats=20, array_max_repeats=60)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, L)
eng.set_stage_timing(2)
f, s, tot = [], [], []
for it in range(8):
    try:
        eng.seed_scan(fetch=False)
    except ca.CrassError as e:
        print("seed_scan:", e); break
    c = eng.counters()
    if it >= 2: f.append(c["ms_filter"]); s.append(c["ms_survivor"]); tot.append(c["ms_pass1_total"])
print("CRASS_SURV_DEBUG=%s  hints %.2f ms  survivor %.2f ms  pass 1 %.2f ms  (survivors %d, found %d)" % (
    os.environ.get("CRASS_SURV_DEBUG", "-"), np.mean(f), np.mean(s), np.mean(tot), c["n_filter_survivors"], c["n_pass1_found"]))
eng.close()
