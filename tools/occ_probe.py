import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import crass_amd as ca
n=200000; L=10000
spec = ca.synth_spec(read_len=L, n_dr=50, crispr_per_million=50000, array_min_repeats=20, array_max_repeats=60)
eng = ca.SearchEngine(device=0)
words = ca.synth_packed(spec, 0, n)
eng.load_packed_uniform(words, n, L)
for it in range(2):
    t0=time.time(); eng.seed_scan(fetch=False); print("scan %.2f ms" % ((time.time()-t0)*1e3))
