import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import crass_amd as ca
from tests import test_gpu_parity as T
rng = np.random.default_rng(1)
for lo, hi in ((257, 257), (300, 300), (40, 700)):
    seqs = T._mid_reads(rng, 3000, lo, hi)
    for pad in (0, 2):
        r = ca.search_pipeline(seqs, pad_uniform=pad)
        print(lo, hi, "pad", pad, "n_pass1", r.n_pass1, "filter", r.counters["used_fast_filter"], "surv", r.counters["n_filter_survivors"], flush=True)
