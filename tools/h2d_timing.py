#!/usr/bin/env python3
"""PCIe-inclusive rate of the default workload: packed reads start in HOST memory (what a caller of the C ABI hands over),
one load (H2D) + one step (pass 1 + merge + pass 2):  python tools/h2d_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n, L = 10_000_000, 150
words = ca.synth_packed(ca.synth_spec(read_len=L), 0, n)
eng = ca.SearchEngine(device=0)
for it in range(4):
    t0 = time.perf_counter()
    eng.load_packed_uniform(words, n, L)
    t1 = time.perf_counter()
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    t2 = time.perf_counter()
    print("run %d: load (H2D of %.0f MB from pageable host memory) %.2f ms = %.1f GB/s, step %.2f ms, together %.2f ms -> %.2f G reads/s" % (
        it, words.nbytes / 1e6, 1e3 * (t1 - t0), words.nbytes / (t1 - t0) / 1e9, 1e3 * (t2 - t1), 1e3 * (t2 - t0), n / (t2 - t0) / 1e9), flush=True)
eng.close()
