#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = 150
words = ca.synth_packed(ca.synth_spec(read_len=L), 0, n)
eng = ca.SearchEngine()
eng.load_packed_uniform(words, n, L)
eng.seed_scan(); eng.merge()
rs = [eng.recruit() for _ in range(6)]
for i in range(1, 6):
    d = np.nonzero((rs[i].start != rs[0].start) | ((rs[i].low_lexi & 1) != (rs[0].low_lexi & 1)))[0]
    print("recruit", i, "vs 0: n", rs[i].n, rs[0].n, "start diffs", len(d), "low diffs", int((rs[i].low_lexi != rs[0].low_lexi).sum()), "slots", d[:8].tolist(), "reads", rs[0].read_idx[d[:4]].tolist())

eng.close()
