#!/usr/bin/env python3
"""Host-side wall time of the three calls of a step (seed_scan / merge / recruit), 10 M x 150 bp reads resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
spec = ca.synth_spec(read_len=L)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, L)
def step(acc):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False)
    t1 = time.perf_counter(); eng.merge(fetch=False)
    t2 = time.perf_counter(); eng.recruit(fetch=False)
    t3 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2]
for lvl in (1, 0, 1):
    eng.set_stage_timing(lvl)
    if lvl == 1: eng.set_timing_focus(1)
    for _ in range(5): step(np.zeros(3))
    acc = np.zeros(3); K = 100
    t0 = time.perf_counter()
    for _ in range(K): step(acc)
    tot = (time.perf_counter() - t0) / K
    print("timing level %d: %.1f us/step  seed_scan %.1f  merge %.1f  recruit %.1f" % ((lvl, tot * 1e6) + tuple(acc / K * 1e6)), flush=True)
eng.close()
