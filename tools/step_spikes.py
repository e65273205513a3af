#!/usr/bin/env python3
"""Which steps are slow?  Runs the default workload for N steps and prints the indices and times of the steps that took
more than 1.5 x the median (wall clock between step ends):  python tools/step_spikes.py [steps]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, L = 10_000_000, 150
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
for _ in range(10):
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
gc.collect(); gc.disable()
marks = np.zeros(K + 1)
marks[0] = time.perf_counter()
for i in range(K):
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    marks[i + 1] = time.perf_counter()
d = np.diff(marks) * 1e3
med = float(np.median(d))
slow = [(int(i), round(float(x), 3)) for i, x in enumerate(d) if x > 1.5 * med]
print("median %.4f ms, mean %.4f ms, %d slow steps:" % (med, float(d.mean()), len(slow)), slow[:40])
eng.close()
