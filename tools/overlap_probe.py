#!/usr/bin/env python3
"""Do the step's latency-bound kernels hide behind its streaming kernels when both are on the device at once?  Two contexts with the
same read set on ONE GPU: K steps each, one context after the other, then both at once from two host threads (their kernels meet at
arbitrary phases).  A concurrent total well below the sequential one says that slicing the seed scan so that the survivor kernel of
slice i runs beside the filter of slice i + 1 would pay.   python tools/overlap_probe.py [reads] [K]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = 150
spec = ca.synth_spec(read_len=L)
w = ca.synth_packed(spec, 0, n)
engs = []
for _ in range(2):
    e = ca.SearchEngine(device=0)
    e.load_packed_uniform(w, n, L)
    for _ in range(3):
        e.seed_scan(fetch=False); e.merge(fetch=False); e.recruit(fetch=False)
    engs.append(e)
def run(e, k=K):
    for _ in range(k):
        e.seed_scan(fetch=False); e.merge(fetch=False); e.recruit(fetch=False)
t0 = time.perf_counter(); run(engs[0]); t1 = time.perf_counter(); run(engs[1]); t2 = time.perf_counter()
print("one after the other: %.3f + %.3f ms per step" % (1e3 * (t1 - t0) / K, 1e3 * (t2 - t1) / K))
for rep in range(3):
    th = [threading.Thread(target=run, args=(e,)) for e in engs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("both at once: %.3f ms per PAIR of steps (%.3f per step)" % (1e3 * dt / K, 1e3 * dt / K / 2))
