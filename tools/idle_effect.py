#!/usr/bin/env python3
"""Does a step cost more when the device was idle in front of it?  Steady-state steps (learnt bounds) of one context at 100 M x
150 bp: back to back, behind a 50 ms pause, and behind a pause followed by ~5 ms of unrelated device work (a torch matmul loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crass_amd as ca
ca.load()
n, L = 100_000_000, 150
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
def step():
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    return 1e3 * (time.perf_counter() - t0)
for _ in range(5): step()
def run(tag, pause, warm_ms):
    v = []
    for _ in range(8):
        if pause: time.sleep(pause)
        if warm_ms:
            t0 = time.perf_counter()
            while 1e3 * (time.perf_counter() - t0) < warm_ms:
                (a @ a).sum().item()
        v.append(step())
    print("%-40s median %.3f ms  min %.3f  max %.3f" % (tag, np.median(v), min(v), max(v)), flush=True)
run("back to back", 0, 0)
run("behind a 50 ms pause", 0.05, 0)
run("behind a 50 ms pause + 5 ms of matmuls", 0.05, 5)
run("behind a 50 ms pause + 30 ms of matmuls", 0.05, 30)
run("behind a 1 ms pause", 0.001, 0)
run("back to back", 0, 0)
eng.close()
