import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import crass_amd as ca
ca.load()
n, L = 10_000_000, 150
spec = ca.synth_spec(read_len=L)
w = ca.synth_packed(spec, 0, n)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(w, n, L)
def step():
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
for _ in range(5): step()
for mode in ("plain", "counters", "plain", "counters"):
    t0 = time.perf_counter()
    for _ in range(50):
        step()
        if mode == "counters": eng.counters()
    dt = (time.perf_counter() - t0) / 50
    print(mode, "%.1f us/step" % (dt * 1e6), flush=True)
for lvl in (0, 1, 2):
    eng.set_stage_timing(lvl)
    for _ in range(3): step()
    t0 = time.perf_counter()
    for _ in range(50): step()
    print("level", lvl, "%.1f us/step" % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
eng.close()
