#!/usr/bin/env python3
"""Where the adapter's three getters spend their time after a step of the headline workload (bench.py's abi_fetch_ms is their sum).
   python tools/abi_fetch_parts.py [reads]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from crass_amd import _abi
from crass_amd.engine import _chk
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = 150
e = ca.SearchEngine(device=0)
e.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
rows = []
for it in range(5):
    e.seed_scan(fetch=False); e.merge(fetch=False); e.recruit(fetch=False)
    import torch; torch.cuda.synchronize()
    c, m, q = _abi.Candidates(), _abi.MergeView(), _abi.Recruits()
    t0 = time.perf_counter(); _chk(e.lib.crass_hip_get_candidates(e.h, C.byref(c)), "c")
    t1 = time.perf_counter(); _chk(e.lib.crass_hip_get_merge(e.h, C.byref(m)), "m")
    t2 = time.perf_counter(); _chk(e.lib.crass_hip_get_recruits(e.h, C.byref(q)), "q")
    t3 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    print("step %d: get_candidates %.3f ms (%d), get_merge %.3f ms (%d tokens), get_recruits %.3f ms (%d)" % (it, rows[-1][0], c.n, rows[-1][1], m.n_tokens, rows[-1][2], q.n), flush=True)
print("median: candidates %.3f, merge %.3f, recruits %.3f ms" % tuple(float(np.median([r[i] for r in rows[1:]])) for i in range(3)))
