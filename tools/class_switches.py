#!/usr/bin/env python3
"""How often does searchCore leave its residue class on BASELINE configs[3] reads (10 kbp, arrays of 20-60 repeats in 5 % of the
reads)?  After a rejected candidate j = back() - 1 (+ skips) (libcrispr.cpp:390,295): the seed walk continues on another residue
class mod 8.  Counted on the oracle (CPU) over a sample of the bench's own synthetic stream.   python tools/class_switches.py [reads]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = 10000
spec = ca.synth_spec(read_len=L, n_dr=50, crispr_per_million=50000, array_min_repeats=20, array_max_repeats=60)
w = ca.synth_packed(spec, 0, n)
asc = ca.unpack_ascii(w, (L + 15) // 16, L, n)
lib = orc.lib()
lib.orc_stats_get.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
a, b = C.c_uint64(), C.c_uint64()
lib.orc_stats_get(C.byref(a), C.byref(b), 1)
p = orc.Params.default() if hasattr(orc.Params, "default") else None
per = []
found = 0
for i in range(n):
    seq = asc[i * L:(i + 1) * L].tobytes()
    r = orc.search_core(seq) if hasattr(orc, "search_core") else None
    lib.orc_stats_get(C.byref(a), C.byref(b), 1)
    per.append((a.value, b.value))
    found += 1 if (r and r[0]) else 0
per = np.array(per)
print("reads %d (found %d): rejected candidates per read mean %.3f max %d; class switches per read mean %.3f, p90 %d, max %d; reads with 0 switches %.1f %%" % (
    n, found, per[:, 0].mean(), per[:, 0].max(), per[:, 1].mean(), int(np.percentile(per[:, 1], 90)), per[:, 1].max(), 100.0 * (per[:, 1] == 0).mean()))
