#!/usr/bin/env python3
"""What ONE long read costs: the reference's control flow counted on the oracle (bmpSearch calls, extension columns, Levenshtein
calls per read, split by reads with / without an array) next to the cycles the wave-per-read kernel spends in each phase of such a
read (CRASS_SURV_PROF=1: k_survivor keeps per-read phase counters and sums them per category).

  python tools/longread_phases.py [gpu_reads] [oracle_reads]      (BASELINE configs[3] stream: 10 kbp, arrays of 20-60 repeats in 5 %)
"""
import ctypes as C, os, sys, time
os.environ.setdefault("CRASS_SURV_PROF", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc

n_gpu = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_orc = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
L = 10000
spec = ca.synth_spec(read_len=L, n_dr=50, crispr_per_million=50000, array_min_repeats=20, array_max_repeats=60)

# ---- oracle side: work per read ----
# the work counters are compiled in only here (-DORC_WORK_COUNTERS): the library the tests and bench.py time carries none
import subprocess, tempfile
_od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_so = os.path.join(tempfile.mkdtemp(prefix="orc_work_"), "liboracle_work.so")
subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-D_POSIX_C_SOURCE=200809L", "-DORC_WORK_COUNTERS",
                       "-shared", "-o", _so, os.path.join(_od, "crass_oracle.c"), os.path.join(_od, "crass_consensus.c"), "-lm"])
orc.ORACLE_LIB_OVERRIDE = _so
lib = orc.lib()
lib.orc_work_get.argtypes = [C.POINTER(C.c_uint64), C.c_int]
w = ca.synth_packed(spec, 0, n_orc)
asc = ca.unpack_ascii(w, (L + 15) // 16, L, n_orc)
buf = (C.c_uint64 * 8)()
lib.orc_work_get(buf, 1)
rows = []
for i in range(n_orc):
    seq = asc[i * L:(i + 1) * L].tobytes()
    r = orc.search_core(seq)
    lib.orc_work_get(buf, 1)
    rows.append([1 if r[0] == 1 else 0] + list(buf))
rows = np.array(rows, dtype=np.float64)
names = ["bmp calls", "seed hits", "scanRight calls", "ext columns", "ext votes", "lev calls", "qc calls", "lev cells"]
for found in (0, 1):
    sel = rows[rows[:, 0] == found]
    if len(sel) == 0:
        continue
    print("oracle: %d reads %s: " % (len(sel), "FOUND (array)" if found else "not found") +
          ", ".join("%s %.1f (max %d)" % (names[k], sel[:, 1 + k].mean(), sel[:, 1 + k].max()) for k in range(8)))

# ---- device side ----
if n_gpu <= 0:
    sys.exit(0)
eng = ca.SearchEngine(device=0)
words = ca.synth_packed(spec, 0, n_gpu)
eng.load_packed_uniform(words, n_gpu, L)
for it in range(3):
    t0 = time.time()
    eng.seed_scan(fetch=False)
    sys.stderr.flush()
    print("seed_scan %d: %.2f ms" % (it, (time.time() - t0) * 1e3))
