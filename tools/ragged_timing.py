#!/usr/bin/env python3
"""Ragged short reads (trimmed Illumina-like: lengths 90..150): which kernels run and how fast."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from crass_amd import _abi
import ctypes as C
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
L = 150
spec = ca.synth_spec(read_len=L)
w = ca.synth_packed(spec, 0, n)
asc = ca.unpack_ascii(w, 10, L, n).reshape(n, L)
rng = np.random.default_rng(3)
lens = rng.integers(90, 151, size=n).astype(np.uint64)
lens[rng.random(n) < 0.5] = 150
if os.environ.get("UNIFORM"):
    lens[:] = 150
n_frac = float(os.environ.get("NFRAC", "0"))
if n_frac > 0:                                     # reads with an N somewhere (exception reads)
    rows = np.nonzero(rng.random(n) < n_frac)[0]
    asc[rows, rng.integers(0, 90, size=len(rows))] = ord("N")
if n_frac > 0 and os.environ.get("CLEAN_DR"):
    # keep the N's out of the repeats: reads whose pass-1 DR would contain one get their original bases back
    # (the device merge takes pure-ACGT DR strings; anything else is merged on the host)
    assert os.environ.get("UNIFORM")
    orig = ca.unpack_ascii(w, 10, L, n).reshape(n, L)
    for _ in range(3):
        eng0 = ca.SearchEngine(device=0)
        pk0 = ca.PackedReads((np.ascontiguousarray(asc.reshape(-1)), np.arange(n + 1, dtype=np.uint64) * np.uint64(L)))
        eng0.load_reads(pk0, None)
        cands = eng0.seed_scan()
        chars, clen = eng0.candidate_dr_view()
        inside = np.arange(chars.shape[1])[None, :] < clen[:, None]
        bad = ((~np.isin(chars, np.frombuffer(b"ACGT", np.uint8))) & inside).any(axis=1)
        rows = np.asarray(cands.read_idx)[bad].astype(np.int64)
        eng0.close(); pk0.close()
        print("reads with a non-ACGT DR:", len(rows), flush=True)
        if len(rows) == 0:
            break
        asc[rows] = orig[rows]
off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum(lens)
t0 = time.time()
mask = np.arange(L)[None, :] < lens[:, None]
flat = asc[mask]                                 # ragged concatenation
print("ragged concat %.1fs, %d bytes" % (time.time() - t0, flat.size), flush=True)
lib = _abi.load()
pk = _abi.Packed()
t0 = time.time()
st = lib.crass_pack_reads(flat.ctypes.data, off.ctypes.data, n, int(os.environ.get('PAD', '2')), C.byref(pk))
assert st == 0
print("pack %.2fs stride_words %d uniform_len %d" % (time.time() - t0, pk.reads.stride_words, pk.reads.uniform_len), flush=True)
eng = ca.SearchEngine(device=0)
st = lib.crass_hip_load_reads(eng.h, C.byref(pk.reads)); assert st == 0
for it in range(4):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); t1 = time.perf_counter(); eng.merge(fetch=False); t2 = time.perf_counter(); eng.recruit(fetch=False); t3 = time.perf_counter()
    c = eng.counters()
    print("step %d: scan %.2f merge %.2f recruit %.2f total %.2f ms -> %.2f G reads/s" % (it, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t3-t0), n/(t3-t0)/1e9),
          {k: (round(c[k], 3) if isinstance(c[k], float) else c[k]) for k in ("ms_filter", "ms_survivor", "ms_recruit", "ms_recruit_finish", "ms_merge_device", "used_fast_filter", "used_device_merge", "n_filter_survivors", "n_pass1_found", "n_pass2_found")}, flush=True)
eng.close()
