#!/usr/bin/env python3
"""Stage-by-stage run of the device merge path with progress prints (debugging aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
spec = ca.synth_spec(read_len=150, n_dr=50, crispr_per_million=10000)
words = ca.synth_packed(spec, 0, n)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(words, n, 150)
for it in range(3):
    t0 = time.time(); eng.seed_scan(fetch=False); print("seed_scan %.3f ms" % ((time.time() - t0) * 1e3), flush=True)
    t0 = time.time(); eng.merge(fetch=False); print("merge %.3f ms" % ((time.time() - t0) * 1e3), flush=True)
    t0 = time.time(); eng.recruit(fetch=False); print("recruit %.3f ms" % ((time.time() - t0) * 1e3), flush=True)
    c = eng.counters()
    print({k: c[k] for k in ("n_pass1_found", "n_pass2_found", "n_patterns", "used_device_merge", "ms_merge_device", "ms_merge_host", "anchor_keys", "anchor_table_kind")}, flush=True)
mv = eng.merge_view()
print("groups", len(mv.groups), "patterns", len(mv.patterns), flush=True)
eng.close()
