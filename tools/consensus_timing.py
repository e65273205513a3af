#!/usr/bin/env python3
"""The stage behind the hot path (crass_hip_consensus = WorkHorse::findConsensusDRs) on the hand-off of
BASELINE.json configs[1] (N synthetic 150 bp reads): wall time on the GPU path, the oracle (single core, same algorithm
class as the reference: scalar ksw, full-matrix double Smith-Waterman, full-matrix Levenshtein) beside it, and the check
that both give the same result.  Prints one JSON line.   python tools/consensus_timing.py [n_reads]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
spec = ca.synth_spec(read_len=L)
words = ca.synth_packed(spec, 0, n)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(words, n, L)
cand = eng.seed_scan(); merge = eng.merge(); rec = eng.recruit(); merge = eng.merge_view()
res = ca.engine.PipelineResult(cand, merge, rec, cand.max_read_len)
eng.close()
# only the found reads' sequences travel (the stage works on the ReadHolders): compact read table
reads = np.unique(res.rec_read)
remap = {int(r): i for i, r in enumerate(reads)}
asc = ca.unpack_ascii(words, 10, L, n).reshape(n, L)[reads.astype(np.int64)].reshape(-1).copy()
off = np.arange(0, (len(reads) + 1) * L, L, dtype=np.uint64)
res.rec_read = np.array([remap[int(r)] for r in res.rec_read], np.uint64)
ca.consensus((asc, off), res)                          # warm-up (allocations, code objects)
t0 = time.perf_counter(); gpu = ca.consensus((asc, off), res); t_wrap = time.perf_counter() - t0
t_gpu = ca.consensus.last_call_s                          # crass_hip_consensus itself; t_wrap adds the Python wrapper's array conversions
calls = []
for _ in range(3):
    ca.consensus((asc, off), res); calls.append(ca.consensus.last_call_s)
t_gpu = float(np.median(calls + [t_gpu]))
t0 = time.perf_counter(); ref = orc.consensus((asc, off), res); t_cpu = time.perf_counter() - t0
same = (gpu.true_drs == ref.true_drs and gpu.gids == ref.gids and gpu.groups == ref.groups and gpu.tokens == ref.tokens and
        gpu.reads_of == ref.reads_of and np.array_equal(gpu.ss_pool, ref.ss_pool) and np.array_equal(gpu.rec_rc, ref.rec_rc) and
        np.array_equal(gpu.rec_alive, ref.rec_alive))
print(json.dumps({"stage": "findConsensusDRs (f-1)", "reads": n, "records": int(res.n_pass1 + res.n_pass2), "groups_in": len(res.groups),
                  "true_drs": len(gpu.gids), "gpu_s": round(t_gpu, 4), "gpu_s_with_python_wrapper": round(t_wrap, 4), "oracle_1core_s": round(t_cpu, 3), "speedup": round(t_cpu / t_gpu, 1),
                  "records_per_s_gpu": round((res.n_pass1 + res.n_pass2) / t_gpu, 1), "identical_to_oracle": bool(same), "counters": gpu.counters}))
