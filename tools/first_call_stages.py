import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import crass_amd as ca
ca.load()
n, L = 100_000_000, 150
spec = ca.synth_spec(read_len=L)
w = ca.synth_packed(spec, 0, n)
def run(eng, tag):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); t1 = time.perf_counter(); eng.merge(fetch=False); t2 = time.perf_counter(); eng.recruit(fetch=False); t3 = time.perf_counter()
    c = eng.counters()
    print(tag, "total %.3f seed %.3f merge %.3f recruit %.3f |" % tuple(1e3 * x for x in (t3 - t0, t1 - t0, t2 - t1, t3 - t2)),
          {k: round(c[k], 3) for k in ("ms_filter", "ms_survivor", "ms_pass1_total", "ms_merge_device", "ms_recruit", "ms_recruit_finish", "ms_pass2_total", "ms_merge_host", "ms_sink_host")}, c["n_bound_overflows"], flush=True)
for rep in range(3):
    eng = ca.SearchEngine(device=0)
    eng.set_stage_timing(2)
    eng.load_packed_uniform(w, n, L)
    for i in range(4):
        run(eng, "ctx%d call%d" % (rep, i))
    eng.close()
