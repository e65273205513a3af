#!/usr/bin/env python3
"""Which steps are slow?  Runs the default workload for N steps and prints the indices and times of the steps that took
more than 1.5 x the median (wall clock between step ends):  python tools/step_spikes.py [steps]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, L = int(os.environ.get("READS", "10000000")), 150
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
for _ in range(10):
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
gc.collect(); gc.disable()
stages = os.environ.get("STAGES") == "1"          # every stage bracketed by HIP events: is a slow step slow on the device?
if stages:
    eng.set_stage_timing(2)
marks = np.zeros(K + 1)
parts = np.zeros((K, 3))
dev = []
def _cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}
cs0 = _cpu_stat()
cpu0 = time.process_time()
marks[0] = time.perf_counter()
for i in range(K):
    t0 = time.perf_counter(); eng.seed_scan(fetch=False); t1 = time.perf_counter(); eng.merge(fetch=False); t2 = time.perf_counter(); eng.recruit(fetch=False)
    marks[i + 1] = time.perf_counter()
    parts[i] = (t1 - t0, t2 - t1, marks[i + 1] - t2)
    if stages:
        c = eng.counters()
        dev.append((c["ms_pass1_total"], c["ms_merge_device"], c["ms_pass2_total"]))
cs1 = _cpu_stat()
print("cgroup cpu.stat over the loop:", {k: cs1[k] - cs0.get(k, 0) for k in cs1 if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")})
print("CPUs kept busy by this process during the loop: %.1f" % ((time.process_time() - cpu0) / (marks[-1] - marks[0])))
d = np.diff(marks) * 1e3
med = float(np.median(d))
slow = [(int(i), round(float(x), 3)) for i, x in enumerate(d) if x > 1.5 * med]
print("median %.4f ms, mean %.4f ms, %d slow steps:" % (med, float(d.mean()), len(slow)), slow[:40])
print("host wall clock of the calls (seed_scan, merge, recruit), median:", np.round(np.median(parts, axis=0) * 1e3, 3))
for i, x in slow[:12]:
    print("  step %d: %.3f ms = calls" % (i, x), np.round(parts[i] * 1e3, 3), ("device stages (pass 1, merge, pass 2) %s" % (np.round(dev[i], 3),)) if stages else "")
eng.close()
