#!/usr/bin/env python3
"""First calls of a GROUP of W contexts sharing cuda:0 (device copies instead of the collective): per-phase wall times of the first
three steps and the bound-overflow counters, to see what the first call still sizes or repeats.   python tools/group_first_call.py [total_reads] [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crass_amd as ca
ca.load()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = 150
spec = ca.synth_spec(read_len=L)
words = ca.synth_packed(spec, 0, total)
g = ca.SearchGroup([0] * W, local_copies=True)
t0 = time.perf_counter(); g.load_packed_uniform(words, total, L); print("load %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
for k in range(4):
    t = [time.perf_counter()]
    g.seed_scan(); t.append(time.perf_counter())
    g.merge(); t.append(time.perf_counter())
    g.recruit(); t.append(time.perf_counter())
    print("step %d: seed %.2f merge %.2f recruit %.2f total %.2f ms; overflows %s fallbacks %s" % (
        k, *[1e3 * (b - a) for a, b in zip(t, t[1:])], 1e3 * (t[-1] - t[0]),
        [g.rank_counters(r)["n_bound_overflows"] for r in range(W)], [g.rank_counters(r)["n_merge_fallbacks"] for r in range(W)]), flush=True)
for k in range(3):
    t0 = time.perf_counter(); g.step(); print("one-dispatch step: %.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
g.close()
