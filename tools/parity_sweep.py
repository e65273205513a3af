#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: many synthetic read sets with varied shapes through the HIP path and
the oracle, field-by-field comparison (tests/parity.py).  Not part of the test suite (minutes of oracle time);
run it after touching a kernel:  python tools/parity_sweep.py [n_cases] [seed]"""
import os, sys, random, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
from tests.parity import assert_same_pipeline
ca.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = ca.SearchEngine()          # one context reused: speculative bounds, buffer reuse, helper thread all get exercised
bad = 0
t_start = time.time()
for case in range(n_cases):
    L = rng.choice([75, 100, 101, 125, 150, 150, 150, 151, 250, 251, 64, 200])
    n = rng.choice([3000, 20000, 60000, 150000])
    if rng.random() < 0.25:                      # reads between the lane-per-read filter and the long-read path: the hint filter
        L = rng.choice([257, 300, 301, 400, 512, 513, 700, 1000, 1500, 2048])
        n = max(1000, min(n, 9_000_000 // L))
    n_dr = rng.choice([1, 3, 10, 50, 200, 1500])
    cpm = rng.choice([2000, 10000, 50000, 300000])
    k = rng.choice([6, 6, 6, 4, 8, 12])
    # (round 6) a tenth of the cases with SHORT repeats and -d 19 .. 22: the device merge from 19 bases, pass 2's windows every 4 bases
    short_dr = rng.random() < 0.1
    skw = {}
    if short_dr:
        lo = rng.choice([19, 20, 21, 22])
        skw = dict(dr_len_min=lo, dr_len_max=lo + rng.choice([3, 6, 12]), spacer_len_min=26, spacer_len_max=36)
    spec = ca.synth_spec(read_len=L, n_dr=n_dr, crispr_per_million=cpm, seed=rng.randrange(1 << 30), **skw)
    w = ca.synth_packed(spec, rng.randrange(1 << 20), n)
    asc = ca.unpack_ascii(w, (L + 15) // 16, L, n)
    seqs = [asc[i * L:(i + 1) * L].tobytes() for i in range(n)]
    ragged = rng.random() < 0.25
    if ragged:
        seqs = [s[:rng.randint(40, L)] if rng.random() < 0.3 else s for s in seqs]
    with_n = rng.random() < 0.35
    if with_n:                                   # exception reads: a byte outside ACGT somewhere
        alphabet = rng.choice([b"N", b"N", b"NNNna"])     # N only: DR variants with an N stay on the device merge
        for i in rng.sample(range(n), max(1, n // rng.choice([5, 20, 100, 400]))):
            b = bytearray(seqs[i])
            for _ in range(rng.choice([1, 1, 3])):
                b[rng.randrange(len(b))] = rng.choice(alphabet)
            seqs[i] = bytes(b)
    # a quarter of the cases with other search options (-w / -d / -D / -s / -S / -n): the run-time-range and every-position forms of
    # the bit-parallel filter when the stride is uniform, k_filter_general otherwise
    okw = {}
    if rng.random() < 0.25:
        okw = rng.choice([dict(searchWindowLength=6), dict(searchWindowLength=7), dict(searchWindowLength=9), dict(lowDRsize=20, highDRsize=40),
                          dict(lowDRsize=30, highDRsize=60), dict(lowSpacerSize=20, highSpacerSize=60), dict(lowSpacerSize=30, highSpacerSize=40),
                          dict(highDRsize=64, highSpacerSize=70), dict(minNumRepeats=3), dict(lowDRsize=15, searchWindowLength=8),
                          dict(lowDRsize=25, searchWindowLength=9, lowSpacerSize=22), dict(lowDRsize=17, highDRsize=35, searchWindowLength=6, lowSpacerSize=15, highSpacerSize=45)])
    if short_dr:
        okw = dict(lowDRsize=skw["dr_len_min"], highDRsize=skw["dr_len_min"] + 22)
    p = ca.default_params(kmer_clust_size=k, **okw)
    host = rng.random() < 0.15
    use_eng = rng.random() < 0.7 and k == 6 and not okw
    pad = rng.choice([0, 2])
    # a third of the cases go through a GROUP of 2-4 contexts sharing the GPU (crass_hip_group_*: contiguous shards, the
    # exchange as device copies, one host view), sometimes with duplicate headers across the shards and a tiny exchange buffer
    grp = rng.choice([0, 0, 2, 3, 4])
    hdrs = None
    if grp and rng.random() < 0.3:
        hdrs = [b"h%d" % (i if rng.random() > 0.05 else rng.randrange(0, i + 1)) for i in range(n)]
    small_cap = grp and rng.random() < 0.2
    fused = bool(grp) and rng.random() < 0.5
    tag = "L=%d n=%d n_dr=%d cpm=%d k=%d ragged=%d host=%d N=%d grp=%d%s%s%s pad=%d eng=%d %s" % (
        L, n, n_dr, cpm, k, ragged, host, with_n, grp, "h" if hdrs else "", "c" if small_cap else "", "f" if fused else "", pad, use_eng,
        ",".join("%s=%s" % kv for kv in okw.items()))
    only = os.environ.get("ONLY")                # comma-separated case indices: replay just those (same random stream)
    if only and case not in {int(x) for x in only.split(",")}:
        continue
    print("run  case %d %s" % (case, tag), file=sys.stderr, flush=True)
    if host:
        os.environ["CRASS_HOST_MERGE"] = "1"
    if small_cap:
        os.environ["CRASS_GROUP_CAP_ROWS"] = "64"
    try:
        if grp:
            gpu = ca.search_pipeline_group(seqs, [0] * grp, headers=hdrs, params=p, local_copies=True, pad_uniform=pad, fused=fused)
            gpu.counters = gpu.counters[0]
        else:
            gpu = ca.search_pipeline(seqs, params=p, engine=eng if use_eng else None, pad_uniform=pad)
    finally:
        os.environ.pop("CRASS_HOST_MERGE", None)
        os.environ.pop("CRASS_GROUP_CAP_ROWS", None)
    ref = orc.pipeline(seqs, hdrs, params=orc.Params(p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize, p.searchWindowLength,
                                               p.minNumRepeats, p.kmer_clust_size))
    try:
        assert_same_pipeline(gpu, ref)
        print("ok   %-60s pass1 %6d pass2 %6d groups %4d patterns %5d devmerge %d mixed tokens %d" % (
            tag, gpu.n_pass1, gpu.n_pass2, gpu.n_groups, gpu.n_patterns, gpu.counters["used_device_merge"],
            sum(1 for t in ref.tokens if not set(t) <= set(b"ACGT"))), flush=True)
    except AssertionError as e:
        bad += 1
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("FAIL case %d %s: %s:%d %s | %s" % (case, tag, os.path.basename(tb.filename), tb.lineno, tb.line, str(e)[:300]), flush=True)
        if os.environ.get("DUMP_FAIL"):
            import pickle
            with open(os.path.join(os.environ["DUMP_FAIL"], "case%d.pkl" % case), "wb") as fh:
                pickle.dump(dict(seqs=seqs, k=k, pad=pad), fh)
print("%d cases, %d failures, %.0fs" % (n_cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
