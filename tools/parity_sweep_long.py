#!/usr/bin/env python3
"""Randomised parity sweep for LONG reads on the GPU box (the walking kernel, position hints, the host-loop sink): read sets of
2-14 kbp with repeat arrays of varied length, spacing and mutation rate through the HIP path and the oracle, field by field
(tests/parity.py).  Each case also draws the long-read switches that are read per call (hint slices, ASCII window, full layout);
CRASS_SINK_EAGER / CRASS_HL_HOST_DEDUPE are read once per process: set them outside for a whole sweep.  Not part of the test
suite:  python tools/parity_sweep_long.py [n_cases] [seed]"""
import os, sys, random, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
from tests.parity import assert_same_pipeline
ca.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
nrg = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", np.uint8)
bad = 0
t_start = time.time()
for case in range(n_cases):
    n = rng.choice([200, 600, 1500, 4500])
    lo = rng.choice([2049, 2100, 3000, 6000])
    hi = rng.choice([lo + 1, 5000, 9000, 14000])
    if hi <= lo:
        hi = lo + 1
    uniform = rng.random() < 0.25
    n_dr = rng.choice([1, 4, 12, 40])
    every = rng.choice([3, 10, 40])
    mut = rng.choice([0.0, 0.0, 0.02, 0.06])                 # per-base substitution rate inside the repeats
    drs = [acgt[nrg.integers(0, 4, size=int(nrg.integers(23, 48)))] for _ in range(n_dr)]
    seqs = []
    for i in range(n):
        L = lo if uniform else int(nrg.integers(lo, hi + 1))
        s = acgt[nrg.integers(0, 4, size=L)].copy()
        if i % every == 0:
            for _ in range(rng.choice([1, 1, 2])):          # one or two arrays in the read
                dr = drs[int(nrg.integers(0, n_dr))]
                pos = int(nrg.integers(0, max(1, L - 200)))
                for _ in range(int(nrg.integers(2, 60))):
                    d = dr.copy()
                    if mut:
                        m = nrg.random(len(d)) < mut
                        d[m] = acgt[nrg.integers(0, 4, size=int(m.sum()))]
                    unit = np.concatenate([d, acgt[nrg.integers(0, 4, size=int(nrg.integers(26, 51)))]])
                    if pos + len(unit) > L:
                        break
                    s[pos:pos + len(unit)] = unit
                    pos += len(unit)
        seqs.append(s.tobytes())
    with_n = rng.random() < 0.3
    if with_n:
        for i in rng.sample(range(n), max(1, n // rng.choice([10, 50, 200]))):
            b = bytearray(seqs[i])
            for _ in range(rng.choice([1, 3])):
                b[rng.randrange(len(b))] = ord("N")
            seqs[i] = bytes(b)
    env = {}
    if rng.random() < 0.5:
        env["CRASS_HINT_PARTS"] = rng.choice(["1", "2", "3", "4"])
    r = rng.random()
    if r < 0.25:
        env["CRASS_SEQ_WINDOW"] = rng.choice(["512", "1024", "2048"])
    elif r < 0.4:
        env["CRASS_LONG_FULL_LAYOUT"] = "1"
    if rng.random() < 0.15:
        env["CRASS_NO_LIGHT"] = "1"                          # the full wave kernel for every read (the A/B of k_long_light)
    # a third of the cases with non-default options: any -w/-d/-D/-s/-S change drops the position hints and the
    # light walk — the un-hinted wave kernel —, -n changes what is accepted only
    kw = {}
    if rng.random() < 0.33:
        kw = rng.choice([dict(searchWindowLength=7), dict(searchWindowLength=6), dict(searchWindowLength=9), dict(lowDRsize=20, highDRsize=40),
                         dict(lowSpacerSize=20, highSpacerSize=60), dict(minNumRepeats=3), dict(minNumRepeats=4), dict(searchWindowLength=9, minNumRepeats=3)])
    params = ca.default_params(**kw) if kw else None
    oparams = orc.Params(params.lowDRsize, params.highDRsize, params.lowSpacerSize, params.highSpacerSize, params.searchWindowLength,
                         params.minNumRepeats, params.kmer_clust_size) if kw else None
    # a quarter of the cases through a group of 2-3 contexts sharing the GPU (contiguous shards, the exchange as device copies)
    grp = rng.choice([0, 0, 0, 2, 3]) if os.environ.get("LONG_SWEEP_GROUPS", "1") != "0" else 0
    fused = bool(grp) and rng.random() < 0.5
    tag = "n=%d L=%d..%d%s n_dr=%d every=%d mut=%.2f N=%d grp=%d%s %s %s" % (n, lo, lo if uniform else hi, "u" if uniform else "", n_dr, every, mut, with_n,
                                                                grp, "f" if fused else "", " ".join("%s=%s" % (k[6:], v) for k, v in sorted(env.items())),
                                                                ",".join("%s=%s" % kv for kv in kw.items()))
    only = os.environ.get("ONLY")
    if only and case not in {int(x) for x in only.split(",")}:
        continue
    print("run  case %d %s" % (case, tag), file=sys.stderr, flush=True)
    os.environ.update(env)
    try:
        if grp:
            gpu = ca.search_pipeline_group(seqs, [0] * grp, params=params, local_copies=True, fused=fused)
            gpu.counters = gpu.counters[0]
        else:
            gpu = ca.search_pipeline(seqs, params=params)
    finally:
        for k in env:
            os.environ.pop(k, None)
    ref = orc.pipeline(seqs, params=oparams) if kw else orc.pipeline(seqs)
    try:
        assert_same_pipeline(gpu, ref)
        print("ok   %-78s pass1 %5d pass2 %5d groups %3d devmerge %d" % (tag, gpu.n_pass1, gpu.n_pass2, gpu.n_groups,
                                                                          gpu.counters["used_device_merge"]), flush=True)
    except AssertionError as e:
        bad += 1
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("FAIL case %d %s: %s:%d %s | %s" % (case, tag, os.path.basename(tb.filename), tb.lineno, tb.line, str(e)[:300]), flush=True)
print("%d cases, %d failures, %.0fs" % (n_cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
