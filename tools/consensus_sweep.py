#!/usr/bin/env python3
"""Randomised parity sweep of the stage behind the search (findConsensusDRs, crass_hip_consensus) on the GPU box: synthetic read
sets of varied length, repeat count and CRISPR density — some ragged, some with N bytes — through the HIP search + consensus and
the oracle's, every field compared (tests/test_gpu_consensus.py: assert_same_consensus).  Not part of the test suite:
python tools/consensus_sweep.py [n_cases] [seed]"""
import os, sys, random, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
from tests.test_gpu_consensus import assert_same_consensus
ca.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_start = time.time()
for case in range(n_cases):
    L = rng.choice([100, 101, 125, 150, 150, 250])
    n = rng.choice([5000, 20000, 60000, 150000])
    n_dr = rng.choice([1, 3, 10, 50, 200])
    cpm = rng.choice([5000, 20000, 50000, 200000])
    spec = ca.synth_spec(read_len=L, n_dr=n_dr, crispr_per_million=cpm, seed=rng.randrange(1 << 30))
    w = ca.synth_packed(spec, rng.randrange(1 << 20), n)
    asc = ca.unpack_ascii(w, (L + 15) // 16, L, n)
    seqs = [asc[i * L:(i + 1) * L].tobytes() for i in range(n)]
    ragged = rng.random() < 0.3
    if ragged:
        seqs = [s[:rng.randint(50, L)] if rng.random() < 0.3 else s for s in seqs]
    with_n = rng.random() < 0.3
    if with_n:
        for i in rng.sample(range(n), max(1, n // rng.choice([20, 100, 400]))):
            b = bytearray(seqs[i])
            b[rng.randrange(len(b))] = ord("N")
            seqs[i] = bytes(b)
    tag = "L=%d n=%d n_dr=%d cpm=%d ragged=%d N=%d" % (L, n, n_dr, cpm, ragged, with_n)
    only = os.environ.get("ONLY")                # comma-separated case indices: replay just those (same random stream)
    if only and case not in {int(x) for x in only.split(",")}:
        continue
    print("run  case %d %s" % (case, tag), file=sys.stderr, flush=True)
    try:
        gpu = ca.consensus(seqs, ca.search_pipeline(seqs))
        ref = orc.consensus(seqs, orc.pipeline(seqs))
        if gpu.error and gpu.error == ref.error:
            # an input on which the reference itself throws / does not terminate (crass_cons_view.error): both sides say so
            print("ok   %-50s both refuse the input with the same code %d" % (tag, gpu.error), flush=True)
            continue
        assert gpu.error == ref.error, (gpu.error, ref.error)
        assert_same_consensus(gpu, ref)
        print("ok   %-50s groups %3d true DRs %3d alignments %6d placements %7d flips %6d" % (
            tag, len(gpu.gids), len(gpu.true_drs), gpu.counters["n_ksw_alignments"], gpu.counters["n_placements"], gpu.counters.get("n_flips", 0)), flush=True)
    except AssertionError as e:
        bad += 1
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("FAIL case %d %s: %s:%d %s | %s" % (case, tag, os.path.basename(tb.filename), tb.lineno, tb.line, str(e)[:300]), flush=True)
print("%d cases, %d failures, %.0fs" % (n_cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
