#!/usr/bin/env python3
"""Replay one parity_sweep case (pickled with DUMP_FAIL=dir) and show where the device merge and the oracle part ways."""
import os, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
from tests import orc
ca.load()
d = pickle.load(open(sys.argv[1], "rb"))
seqs, k, pad = d["seqs"], d["k"], d["pad"]
p = ca.default_params(kmer_clust_size=k)
ref = orc.pipeline(seqs, params=orc.Params(p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize, p.searchWindowLength, p.minNumRepeats, p.kmer_clust_size))
for mode in ("device", "host"):
    if mode == "host": os.environ["CRASS_HOST_MERGE"] = "1"
    else: os.environ.pop("CRASS_HOST_MERGE", None)
    g = ca.search_pipeline(seqs, params=p, pad_uniform=pad)
    print(mode, "groups", g.n_groups, "ref", ref.n_groups, "tokens", g.n_tokens, ref.n_tokens, "patterns", g.n_patterns, ref.n_patterns, "devmerge", g.counters["used_device_merge"])
    if g.groups != ref.groups:
        gid_of = {}
        for gi, grp in enumerate(g.groups):
            for t in grp: gid_of[t] = gi + 1
        n = 0
        for gi, grp in enumerate(ref.groups):
            for t in grp:
                if gid_of.get(t) != gi + 1 and n < 12:
                    print("  token", t, ref.tokens[t - 2], "ref gid", gi + 1, "gpu gid", gid_of.get(t)); n += 1
