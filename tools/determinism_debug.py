#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = 150
words = ca.synth_packed(ca.synth_spec(read_len=L), 0, n)
eng = ca.SearchEngine()
eng.load_packed_uniform(words, n, L)
runs = []
for it in range(4):
    c = eng.seed_scan(); m = eng.merge(); r = eng.recruit(); m = eng.merge_view(); cnt = eng.counters()
    runs.append((c, m, r))
    print("run", it, "pass1", c.n, "pass2", r.n, "tokens", m.n_tokens, "groups", m.n_groups, "patterns", m.n_patterns, "devmerge", cnt["used_device_merge"],
          "keys", cnt["anchor_keys"], "kind", cnt["anchor_table_kind"], flush=True)
c0, m0, r0 = runs[0]
for it in range(1, 4):
    c, m, r = runs[it]
    print("run", it, "vs 0: cand", np.array_equal(c.read_idx, c0.read_idx), "tokens", m.tokens == m0.tokens, "groups", m.groups == m0.groups,
          "patterns", sorted(m.patterns) == sorted(m0.patterns), len(set(m.patterns) ^ set(m0.patterns)),
          "rec reads", np.array_equal(r.read_idx, r0.read_idx), "start", int((r.start != r0.start).sum()) if r.n == r0.n else -1,
          "token", int((r.token != r0.token).sum()) if r.n == r0.n else -1)
    if r.n == r0.n:
        bad = np.nonzero(r.start != r0.start)[0][:5]
        for k in bad:
            i = int(r.read_idx[k])
            seq = ca.unpack_ascii(words[i * 10:(i + 1) * 10], 10, 150, 1).tobytes()
            from tests import orc
            ps = orc.PatternSet(m0.patterns)
            print("   read", i, "run0", int(r0.start[k]), int(r0.end[k]), "low", int(r0.low_lexi[k]), "run", it, int(r.start[k]), int(r.end[k]), "low", int(r.low_lexi[k]),
                  "oracle first match (end_excl, len):", ps.first(seq), "token", int(r.token[k]), m0.tokens[int(r.token[k]) - 2])
            print("      ", seq)
eng.close()
