import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import crass_amd as ca
ca.load()
n = 100_000_000
spec = ca.synth_spec(read_len=150)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, 150)
eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
m = eng.merge_view()
sz = sorted((len(g) for g in m.groups), reverse=True)
print("groups", len(sz), "top", sz[:12], "sum sq", sum(s*s for s in sz), "tokens", m.n_tokens)
from collections import Counter
print("len hist", sorted(Counter(len(t) for t in m.tokens).items()))
