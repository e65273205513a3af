import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import crass_amd as ca
ca.load()
n, L = int(sys.argv[1]), 150
eng = ca.SearchEngine()
eng.load_packed_uniform(ca.synth_packed(ca.synth_spec(read_len=L), 0, n), n, L)
c = eng.seed_scan(); m = eng.merge(); r = eng.recruit(); m = eng.merge_view()
print("reads", n, "found", c.n, "tokens", m.n_tokens, "groups", m.n_groups, "patterns", m.n_patterns, "recruits", r.n, "anchor_keys", eng.counters()["anchor_keys"])
