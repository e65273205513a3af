"""The DR merge on the device (dmerge.hip) against the host merge (merge.cpp) and the oracle:
same tokens, groups, pattern list (including its order) and pass-2 records."""
import os

import numpy as np
import pytest

from tests import orc, fastx
from tests.parity import assert_same_pipeline
from tests.test_gpu_parity import synth_reads, DATA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    crass_amd.load()
    return crass_amd


def both_paths(ca, seqs, hdrs=None):
    os.environ.pop("CRASS_HOST_MERGE", None)
    dev = ca.search_pipeline(seqs, hdrs)
    os.environ["CRASS_HOST_MERGE"] = "1"
    try:
        host = ca.search_pipeline(seqs, hdrs)
    finally:
        os.environ.pop("CRASS_HOST_MERGE", None)
    return dev, host


def assert_same_paths(dev, host, expect_device=True):
    if expect_device:
        assert dev.counters["used_device_merge"] == 1
    assert host.counters["used_device_merge"] == 0
    assert dev.tokens == host.tokens
    assert dev.groups == host.groups
    assert dev.patterns == host.patterns            # the same ORDER, not just the same set
    assert list(dev.pat_group) == list(host.pat_group)
    np.testing.assert_array_equal(dev.rec_read, host.rec_read)
    np.testing.assert_array_equal(dev.rec_token, host.rec_token)
    np.testing.assert_array_equal(dev.rec_lowlexi, host.rec_lowlexi)
    assert dev.n_pass2 == host.n_pass2


@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "CN_gDC.fa.gz", "front_offset_bug.fa.gz", "Ill.nr.miss.fa.gz", "poor_dr_ext.fa.gz"])
def test_reference_inputs_device_vs_host(ca, fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs = [r[2] for r in recs]
    hdrs = [r[0] for r in recs]
    dev, host = both_paths(ca, seqs, hdrs)
    assert_same_paths(dev, host, expect_device=False)      # inputs with N reads keep the host merge
    assert_same_pipeline(dev, orc.pipeline(seqs, hdrs))


@pytest.mark.parametrize("L,n_dr,n", [(150, 50, 200000), (101, 5, 60000), (250, 200, 100000)])
def test_synthetic_device_vs_host(ca, L, n_dr, n):
    seqs = synth_reads(ca, n, read_len=L, n_dr=n_dr, crispr_per_million=30000)
    dev, host = both_paths(ca, seqs)
    assert_same_paths(dev, host)
    assert dev.n_groups >= min(n_dr, 5)
    assert_same_pipeline(dev, orc.pipeline(seqs))


def test_host_view_of_a_large_merge_matches_the_host_merge(ca):
    """>= 8192 distinct DR strings: the host view of a device merge is then built by token-range chunks on the host pool
    (merge.cpp, merge_from_device_finish); tokens, groups, pattern order and pass-2 records against the host merge"""
    seqs = synth_reads(ca, 1_500_000, read_len=150, n_dr=50, crispr_per_million=250000)
    dev, host = both_paths(ca, seqs)
    assert len(dev.tokens) >= 8192
    assert_same_paths(dev, host)
    assert dev.n_groups >= 50


def test_group_size_cap_falls_back(ca):
    """a group beyond the device merge's size cap (its per-group steps are quadratic) must end in the host merge"""
    seqs = synth_reads(ca, 40000, read_len=150, n_dr=2, crispr_per_million=300000)
    ref = orc.pipeline(seqs)
    os.environ["CRASS_DM_GROUP_CAP"] = "20"
    try:
        got = ca.search_pipeline(seqs)
    finally:
        os.environ.pop("CRASS_DM_GROUP_CAP", None)
    assert max(len(g) for g in got.groups) > 20
    assert_same_pipeline(got, ref)


def test_many_variants_one_group(ca):
    """one DR, a CRISPR read in every second read: hundreds of variants in a single group (the
    removeRedundantRepeats stress) and long owner chains in the greedy pass"""
    seqs = synth_reads(ca, 60000, read_len=150, n_dr=1, crispr_per_million=500000)
    dev, host = both_paths(ca, seqs)
    assert_same_paths(dev, host)
    assert_same_pipeline(dev, orc.pipeline(seqs))


def test_non_default_cluster_size(ca):
    seqs = synth_reads(ca, 80000, read_len=150, n_dr=20, crispr_per_million=40000)
    for k in (1, 2, 12):
        p = ca.default_params()
        p.kmer_clust_size = k
        os.environ.pop("CRASS_HOST_MERGE", None)
        dev = ca.search_pipeline(seqs, params=p)
        assert dev.counters["used_device_merge"] == 1
        ref = orc.pipeline(seqs, params=orc.Params(p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize,
                                                   p.searchWindowLength, p.minNumRepeats, p.kmer_clust_size))
        assert_same_pipeline(dev, ref)


def test_fallback_to_host_merge(ca):
    """a device-side failure flag (key set too large, cuckoo insertion giving up, ...) must end in the
    host merge with identical results"""
    seqs = synth_reads(ca, 50000, read_len=150, n_dr=10, crispr_per_million=30000)
    ref = orc.pipeline(seqs)
    os.environ["CRASS_DM_INJECT_FAIL"] = "1"
    try:
        got = ca.search_pipeline(seqs)
        eng = ca.SearchEngine()
        try:            # the bench's call order: nothing fetched between the stages
            packed = ca.PackedReads(seqs)
            eng.load_reads(packed, None)
            eng.seed_scan(fetch=False)
            eng.merge(fetch=False)
            rec = eng.recruit()
            assert len(rec.read_idx) == ref.n_pass2
            cnt = eng.counters()                       # the fallback is visible, with the device's reason
            assert cnt["n_merge_fallbacks"] >= 1 and cnt["last_fallback_bits"] != 0 and cnt["used_device_merge"] == 0
            packed.close()
        finally:
            eng.close()
    finally:
        os.environ.pop("CRASS_DM_INJECT_FAIL", None)
    assert_same_pipeline(got, ref)


def test_speculative_survivor_bound_overflow(ca):
    """the second seed scan of a context launches the survivor stage with a bound learnt from the first
    (no host round trip); a much larger survivor set must be detected and redone exactly"""
    small = synth_reads(ca, 20000, read_len=150, n_dr=5, crispr_per_million=10000)
    big = synth_reads(ca, 200000, read_len=150, n_dr=20, crispr_per_million=600000)
    eng = ca.SearchEngine()
    try:
        a = ca.search_pipeline(small, engine=eng)
        b = ca.search_pipeline(big, engine=eng)            # > 65536 survivors: over the bound
        assert b.counters["n_filter_survivors"] > 65536
        c = ca.search_pipeline(big, engine=eng)            # bound now fits: speculative path
        d = ca.search_pipeline(small, engine=eng)
    finally:
        eng.close()
    assert_same_pipeline(a, orc.pipeline(small))
    ref = orc.pipeline(big)
    assert_same_pipeline(b, ref)
    assert_same_pipeline(c, ref)
    assert_same_pipeline(d, orc.pipeline(small))


@pytest.mark.parametrize("which,bounds", [(0, "64,0,0,0"), (1, "0,16,0,0"), (2, "0,0,16,0")])
def test_every_first_call_bound_overflows_cleanly(ca, which, bounds):
    """crass_hip_load_reads sets the three speculation bounds of a context's FIRST call from the read set; each one too
    small must be detected on the device, the stage repeated with the exact count, and the result unchanged
    (CRASS_TEST_BOUNDS forces tiny bounds; crass_counters.n_bound_overflows says which repeat happened)"""
    seqs = synth_reads(ca, 60000, read_len=150, n_dr=40, crispr_per_million=60000)
    ref = orc.pipeline(seqs)
    os.environ["CRASS_TEST_BOUNDS"] = bounds
    try:
        eng = ca.SearchEngine()
        try:
            a = ca.search_pipeline(seqs, engine=eng)
            b = ca.search_pipeline(seqs, engine=eng)        # (the load sets the tiny bounds again)
        finally:
            eng.close()
    finally:
        os.environ.pop("CRASS_TEST_BOUNDS", None)
    assert_same_pipeline(a, ref)
    assert_same_pipeline(b, ref)
    assert b.counters["n_bound_overflows"][which] >= 2 and b.counters["n_merge_fallbacks"] == 0
    # and without the hook no bound overflows on this input: the first call is speculative end to end
    c = ca.search_pipeline(seqs)
    assert c.counters["n_bound_overflows"] == [0, 0, 0, 0] and c.counters["used_device_merge"] == 1
    assert_same_pipeline(c, ref)


def test_dedup_table_sized_for_the_distinct_bound_overflows_cleanly(ca):
    """Pass 1's de-duplication table is sized from the bound on the DISTINCT strings (not from the survivor slots): with a
    bound of 16 (a 64-slot table under the test hook) and hundreds of distinct repeats every insert runs out of probes; the
    host must de-duplicate that call itself, later calls of the context use the full-size table, results unchanged."""
    seqs = synth_reads(ca, 80000, read_len=150, n_dr=300, crispr_per_million=80000)
    ref = orc.pipeline(seqs)
    assert len(ref.tokens) > 200
    os.environ["CRASS_TEST_BOUNDS"] = "0,16,0,0"
    try:
        eng = ca.SearchEngine()
        try:
            a = ca.search_pipeline(seqs, engine=eng)
            b = ca.search_pipeline(seqs, engine=eng)
        finally:
            eng.close()
    finally:
        os.environ.pop("CRASS_TEST_BOUNDS", None)
    assert a.counters["n_bound_overflows"][1] >= 1
    assert_same_pipeline(a, ref)
    assert_same_pipeline(b, ref)
    os.environ["CRASS_DD_FULL_TABLE"] = "1"                 # the A/B switch: the table of round 3
    try:
        c = ca.search_pipeline(seqs)
    finally:
        os.environ.pop("CRASS_DD_FULL_TABLE", None)
    assert c.counters["n_bound_overflows"] == [0, 0, 0, 0]
    assert_same_pipeline(c, ref)


def test_exception_reads_stay_on_the_device_path(ca):
    """Reads with non-ACGT bytes: screened on their packed words (a non-ACGT byte packs as 'A': still a superset),
    evaluated byte-wise in pass 1, and — the pattern set being pure ACGT — filtered and verified on the device in
    pass 2 with a byte check per candidate (the automaton of libcrispr.cpp:503 cannot match across another byte).
    N, IUPAC codes and lower-case bases are injected everywhere; reads whose DR would then contain such a byte
    are restored first, so that the device merge applies."""
    import random
    rng = random.Random(11)
    base = synth_reads(ca, 80000, read_len=150, crispr_per_million=30000)
    seqs = [bytearray(s) for s in base]
    for i in range(len(seqs)):
        r = rng.random()
        k = 1 if r < 0.04 else 4 if r < 0.06 else 30 if r < 0.062 else 0
        for _ in range(k):
            seqs[i][rng.randrange(150)] = ord(rng.choice("NNNNRYacgtn"))
    ok = set(b"ACGT")
    for _ in range(6):
        cur = [bytes(s) for s in seqs]
        ref = orc.pipeline(cur)
        bad_tok = {t + 2 for t, s in enumerate(ref.tokens) if not set(s) <= ok}      # token ids start at 2
        bad_reads = [int(r) for r, t in zip(ref.rec_read[:ref.n_pass1], ref.rec_token[:ref.n_pass1]) if int(t) in bad_tok]
        if not bad_reads:
            break
        for r in bad_reads:
            seqs[r] = bytearray(base[r])
    assert not bad_reads
    dev, host = both_paths(ca, cur)
    assert dev.counters["n_exceptions"] > 3000
    assert_same_paths(dev, host)
    assert_same_pipeline(dev, ref)
    exc = np.array([not set(s) <= ok for s in cur])
    n1 = ref.n_pass1
    assert exc[ref.rec_read[:n1]].sum() > 20          # exception reads found by pass 1 ...
    assert exc[ref.rec_read[n1:n1 + ref.n_pass2]].sum() > 20   # ... and recruited by pass 2
    # 'N' inside the repeats themselves: DR variants with an N.  The device merge carries an N-position mask
    # through clustering (11-mers with an N get their identity from an all-pairs pass), removeRedundantRepeats,
    # the pattern list and the pass-2 verification (byte for byte, as the automaton of libcrispr.cpp:503).
    L = 150
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    cur2 = list(cur)
    picked = 0
    for k in range(n1):
        r = int(ref.rec_read[k])
        if exc[r] or picked >= 300:
            continue
        ss = ref.ss(k)
        replen = int(ref.rec_replen[k])
        # reads with the same variant get their N at the same offset: shared N-containing 11-mers across tokens
        d = sum(ref.tokens[int(ref.rec_token[k]) - 2]) % replen if k % 3 else rng.randrange(replen)
        s = bytearray(cur2[r])
        for q in range(0, len(ss), 2):
            p = ss[q] + d
            p = p if ref.rec_lowlexi[k] else L - 1 - p
            if 0 <= p < L:
                s[p] = ord("N")
        cur2[r] = bytes(s)
        picked += 1
    ref2 = orc.pipeline(cur2)
    mixed = [t for t in ref2.tokens if not set(t) <= ok]
    assert len(mixed) >= 10 and all(set(t) <= set(b"ACGTN") for t in mixed)
    # single-copy reads that only pass 2 can recruit: a variant with its N, its reverse complement, and two
    # decoys (N replaced by A / by R) that must not match the N pattern
    filler = lambda n: bytes(rng.choice(b"ACGT") for _ in range(n))
    for t in mixed[:25]:
        for body in (t, t.translate(comp)[::-1], t.replace(b"N", b"A"), t.replace(b"N", b"R")):
            pre = rng.randrange(5, 60)
            cur2.append(filler(pre) + body + filler(L - pre - len(body)))
    ref2 = orc.pipeline(cur2)
    assert any(not set(t) <= ok for t in ref2.patterns)
    n12 = ref2.n_pass1
    assert (ref2.rec_read[n12:n12 + ref2.n_pass2] >= len(cur)).sum() >= 20
    dev2, host2 = both_paths(ca, cur2)
    assert_same_paths(dev2, host2)
    assert_same_pipeline(dev2, ref2)
    # any other byte in a DR (IUPAC code, lower case): the device merge declines, the host merges
    cur3 = list(cur2)
    for k0 in range(20):
        r = int(ref2.rec_read[k0])
        ss = ref2.ss(k0)
        s = bytearray(cur3[r])
        for q in range(0, len(ss), 2):
            p = ss[q] + 3
            p = p if ref2.rec_lowlexi[k0] else L - 1 - p
            if 0 <= p < L:
                s[p] = ord("R")
        cur3[r] = bytes(s)
    ref3 = orc.pipeline(cur3)
    assert any(b"R" in t or b"Y" in t for t in ref3.tokens)
    dev3, host3 = both_paths(ca, cur3)
    assert dev3.counters["used_device_merge"] == 0
    assert_same_paths(dev3, host3, expect_device=False)
    assert_same_pipeline(dev3, ref3)


def test_queued_merge_call_orders(ca):
    """From its second call on, a context's seed scan queues the device merge itself (token count on the device);
    crass_hip_merge adopts it.  Unusual call orders must give the same results: a repeated seed scan, a merge from
    caller-supplied strings while a queued merge is in flight, a much larger and a much smaller read set on the same
    context (bounds learnt from the previous call do not fit)."""
    small = synth_reads(ca, 30000, read_len=150, n_dr=5, crispr_per_million=20000)
    big = synth_reads(ca, 200000, read_len=150, n_dr=200, crispr_per_million=40000, first=50000)
    ref_small, ref_big = orc.pipeline(small), orc.pipeline(big)
    eng = ca.SearchEngine()
    try:
        def run(seqs, twice=False, explicit=False):
            packed = ca.PackedReads(seqs)
            eng.load_reads(packed, None)
            cands = eng.seed_scan()
            if twice:
                cands = eng.seed_scan()
            if explicit:
                chars, lens = eng.candidate_dr_view()
                m = eng.merge(np.array(chars), np.array(lens))
            else:
                m = eng.merge()
            rec = eng.recruit()
            used = eng.counters()["used_device_merge"]
            packed.close()
            return cands, m, rec, used

        for seqs, ref, kw, want_dev in ((small, ref_small, {}, 1), (small, ref_small, {}, 1), (big, ref_big, {}, 1),
                                        (big, ref_big, dict(twice=True), 1), (small, ref_small, {}, 1),
                                        (big, ref_big, dict(explicit=True), 0), (big, ref_big, {}, 1), (small, ref_small, {}, 1)):
            cands, m, rec, used = run(seqs, **kw)
            assert used == want_dev
            assert m.tokens == ref.tokens and m.groups == ref.groups and sorted(m.patterns) == sorted(ref.patterns)
            n1 = ref.n_pass1
            assert cands.read_idx.tolist() == ref.rec_read[:n1].tolist()
            assert m.cand_token.tolist() == ref.rec_token[:n1].tolist()
            assert rec.read_idx.tolist() == ref.rec_read[n1:n1 + ref.n_pass2].tolist()
            assert rec.token.tolist() == ref.rec_token[n1:n1 + ref.n_pass2].tolist()
    finally:
        eng.close()


def test_out_of_memory_is_reported_and_the_context_survives(ca):
    """fault path: a device allocation that fails (forced with CRASS_POOL_CAP_MB, a cap on this library's device
    allocations) must come back as CRASS_ERR_OOM from whichever call hit it, and the context must stay usable"""
    big = synth_reads(ca, 1500000, read_len=150)                 # 60 MB of packed words + ~30 MB of per-read scratch
    small = synth_reads(ca, 40000, read_len=150, crispr_per_million=30000)
    ref = orc.pipeline(small)
    os.environ["CRASS_POOL_CAP_MB"] = "48"
    try:
        eng = ca.SearchEngine()
        pk_big, pk_small = ca.PackedReads(big), ca.PackedReads(small)
        with pytest.raises(ca.CrassError) as ei:
            eng.load_reads(pk_big, None)
        assert ei.value.status == 5 and "memory" in str(ei.value)       # CRASS_ERR_OOM
        with pytest.raises(ca.CrassError) as ei:                          # nothing half-loaded is left behind
            eng.seed_scan()
        assert ei.value.status == 6                                      # CRASS_ERR_STATE
        got = ca.search_pipeline(small, engine=eng)                      # the same context, a batch that fits
        assert_same_pipeline(got, ref)
        # the cap can also bite later, inside the search: a fresh context, load under a roomy cap, lower it below what pass 1
        # needs.  (With the pools sized at load — the default — pass 1 allocates nothing any more: the in-step allocation failure
        # is reached with the load-time sizing switched off, which is also what an overflowing bound falls back to.)
        os.environ["CRASS_POOL_CAP_MB"] = "400"
        os.environ["CRASS_NO_PRESIZE"] = "1"
        eng2 = ca.SearchEngine()
        eng2.load_reads(pk_big, None)
        os.environ["CRASS_POOL_CAP_MB"] = "1"
        eng2.reload_env()
        with pytest.raises(ca.CrassError) as ei:
            eng2.seed_scan()
        assert ei.value.status == 5
        os.environ.pop("CRASS_POOL_CAP_MB")
        os.environ.pop("CRASS_NO_PRESIZE")
        eng2.reload_env()
        c = eng2.seed_scan()                                             # ... and the retry without the cap succeeds
        assert c.n > 0
        eng2.close()
        # the first context again, after all of this
        assert_same_pipeline(ca.search_pipeline(small, engine=eng), ref)
        eng.close(); pk_big.close(); pk_small.close()
    finally:
        os.environ.pop("CRASS_POOL_CAP_MB", None)
        os.environ.pop("CRASS_NO_PRESIZE", None)


def test_n_kmers_decide_group_membership(ca):
    """DR variants with an N whose CLEAN 11-mers alone do not reach kmer_clust_size: they join their group only if the
    11-mers that contain the N get the reference's identity (laurenized string over A<C<G<N<T, N complements to N) —
    parity_sweep seed 11 case 74 caught a wrong complement table here"""
    import random
    rng = random.Random(74)
    seqs = [bytearray(s) for s in synth_reads(ca, 60000, read_len=250, n_dr=1, crispr_per_million=300000)]
    for i in rng.sample(range(len(seqs)), 6000):
        seqs[i][rng.randrange(250)] = ord("N")
    seqs = [bytes(s) for s in seqs]
    p = ca.default_params(kmer_clust_size=12)
    got = ca.search_pipeline(seqs, params=p)
    ref = orc.pipeline(seqs, params=orc.Params(p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize, p.searchWindowLength,
                                               p.minNumRepeats, p.kmer_clust_size))
    assert got.counters["used_device_merge"] == 1
    assert sum(1 for t in ref.tokens if b"N" in t) >= 20
    assert_same_pipeline(got, ref)


_SLOT_SCRATCH = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import crass_amd as ca
from tests import orc
from tests.parity import assert_same_pipeline
from tests.test_gpu_parity import synth_reads
ca.load()
# 70 000 reads, most of them with a lattice seed hit: the survivor bound (1.5 x, rounded up to 65 536) is 131 072 slots, i.e.
# 2 048 mask words where the scratch sized from the read count has 1 095
seqs = synth_reads(ca, 70000, read_len=150, n_dr=30, crispr_per_million=900000)
for rep in range(2):                                    # the second call speculates with the bound learnt from the first
    gpu = ca.search_pipeline(seqs)
    assert gpu.counters["n_filter_survivors"] > 46667, gpu.counters
    assert_same_pipeline(gpu, orc.pipeline(seqs))
with ca.SearchEngine() as eng:                          # ... and on one context, step after step
    eng.load_reads(ca.PackedReads(seqs))
    for _ in range(3):
        eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    assert eng.counters()["n_pass1_found"] == gpu.n_pass1
print("slot scratch ok")
'''


@pytest.mark.parametrize("env", [dict(CRASS_NO_LOOKBACK="1", CRASS_GUARD_PAGES="1"), dict(CRASS_NO_LOOKBACK="1")])
def test_slot_compactions_fit_their_scratch(env):
    """the three-kernel compactions (no look-back) over survivor / hit SLOTS write (bound + 63) / 64 mask words; the
    scratch was sized from the READ count only.  In a child process, with every buffer between unmapped pages."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SLOT_SCRATCH], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "slot scratch ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("switch", ["CRASS_DV_SINGLE", "CRASS_P2_BLOB12", "CRASS_DD_FULL_TABLE", "CRASS_SURV_NO_REGROUP",
                                    "CRASS_DX_PINNED", "CRASS_NO_LOOKBACK", "CRASS_FF_RPL=4",
                                    # round 6: events instead of the polled stage flags; the ranks' seed scans waiting for pass 1; the found
                                    # flags cleared by a fill kernel; no light walk over the dense path's survivors; the view in one copy
                                    "CRASS_NO_POLL", "CRASS_NO_DEFER_P1", "CRASS_FOUND_MEMSET", "CRASS_NO_DENSE_LIGHT", "CRASS_NO_WARM_LAUNCH",
                                    "CRASS_STAGE_TIMING=2"])
def test_ab_switches_keep_the_results(switch):
    """The A/B switches select the round-3 form of something (one candidate per lane in pass 2's verification, the 12-byte
    hand-off record, the full-size de-duplication table, survivors in slot order, the distinct list written to pinned memory, the
    three-kernel compactions) or, for CRASS_FF_RPL=4, the big-set form of the seed scan — four rows per lane — on small ragged
    sets.  Most are read once per process, so each gets a fresh one: a short randomised sweep
    (tools/parity_sweep.py: HIP path == oracle, field by field) with the switch set."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    name, _, value = switch.partition("=")
    env[name] = value or "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "parity_sweep.py"), "24", "77"], env=env, capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    assert r.returncode == 0 and "24 cases, 0 failures" in tail, tail
