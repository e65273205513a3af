"""GPU parity tests proper: the HIP path (through the C ABI) vs the oracle on the same
inputs — bit-exact records, tokens, groups and pattern sets."""
import os
import random

import numpy as np
import pytest

from tests import orc, fastx
from tests.parity import assert_same_pipeline

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(__file__), "golden", "data")
KNOWN = {
    "Ill100.fx.gz": (4324, 837, 140, 1, 42, 4312),
    "CN_gDC.fa.gz": (4740, 2761, 92, 1, 36, 4740),
    "front_offset_bug.fa.gz": (618, 54, 50, 2, 58, 589),
    "Ill.nr.miss.fa.gz": (395, 10, 9, 1, 12, 344),
    "poor_dr_ext.fa.gz": (8, 6, 4, 1, 4, 8),
}


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    crass_amd.load()
    return crass_amd


def to_orc_params(p):
    return orc.Params(p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize, p.searchWindowLength,
                      p.minNumRepeats, p.kmer_clust_size)


@pytest.mark.parametrize("fname", sorted(KNOWN))
def test_reference_test_inputs(ca, fname):
    """config 1 and friends: the reference's own regression inputs, default options."""
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs = [r[2] for r in recs]
    hdrs = [r[0] for r in recs]
    gpu = ca.search_pipeline(seqs, hdrs)
    ref = orc.pipeline(seqs, hdrs)
    assert_same_pipeline(gpu, ref)
    got = (len(recs), gpu.n_pass1, len(set(gpu.rec_token[:gpu.n_pass1].tolist())), gpu.n_groups, gpu.n_patterns,
           gpu.n_pass1 + gpu.n_pass2)
    assert got == KNOWN[fname]          # the compiled reference's known answers


def synth_reads(ca, n, first=0, **kw):
    spec = ca.synth_spec(**kw)
    w = ca.synth_packed(spec, first, n)
    L = spec.read_len
    asc = ca.unpack_ascii(w, (L + 15) // 16, L, n)
    return [asc[i * L:(i + 1) * L].tobytes() for i in range(n)]


@pytest.mark.parametrize("L", [150, 101, 250, 57, 58, 64, 75])
def test_synthetic_uniform_lengths(ca, L):
    seqs = synth_reads(ca, 40000, read_len=L, crispr_per_million=30000)
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    if L >= 101:
        assert gpu.n_pass1 > 0 and gpu.n_pass2 > 0
        assert gpu.counters["used_fast_filter"] == 1


def test_synthetic_ragged_with_exceptions_and_duplicate_headers(ca):
    rng = random.Random(7)
    base = synth_reads(ca, 30000, read_len=150, crispr_per_million=50000)
    seqs, hdrs = [], []
    for i, s in enumerate(base):
        s = bytearray(s[:rng.randint(40, 150)]) if rng.random() < 0.5 else bytearray(s)
        r = rng.random()
        if r < 0.03:
            s[rng.randrange(len(s))] = ord("N")
        elif r < 0.04:
            for _ in range(5):
                s[rng.randrange(len(s))] = ord("N")
        elif r < 0.045:
            s[rng.randrange(len(s))] = ord("a")
        seqs.append(bytes(s))
        # duplicate headers: later reads re-use an earlier header (readsFound is keyed by header)
        hdrs.append(b"r%d" % (i if rng.random() > 0.05 else rng.randrange(0, i + 1)))
    gpu = ca.search_pipeline(seqs, hdrs)
    ref = orc.pipeline(seqs, hdrs)
    assert_same_pipeline(gpu, ref)
    assert gpu.counters["n_exceptions"] > 500
    # the same reads padded to one stride (what the adapter does for trimmed short reads): bit-parallel filter and
    # lane-per-read kernels with per-read lengths
    padded = ca.search_pipeline(seqs, hdrs, pad_uniform=2)
    assert padded.counters["used_fast_filter"] == 1
    assert_same_pipeline(padded, ref)
    # the wave kernel's two-launch form (short Levenshtein rows first, reads that need longer ones redone with the
    # full layout): rows of 8 entries force the second launch for every fallback comparison
    os.environ["CRASS_ROW_CAP"] = "8"
    try:
        again = ca.search_pipeline(seqs, hdrs)
    finally:
        os.environ.pop("CRASS_ROW_CAP", None)
    assert_same_pipeline(again, ref)


PARAM_SETS = [
    dict(searchWindowLength=6), dict(searchWindowLength=7), dict(searchWindowLength=9),
    dict(minNumRepeats=3), dict(lowDRsize=20, highDRsize=40), dict(lowSpacerSize=20, highSpacerSize=60),
    dict(kmer_clust_size=4), dict(lowDRsize=15, searchWindowLength=8),   # skips == 0 -> 1
    dict(lowDRsize=11, highDRsize=30, searchWindowLength=6, lowSpacerSize=10, highSpacerSize=40),
]


@pytest.mark.parametrize("kw", PARAM_SETS, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_non_default_options(ca, kw):
    seqs = synth_reads(ca, 20000, read_len=150, crispr_per_million=50000)
    # real CRISPR reads too
    seqs += [r[2] for r in fastx.read_fastx(os.path.join(DATA, "front_offset_bug.fa.gz"))]
    p = ca.default_params(**kw)
    gpu = ca.search_pipeline(seqs, params=p)
    ref = orc.pipeline(seqs, params=to_orc_params(p))
    assert_same_pipeline(gpu, ref)


FAST_PARAM_SETS = [
    dict(lowSpacerSize=20, highSpacerSize=60), dict(highDRsize=64, highSpacerSize=70), dict(lowSpacerSize=30, highSpacerSize=34),   # shift range only
    dict(searchWindowLength=6), dict(searchWindowLength=7), dict(searchWindowLength=9),                                            # another window
    dict(lowDRsize=20, highDRsize=40), dict(lowDRsize=30, highDRsize=60), dict(lowDRsize=15, searchWindowLength=8),                # another lattice (skips 5 / 15 / 1)
    dict(lowDRsize=17, highDRsize=35, searchWindowLength=6, lowSpacerSize=15, highSpacerSize=45),
]


@pytest.mark.parametrize("kw", FAST_PARAM_SETS, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
@pytest.mark.parametrize("L", [150, 101, 250])
def test_bit_parallel_filter_with_non_default_options(ca, kw, L):
    """any -w / -d / -D / -s / -S used to send every read through k_filter_general (40-80 x slower than the
    defaults' filter).  With a uniform stride they now take the bit-parallel filter too — its run-time-range form when only the
    distances change, the every-position form for another window or seed lattice — and the seed hints the lane kernel walks by.
    Uniform reads and trimmed ones padded to one stride (per-lane searchEnd), N reads; parity with the oracle under the same options."""
    rng = random.Random(len(str(kw)) + L)
    base = synth_reads(ca, 24000, read_len=L, crispr_per_million=50000)
    p = ca.default_params(**kw)
    gpu = ca.search_pipeline(base, params=p, pad_uniform=2)
    assert gpu.counters["used_fast_filter"] == 1
    assert_same_pipeline(gpu, orc.pipeline(base, params=to_orc_params(p)))
    assert gpu.n_pass1 > 0
    seqs = []
    for i, s in enumerate(base[:12000]):
        s = bytearray(s[:rng.randint(max(40, L - 40), L)]) if rng.random() < 0.4 else bytearray(s)
        if rng.random() < 0.02:
            s[rng.randrange(len(s))] = ord("N")
        seqs.append(bytes(s))
    padded = ca.search_pipeline(seqs, params=p, pad_uniform=2)
    assert padded.counters["used_fast_filter"] == 1
    assert_same_pipeline(padded, orc.pipeline(seqs, params=to_orc_params(p)))


@pytest.mark.parametrize("low", [19, 20, 22])
@pytest.mark.parametrize("L", [150, 101, 300])
def test_short_repeats_take_the_device_merge_and_the_anchor_probe(ca, low, L):
    """-d 19 .. 22 (crass accepts -d >= 8, crass.cpp:264-271): DR strings below 23 bases.  Until round 6 they took the host merge
    and the byte-wise automaton kernels (pass 2 at 77 x the defaults' cost); now the device merge takes strings from 19 bases and
    pass 2 probes 16-base windows every 4 bases (every pattern of >= 19 bases contains one), verified exactly.  Reads whose
    repeats ARE that short, beside longer ones; uniform, ragged and N reads; groups of contexts."""
    rng = random.Random(low * 1000 + L)
    seqs = synth_reads(ca, 30000, read_len=L, n_dr=40, dr_len_min=low, dr_len_max=low + 6, spacer_len_min=26, spacer_len_max=34, crispr_per_million=60000)
    seqs += synth_reads(ca, 10000, read_len=L, n_dr=20, crispr_per_million=40000)
    p = ca.default_params(lowDRsize=low, highDRsize=low + 22)
    gpu = ca.search_pipeline(seqs, params=p)
    ref = orc.pipeline(seqs, params=to_orc_params(p))
    assert_same_pipeline(gpu, ref)
    assert gpu.counters["used_device_merge"] == 1 and gpu.counters["used_lds_automaton"] == 2      # (2: anchor probe + exact verification)
    assert min(len(t) for t in gpu.tokens) < 23 and gpu.n_pass2 > 0
    rag = []
    for q in seqs[:16000]:
        q = bytearray(q[:rng.randint(max(60, L - 50), L)]) if rng.random() < 0.5 else bytearray(q)
        if rng.random() < 0.02:
            q[rng.randrange(len(q))] = ord("N")
        rag.append(bytes(q))
    gpu = ca.search_pipeline(rag, params=p)
    assert_same_pipeline(gpu, orc.pipeline(rag, params=to_orc_params(p)))
    assert gpu.counters["used_device_merge"] == 1
    grp = ca.search_pipeline_group(seqs[:24000], [0, 0], params=p, local_copies=True)
    assert_same_pipeline(grp, orc.pipeline(seqs[:24000], params=to_orc_params(p)))


def test_unsigned_skips_wrap_is_refused(ca):
    """-d < 2w-1 wraps the reference's unsigned `skips` (libcrispr.cpp:281-285); the reference's
    seed loop is then ill-defined (can walk backwards forever).  The engine refuses it."""
    with pytest.raises(ca.CrassError) as e:
        ca.SearchEngine(ca.default_params(lowDRsize=8, highDRsize=60, searchWindowLength=9))
    assert e.value.status == 2


def test_long_reads(ca):
    """config 4 shape (10 kbp reads with 20-60 repeat arrays), small count."""
    rng = random.Random(11)

    def rs(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    drs = [rs(rng.randint(28, 37)) for _ in range(5)]
    seqs = []
    for i in range(60):
        L = 10000 if i % 3 else rng.randint(3000, 12000)
        s = bytearray(rs(L))
        if i % 2 == 0:
            dr = rng.choice(drs)
            arr = b"".join(dr + rs(rng.randint(30, 38)) for _ in range(rng.randint(20, 60)))
            at = rng.randint(0, max(0, L - len(arr) - 1))
            s[at:at + len(arr)] = arr
            s = s[:L]
        if i == 7:
            s[100] = ord("N")
        seqs.append(bytes(s))
    seqs += [r[2] for r in fastx.read_fastx(os.path.join(DATA, "poor_dr_ext.fa.gz"))]
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    assert gpu.n_pass1 >= 20
    assert max(gpu.rec_nss[:gpu.n_pass1]) > 20


def test_edge_inputs(ca):
    seqs = [b"ACGT" * 10, b"A" * 150, b"ACGTACGTAC" * 15, b"N" * 80, b"ACGT" * 14 + b"AC"]   # too short / low complexity
    pal = b"GTTTTAGAGCTATGCTGTTTTGAATGGTCCCAAAAC"
    # palindromic DR (its own reverse complement) — flips the read (ReadHolder.cpp:573-590)
    pdr = b"ACGTTGCATGCAACGTACGTTGCATGCAACGT"
    rng = random.Random(3)

    def rs(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    for dr in (pal, pdr):
        for _ in range(20):
            s = rs(rng.randint(0, 20))
            while len(s) < 200:
                s += dr + rs(rng.randint(28, 40))
            seqs.append(s[:rng.randint(120, 200)])
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    # empty input
    e = ca.search_pipeline([])
    assert e.n_pass1 == 0 and e.n_pass2 == 0 and e.n_patterns == 0


def test_levenshtein_batch(ca):
    rng = random.Random(5)
    pairs = [(b"ABCDEF", b"ABDCEF"), (b"ABCDEF", b"BACDEF"), (b"", b"ACGT"), (b"AC", b"ACGT"), (b"ACG", b"ACG"),
             (b"A" * 50, b"A" * 41 + b"C" * 9)]
    for _ in range(400):
        n = rng.choice([rng.randint(0, 60), rng.randint(60, 70), rng.randint(100, 300)])
        m = rng.choice([rng.randint(0, 60), rng.randint(60, 70), rng.randint(100, 300)])
        s = bytes(rng.choice(b"ACGT") for _ in range(n))
        if rng.random() < 0.5 and n > 3:
            t = bytearray(s)
            for _k in range(rng.randint(0, 8)):
                i = rng.randrange(len(t) - 1)
                t[i], t[i + 1] = t[i + 1], t[i]
            t = bytes(t)
        else:
            t = bytes(rng.choice(b"ACGT") for _ in range(m))
        pairs.append((s, t))
    # non-ACGT bytes: in the longer string they just never match; in the shorter one the
    # bit-parallel lane kernel must hand the pair to the wavefront kernel
    pairs += [(b"ACGTNACGTACGTTTGACNA", b"ACGTACGTACGTTGACA"), (b"ACGTACGTAC", b"ACNTACGTACGGTTACAGT"),
              (b"acgtacgtacgt", b"ACGTACGTACGT"), (b"A" * 64 + b"C", b"A" * 64), (b"ACGT" * 16, b"ACGT" * 16 + b"TT")]
    with ca.SearchEngine() as eng:
        dist, sim = eng.levenshtein_batch(pairs)
    L = orc.lib()
    for k, (s, t) in enumerate(pairs):
        assert dist[k] == L.orc_levenshtein(s, len(s), t, len(t)), (s, t)
        assert np.float32(sim[k]).tobytes() == np.float32(L.orc_similarity(s, len(s), t, len(t))).tobytes()


def test_filter_is_superset_of_lattice_hits(ca):
    """the device filter may only ADD survivors: every read the oracle says has a lattice
    seed hit must survive (checked through the end result + the survivor counter)."""
    seqs = synth_reads(ca, 50000, read_len=150)
    import ctypes as C
    p = orc.Params.default()
    n_hit = sum(orc.lib().orc_has_lattice_hit(s, len(s), C.byref(p)) for s in seqs)
    gpu = ca.search_pipeline(seqs)
    assert gpu.counters["n_filter_survivors"] >= n_hit
    assert gpu.counters["n_filter_survivors"] < n_hit * 1.5 + 100


def test_config5_shape_many_drs_gc_classes(ca):
    """config 5 shape at test size: 500 seeded DRs, 4 GC-content classes (high-diversity background)."""
    seqs = synth_reads(ca, 150000, read_len=150, n_dr=500, gc_classes=4, crispr_per_million=40000)
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    assert gpu.n_groups > 300 and gpu.n_patterns > 1000
    # pattern sets too large for the LDS anchor table keep the anchor path (table probed in L2)
    assert gpu.counters["used_lds_automaton"] == 2


@pytest.mark.parametrize("n_dr,kind,host_merge", [(2000, 1, True), (9000, 2, True), (1200, 1, False), (2000, (1, 2), False), (9000, 2, False)])
def test_anchor_table_tiers(ca, n_dr, kind, host_merge):
    """pattern sets beyond the exact LDS anchor table: fingerprint buckets in LDS (kind 1, host-built tables
    only), then exact keys probed in L2 (kind 2) — same records either way, whichever side built the table."""
    seqs = synth_reads(ca, 200000, read_len=150, n_dr=n_dr, crispr_per_million=150000)
    if host_merge:
        os.environ["CRASS_HOST_MERGE"] = "1"
    try:
        gpu = ca.search_pipeline(seqs)
    finally:
        os.environ.pop("CRASS_HOST_MERGE", None)
    assert gpu.counters["used_device_merge"] == (0 if host_merge else 1)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    assert gpu.counters["used_lds_automaton"] == 2
    assert gpu.counters["anchor_table_kind"] in (kind if isinstance(kind, tuple) else (kind,)), gpu.counters


def test_full_size_properties(ca):
    """BASELINE configs[1] size (10 M reads) through size-independent properties: determinism across two
    runs, record order, every recruited read absent from pass 1, tokens within range, and an exact
    oracle comparison on a 300 k-read prefix processed separately."""
    n, L = 10_000_000, 150
    spec = ca.synth_spec(read_len=L)
    words = ca.synth_packed(spec, 0, n)
    with ca.SearchEngine() as eng:
        eng.load_packed_uniform(words, n, L)
        c1 = eng.seed_scan(); m1 = eng.merge(); r1 = eng.recruit()
        c2 = eng.seed_scan(); m2 = eng.merge(); r2 = eng.recruit()
    for a, b in ((c1.read_idx, c2.read_idx), (c1.ss_pool, c2.ss_pool), (c1.dr_chars, c2.dr_chars), (r1.read_idx, r2.read_idx),
                 (r1.start, r2.start), (r1.token, r2.token), (m1.cand_token, m2.cand_token)):
        assert np.array_equal(a, b)
    assert m1.patterns == m2.patterns and m1.groups == m2.groups
    assert np.all(np.diff(c1.read_idx.astype(np.int64)) > 0) and np.all(np.diff(r1.read_idx.astype(np.int64)) > 0)
    assert len(np.intersect1d(c1.read_idx, r1.read_idx)) == 0
    assert r1.token.min() >= 2 and r1.token.max() <= m1.n_tokens + 1
    assert np.all(r1.end - r1.start + 1 == r1.dr_len)
    # every recruit's DR is the token's string
    for k in range(0, r1.n, 997):
        assert r1.dr(k) == m1.tokens[int(r1.token[k]) - 2]
    # prefix vs oracle
    m = 300000
    asc = ca.unpack_ascii(words, 10, L, m)
    seqs = [asc[i * L:(i + 1) * L].tobytes() for i in range(m)]
    assert_same_pipeline(ca.search_pipeline(seqs), orc.pipeline(seqs))
    # pass-1 decisions are per read: the prefix of the big run equals the small run
    k = int(np.searchsorted(c1.read_idx, m))
    small = ca.search_pipeline(seqs, do_pass2=False)
    assert np.array_equal(c1.read_idx[:k], small.rec_read[:small.n_pass1])


def test_long_reads_ragged_mid_size(ca):
    """6 000 reads of 2.1-12 kbp (42 M bases): the hint kernel finds a tile's read from a per-block index when the lengths
    differ, the wave kernel prefetches reads of differing length, the sink fills its arrays in parallel"""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    drs = [acgt[rng.integers(0, 4, size=int(rng.integers(28, 38)))] for _ in range(12)]
    seqs = []
    for i in range(6000):
        L = int(rng.integers(2100, 12001))
        s = acgt[rng.integers(0, 4, size=L)]
        if i % 20 == 0:
            dr = drs[int(rng.integers(0, len(drs)))]
            pos = int(rng.integers(0, max(1, L - 2000)))
            for _ in range(int(rng.integers(3, 40))):
                unit = np.concatenate([dr, acgt[rng.integers(0, 4, size=int(rng.integers(30, 39)))]])
                if pos + len(unit) > L:
                    break
                s[pos:pos + len(unit)] = unit
                pos += len(unit)
        seqs.append(s.tobytes())
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    assert gpu.n_pass1 >= 200
    # (no exception reads, one chunk: the found records are de-duplicated on the device and the DEVICE merge runs — the
    # host-loop sink of long reads no longer means the host merge)
    assert gpu.counters["used_device_merge"] == 1 and gpu.counters["n_merge_fallbacks"] == 0
    # the walking kernel's ASCII window (long reads: room for the region around a candidate's repeats, not for the read): with
    # 1 024 bytes every array beyond ~700 bases does not fit and goes to the second launch with the full layout; with the full
    # layout for everybody (the A/B switch) nothing does
    # ... and without the light walk (k_long_light decides the reads without an array and hands the others to the wave kernel from
    # a list): the wave kernel for every read, the A/B switch
    for env in ({"CRASS_SEQ_WINDOW": "1024"}, {"CRASS_LONG_FULL_LAYOUT": "1"}, {"CRASS_NO_LIGHT": "1"}, {"CRASS_NO_LIGHT": "1", "CRASS_SEQ_WINDOW": "1024"}):
        os.environ.update(env)
        try:
            alt = ca.search_pipeline(seqs)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert_same_pipeline(alt, ref)
    # the hint kernel in slices beside the walk (the default only slices sets of >= 4 096 reads): every slice count, ragged
    # lengths — slice boundaries fall inside reads' hint words and inside the per-block read index
    for parts in ("1", "3", "4"):
        os.environ["CRASS_HINT_PARTS"] = parts
        try:
            sliced = ca.search_pipeline(seqs)
        finally:
            os.environ.pop("CRASS_HINT_PARTS", None)
        assert_same_pipeline(sliced, ref)


def _mid_reads(rng, n, lo, hi, every=15):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    drs = [acgt[rng.integers(0, 4, size=int(rng.integers(24, 40)))] for _ in range(9)]
    seqs = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = acgt[rng.integers(0, 4, size=L)]
        if i % every == 0:
            dr = drs[int(rng.integers(0, len(drs)))]
            pos = int(rng.integers(0, max(1, L // 3)))
            for _ in range(int(rng.integers(2, 30))):
                unit = np.concatenate([dr, acgt[rng.integers(0, 4, size=int(rng.integers(27, 45)))]])
                if pos + len(unit) > L:
                    break
                s[pos:pos + len(unit)] = unit
                pos += len(unit)
        seqs.append(s.tobytes())
    return seqs


@pytest.mark.parametrize("lo,hi", [(257, 257), (300, 300), (301, 301), (500, 500), (960, 960), (1000, 1000), (2048, 2048), (40, 700), (30, 2048), (250, 320)])
def test_mid_length_reads_take_the_hint_filter(ca, lo, hi):
    """reads of 257 .. 2 048 bases (MiSeq 2 x 300, 454, merged pairs) and sets whose strides differ: no lane-per-read filter, the
    position hints are the filter (k_hint_positions flags the reads that have one) and the survivor kernels walk on them; up to 2 048 bases the
    found records stay on the dense hand-off (start/stop lists of up to 128 entries).  CRASS_NO_HINT_FILTER: k_filter_general, the
    same records"""
    rng = np.random.default_rng(lo * 7 + hi)
    seqs = _mid_reads(rng, 12000 if hi <= 1000 else 5000, lo, hi)
    gpu = ca.search_pipeline(seqs)
    ref = orc.pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    assert gpu.n_pass1 >= 100
    assert gpu.counters["used_fast_filter"] == 2
    os.environ["CRASS_NO_HINT_FILTER"] = "1"
    try:
        alt = ca.search_pipeline(seqs)
    finally:
        os.environ.pop("CRASS_NO_HINT_FILTER", None)
    assert_same_pipeline(alt, ref)
    assert alt.counters["used_fast_filter"] == 0


def test_mid_length_reads_with_exception_reads_and_other_options(ca):
    """the hint filter with reads that hold bytes outside ACGT (screened on their packed words, evaluated byte-wise) and with a
    shift range other than the default (-s / -S: the run-time-range hint kernel); another window or seed lattice (-w, -d) has no
    position hints to walk on but the same bits in their every-position form as its filter (k_hint_filter_any); a shift beyond
    127 bases is outside both forms' tile: the general filter"""
    rng = np.random.default_rng(99)
    seqs = _mid_reads(rng, 6000, 280, 900)
    for i in range(0, len(seqs), 37):
        b = bytearray(seqs[i])
        for _ in range(3):
            b[int(rng.integers(0, len(b)))] = ord("N")
        seqs[i] = bytes(b)
    gpu = ca.search_pipeline(seqs)
    assert_same_pipeline(gpu, orc.pipeline(seqs))
    assert gpu.counters["used_fast_filter"] == 2 and gpu.n_pass1 >= 50
    for kw, want in ((dict(lowSpacerSize=20, highSpacerSize=60), 2), (dict(lowDRsize=20, highDRsize=40), 2), (dict(searchWindowLength=7), 2),
                     (dict(searchWindowLength=6), 2), (dict(searchWindowLength=9, minNumRepeats=4), 2), (dict(lowDRsize=15, highDRsize=30), 2),
                     (dict(lowSpacerSize=8, highSpacerSize=30, lowDRsize=15), 2), (dict(highDRsize=64, highSpacerSize=70), 0)):
        prm = ca.default_params(**kw)
        gpu = ca.search_pipeline(seqs, params=prm)
        assert_same_pipeline(gpu, orc.pipeline(seqs, params=to_orc_params(prm)))
        assert gpu.counters["used_fast_filter"] == want, kw


LONG_PARAM_SETS = [
    dict(searchWindowLength=7), dict(lowDRsize=20, highDRsize=40), dict(lowSpacerSize=20, highSpacerSize=60),
    dict(minNumRepeats=3), dict(searchWindowLength=9, minNumRepeats=4),
]


@pytest.mark.parametrize("kw", LONG_PARAM_SETS, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_long_reads_non_default_options(ca, kw):
    """any -w / -d / -D / -s / -S change drops the position hints (launch_hint_positions refuses them) and the
    wave kernel walks every lattice seed — a different code path from the hinted walk, never compared with non-default options
    before.  Ragged 2.1-9 kbp reads with arrays, damaged copies and chance repeats; -n changes what is accepted only."""
    rng = random.Random(101)

    def rs(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    drs = [rs(rng.randint(20, 40)) for _ in range(8)]
    seqs = []
    for i in range(240):
        L = rng.randint(2100, 9000)
        s = bytearray(rs(L))
        kind = i % 5
        if kind in (1, 2, 3):
            dr = bytearray(rng.choice(drs))
            reps = rng.randint(2, 5) if kind == 1 else rng.randint(6, 40)
            arr = bytearray()
            for _ in range(reps):
                d = bytearray(dr)
                if kind == 3 and rng.random() < 0.4:
                    d[rng.randrange(len(d))] = rng.choice(b"ACGT")
                arr += d + rs(rng.randint(18, 62))
            at = rng.choice([0, max(0, L - len(arr)), rng.randint(0, max(0, L - len(arr)))])
            s[at:at + len(arr)] = arr
            s = s[:L]
        elif kind == 4:
            for k in range(5):
                q = rng.randint(0, L - 200)
                s[q + 55:q + 64] = s[q:q + 9]
        seqs.append(bytes(s))
    p = ca.default_params(**kw)
    gpu = ca.search_pipeline(seqs, params=p)
    ref = orc.pipeline(seqs, params=to_orc_params(p))
    assert_same_pipeline(gpu, ref)
    assert gpu.n_pass1 >= 10
    # the windowed LDS layout's second launch under these options too
    os.environ["CRASS_SEQ_WINDOW"] = "1024"
    try:
        alt = ca.search_pipeline(seqs, params=p)
    finally:
        os.environ.pop("CRASS_SEQ_WINDOW", None)
    assert_same_pipeline(alt, ref)


def test_reference_kat_reads_through_the_hip_path(ca):
    """the 20 reads of the reference's own Catch cases (src/test/test_libcrispr.cpp, fixture
    tests/golden/kat_libcrispr.json) through the ENGINE, each with its case's window / minimum spacer: equal to the oracle record
    for record, and — where searchCore itself reproduces the case's seeds (it accepts the read with the case's repeats) — equal to
    the KAT's own start_stops_out / repeat_length.  Window 11 is outside what crass accepts (-w 6..9, crass.cpp) and the engine
    refuses it like the reference's command line does."""
    import json
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_libcrispr.json")))
    assert len(kat["cases"]) == 20
    pinned = refused = 0
    for case in kat["cases"]:
        seq = case["read"].encode()
        kw = dict(lowSpacerSize=case["min_spacer"])
        if case.get("window"):
            kw["searchWindowLength"] = case["window"]
        p = ca.default_params(**kw)
        if case.get("window", 8) > 9:
            with pytest.raises(ca.CrassError):
                ca.SearchEngine(p)
            refused += 1
            continue
        # the KAT read plus filler reads, so that the lane kernel, the filter and the merge all see more than one read
        filler = synth_reads(ca, 200, read_len=len(seq), crispr_per_million=0)
        seqs = filler[:100] + [seq] + filler[100:]
        gpu = ca.search_pipeline(seqs, params=p)
        ref = orc.pipeline(seqs, params=to_orc_params(p))
        assert_same_pipeline(gpu, ref)
        mine = [k for k in range(gpu.n_pass1) if int(gpu.rec_read[k]) == 100]
        if not mine:
            continue                                     # searchCore rejects this read (QC / DR length): equal to the oracle above
        k = mine[0]
        o = int(gpu.rec_ss_off[k]); n = int(gpu.rec_nss[k])
        ss = [int(x) for x in gpu.ss_pool[o:o + n]]
        if not gpu.rec_lowlexi[k]:                       # reverseStartStops (ReadHolder.cpp:321-380): mirror back
            ss = [len(seq) - 1 - x for x in reversed(ss)]
        if case["func"] == "extendPreRepeat":
            assert ss == case["start_stops_out"] and int(gpu.rec_replen[k]) == case["repeat_length"]
            pinned += 1
        else:
            # scanRight's output is the list BEFORE the extension: same repeats, same starts (every case starts at base 0, so
            # the left extension is empty)
            assert len(ss) == case.get("size", len(case["start_stops_out"])) and ss[0::2] == case["start_stops_out"][0::2]
            pinned += 1
    assert refused == 3 and pinned >= 9


def test_long_reads_position_hints(ca):
    """reads beyond the per-read filter get one seed-hint bit per base (k_hint_positions); the hinted seed loop
    must visit exactly the seeds that matter, on and off the stride lattice (rejected candidates move j off it):
    random reads with chance 8-mer repeats, short and damaged arrays, arrays at the read ends"""
    rng = random.Random(23)

    def rs(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    drs = [rs(rng.randint(24, 40)) for _ in range(8)]
    seqs = []
    for i in range(300):
        L = rng.randint(2100, 7000)
        s = bytearray(rs(L))
        kind = i % 6
        if kind in (1, 2, 3):
            dr = bytearray(rng.choice(drs))
            reps = rng.randint(2, 4) if kind == 1 else rng.randint(5, 25)
            arr = bytearray()
            for _ in range(reps):
                d = bytearray(dr)
                if kind == 3 and rng.random() < 0.5:
                    d[rng.randrange(len(d))] = rng.choice(b"ACGT")          # damaged copy
                arr += d + rs(rng.randint(26, 50))
            at = rng.choice([0, max(0, L - len(arr)), rng.randint(0, max(0, L - len(arr)))])
            s[at:at + len(arr)] = arr
            s = s[:L]
        elif kind == 4:
            # a chance-like 8-mer repeat 60 bases apart, several times, on different residues
            for k in range(6):
                p = rng.randint(0, L - 200)
                s[p + 60:p + 68] = s[p:p + 8]
        seqs.append(bytes(s))
    ref = orc.pipeline(seqs)
    os.environ.pop("CRASS_NO_POS_HINTS", None)
    gpu = ca.search_pipeline(seqs)
    assert_same_pipeline(gpu, ref)
    os.environ["CRASS_NO_POS_HINTS"] = "1"
    try:
        plain = ca.search_pipeline(seqs)
    finally:
        os.environ.pop("CRASS_NO_POS_HINTS", None)
    assert_same_pipeline(plain, ref)
    assert gpu.n_pass1 > 60
    # hint slices (CRASS_HINT_PARTS) with exception reads in the set: the survivor list then skips reads, and the walk's
    # slices are cut by the survivors' read indices
    noisy = [bytearray(q) for q in seqs]
    for i in range(7, len(noisy), 11):
        noisy[i][rng.randrange(len(noisy[i]))] = ord("N")
    noisy = [bytes(q) for q in noisy]
    ref_n = orc.pipeline(noisy)
    for parts in ("2", "4"):
        os.environ["CRASS_HINT_PARTS"] = parts
        try:
            sliced = ca.search_pipeline(noisy)
        finally:
            os.environ.pop("CRASS_HINT_PARTS", None)
        assert_same_pipeline(sliced, ref_n)


def test_one_million_reference_known_answer_on_the_hip_path(ca):
    """The HIP pipeline itself against numbers recorded from the COMPILED REFERENCE (SURVEY.md Appendix C /
    §8c: 1 M reads, random.seed(42) -> 5 575 pass-1 reads, 1 988 DR variants, 52 groups, 700 patterns, 9 931 reads
    after pass 2), plus record-level equality with the oracle on the same stream."""
    from tests import appendix_c
    seqs = appendix_c.generate(1000000)
    gpu = ca.search_pipeline(seqs)
    assert (gpu.n_pass1, gpu.n_tokens, gpu.n_groups, gpu.n_patterns, gpu.n_pass1 + gpu.n_pass2) == appendix_c.KNOWN_1M
    assert gpu.counters["used_device_merge"] == 1 and gpu.counters["used_fast_filter"] == 1
    assert_same_pipeline(gpu, orc.pipeline(seqs))


def test_reload_while_the_hand_off_copy_is_in_flight(ca):
    """seed scans whose results are never fetched, each followed at once by a new (larger) read set: the hand-off copy of
    the abandoned step may still be on its DMA engine when the load re-sizes the buffers it writes — the load waits for it"""
    from crass_amd.engine import PackedReads
    seqs = synth_reads(ca, 260000, read_len=150, crispr_per_million=150000, n_dr=30)
    eng = ca.SearchEngine()
    try:
        for n in (60000, 140000, 200000, 260000):
            packed = PackedReads(seqs[:n])
            try:
                eng.load_reads(packed)
                eng.seed_scan(fetch=False)
            finally:
                packed.close()
        assert_same_pipeline(ca.search_pipeline(seqs, engine=eng), orc.pipeline(seqs))
    finally:
        eng.close()


_COPY_PATHS = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import crass_amd as ca
from tests import orc
from tests.parity import assert_same_pipeline
from tests.test_gpu_parity import synth_reads
ca.load()
seqs = synth_reads(ca, 80000, read_len=150, crispr_per_million=60000, n_dr=40)
eng = ca.SearchEngine()
ref = orc.pipeline(seqs)
for rep in range(3):                                     # (a re-used context: the copy of step k must not leak into step k + 1)
    assert_same_pipeline(ca.search_pipeline(seqs, engine=eng), ref)
print("copy path ok", ref.n_pass1)
"""


@pytest.mark.parametrize("env", [{}, {"CRASS_NO_SDMA": "1"}, {"CRASS_NO_SDMA": "1", "CRASS_COPY_EARLY": "1"}, {"CRASS_COPY_BLIT": "1"}])
def test_hand_off_copy_paths(env):
    """pass 1's hand-off records reach the host on a DMA engine (csrc/sdma.cpp); the fall-backs — the library's copy kernel,
    ordered behind the merge or beside it, and the runtime's hipMemcpyAsync — must deliver the same records (the switches
    are read once per process: child processes)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _COPY_PATHS], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "copy path ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
