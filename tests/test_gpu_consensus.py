"""The stage behind the hot path on the GPU (crass_hip_consensus = WorkHorse::findConsensusDRs):
true DRs, group splits / merges, token table, read lists and every read's repaired start/stops must equal the oracle's
(tests/orc.consensus), and the reference's own known answers must come out of the HIP path."""
import os
import random

import numpy as np
import pytest

from tests import orc, fastx
from tests.test_oracle_consensus import KNOWN

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    crass_amd.load()
    return crass_amd


def assert_same_consensus(gpu, ref):
    assert ref.error == 0 and gpu.error == 0
    assert gpu.next_free_gid == ref.next_free_gid
    assert gpu.tokens == ref.tokens
    assert gpu.gids == ref.gids
    assert gpu.true_drs == ref.true_drs
    assert gpu.groups == ref.groups
    assert gpu.reads_of == ref.reads_of
    np.testing.assert_array_equal(gpu.rec_alive, ref.rec_alive)
    np.testing.assert_array_equal(gpu.rec_rc, ref.rec_rc)
    np.testing.assert_array_equal(gpu.rec_token, ref.rec_token)
    np.testing.assert_array_equal(gpu.rec_nss, ref.rec_nss)
    for k in range(len(ref.rec_nss)):
        assert gpu.ss(k) == ref.ss(k), (k, gpu.ss(k), ref.ss(k))


@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "CN_gDC.fa.gz", "front_offset_bug.fa.gz", "Ill.nr.miss.fa.gz", "poor_dr_ext.fa.gz"])
def test_reference_inputs(ca, fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs, hdrs = [r[2] for r in recs], [r[0] for r in recs]
    search = ca.search_pipeline(seqs, hdrs)                     # the HIP search path feeds the HIP consensus stage
    gpu = ca.consensus(seqs, search)
    ref = orc.consensus(seqs, orc.pipeline(seqs, hdrs))
    assert_same_consensus(gpu, ref)
    if fname in KNOWN:                                          # numbers recorded from the COMPILED reference
        drs, counts = KNOWN[fname]
        assert gpu.gids == sorted(drs)
        for gid, dr, n_reads in zip(gpu.gids, gpu.true_drs, gpu.group_read_counts()):
            assert drs[gid] is None or dr == drs[gid]
            assert counts is None or n_reads == counts[gid]
    assert gpu.counters["n_ksw_alignments"] > 0 and gpu.counters["n_placements"] > 0


def synth_reads(ca, n, **kw):
    spec = ca.synth_spec(**kw)
    L = spec.read_len
    W = (L + 15) // 16
    asc = ca.unpack_ascii(ca.synth_packed(spec, 0, n), W, L, n)
    return [asc[i * L:(i + 1) * L].tobytes() for i in range(n)]


@pytest.mark.parametrize("L,n_dr,n,cpm", [(150, 50, 300000, 20000), (101, 5, 60000, 50000), (250, 30, 60000, 40000)])
def test_synthetic(ca, L, n_dr, n, cpm):
    seqs = synth_reads(ca, n, read_len=L, n_dr=n_dr, crispr_per_million=cpm)
    search = ca.search_pipeline(seqs)
    gpu = ca.consensus(seqs, search)
    ref = orc.consensus(seqs, orc.pipeline(seqs))
    assert_same_consensus(gpu, ref)
    assert len(gpu.gids) >= min(n_dr, 4)
    assert gpu.counters["n_sw_tasks"] > 0 and gpu.counters["n_partials_added"] > 0


def test_collapsed_groups_split(ca):
    """two DRs that differ in one base end up in one k-mer cluster: calculateDRConsensus sees a collapsed column and
    splitGroupedDR takes the group apart again (new GIDs, recursion)"""
    rng = random.Random(5)
    base = "".join(rng.choice("ACGT") for _ in range(32))
    alt = base[:28] + ("A" if base[28] != "A" else "C") + base[29:]       # the variants still share 18 11-mers: one cluster
    seqs = []
    for i in range(3000):
        dr = base if i % 2 else alt
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 30)))
        while len(s) < 200:
            s += dr + "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 36)))
        off = rng.randint(0, 30)
        seqs.append(s[off:off + 150].encode())
        seqs.append("".join(rng.choice("ACGT") for _ in range(150)).encode())
    search = ca.search_pipeline(seqs)
    gpu = ca.consensus(seqs, search)
    ref = orc.consensus(seqs, orc.pipeline(seqs))
    assert_same_consensus(gpu, ref)
    # group 1 held both DRs; it is gone, and two new groups with ~1 500 reads each carry one DR each
    assert search.n_groups == 3 and 1 not in gpu.gids and gpu.gids == [2, 3, 6, 7]
    big = [(d, n) for d, n in zip(gpu.true_drs, gpu.group_read_counts()) if n > 1000]
    rc = lambda s: s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    assert sorted(d for d, _ in big) == sorted(min(x.encode(), rc(x.encode())) for x in (base, alt))


def test_reads_with_n_and_reversed_slaves(ca):
    """reads with N bytes (coverage counts them as A, Aligner.cpp:61-70) and DR variants whose reverse complement aligns
    better to the master (Aligner::alignSlave's reversed branch: reads flipped, new token)"""
    rng = random.Random(9)
    seqs = synth_reads(ca, 80000, read_len=150, n_dr=8, crispr_per_million=60000)
    seqs = [bytearray(s) for s in seqs]
    for i in rng.sample(range(len(seqs)), 3000):
        seqs[i][rng.randrange(150)] = ord("N")
    seqs = [bytes(s) for s in seqs]
    search = ca.search_pipeline(seqs)
    gpu = ca.consensus(seqs, search)
    ref = orc.consensus(seqs, orc.pipeline(seqs))
    assert_same_consensus(gpu, ref)
    assert gpu.n_tokens >= search.n_tokens


def test_a_gid_without_a_group_is_skipped_and_bad_tokens_are_refused(ca):
    """a NULL entry of mDR2GIDMap (empty token range) is no group (WorkHorse.cpp:592-595); a group token outside the token table
    is an invalid argument, not an out-of-bounds read"""
    from tests.test_oracle_consensus import WithGap
    recs = fastx.read_fastx(os.path.join(DATA, "front_offset_bug.fa.gz"))
    seqs, hdrs = [r[2] for r in recs], [r[0] for r in recs]
    search = ca.search_pipeline(seqs, hdrs)
    gap = WithGap(search)
    assert_same_consensus(ca.consensus(seqs, gap), orc.consensus(seqs, WithGap(orc.pipeline(seqs, hdrs))))
    bad = WithGap(search)
    bad.groups = [list(g) for g in search.groups]
    bad.groups[0][0] = len(search.tokens) + 5
    with pytest.raises(ca.CrassError) as e:
        ca.consensus(seqs, bad)
    assert e.value.status == 1
