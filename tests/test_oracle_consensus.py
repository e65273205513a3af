"""The oracle of the stage behind the hot path (findConsensusDRs — true-DR consensus, group
splitting, start/stop repair):
* its ksw_align restatement against the COMPILED reference ksw.c (oracle/_ref), with the scoring the Aligner uses;
* the whole stage against the known answers SURVEY §8c recorded from the compiled reference: true DR strings, group ids
  and per-group read counts on the reference's own regression inputs."""
import os
import random

import numpy as np
import pytest

from tests import orc, fastx

DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


def test_ksw_align_matches_the_compiled_reference():
    if orc.ref() is None or not hasattr(orc.ref(), "ref_ksw_align"):
        pytest.skip("oracle/_ref not available")
    rng = random.Random(11)
    n = 0
    for case in range(6000):
        tl = rng.randint(1, 60)
        t = [rng.randrange(4) for _ in range(tl)]
        kind = rng.random()
        if kind < 0.5:                      # a mutated piece of the target (the DR-variant case)
            a, b = sorted((rng.randrange(tl + 1), rng.randrange(tl + 1)))
            q = t[a:b] or [rng.randrange(4)]
            for _ in range(rng.choice([0, 0, 1, 2, 3])):
                op = rng.random()
                pos = rng.randrange(len(q))
                if op < 0.5:
                    q[pos] = rng.randrange(5)               # 4 = ambiguous base
                elif op < 0.75 and len(q) > 1:
                    del q[pos]
                else:
                    q.insert(pos, rng.randrange(4))
            if rng.random() < 0.3:
                q = [rng.randrange(4) for _ in range(rng.randint(0, 4))] + q + [rng.randrange(4) for _ in range(rng.randint(0, 4))]
        elif kind < 0.8:
            q = [rng.randrange(5) for _ in range(rng.randint(1, 60))]
        else:                                # low-complexity: ties everywhere
            q = [rng.choice([0, 0, 0, 1]) for _ in range(rng.randint(1, 40))]
            t = [rng.choice([0, 0, 0, 1]) for _ in range(tl)]
        got, want = orc.ksw_align(q, t, "oracle"), orc.ksw_align(q, t, "ref")
        assert got == want, (q, t, got, want)
        n += want[0] >= 5
    assert n > 2000


def test_smith_waterman_basics():
    dr = b"GTTTCAATCCACGCGCCCACGCGGGGCGCGAC"
    read = b"TTATTATATTATTTATATATTATTATAT" + dr[-12:]        # the END of a repeat at the read end: found, but it is no repeat start
    r, s, e, a, b = orc.smith_waterman(read, dr, 10, len(read) - 10)
    # (the traceback also takes in the zero-score cell in front of the local alignment: SmithWaterman.cpp:246-262)
    assert r == 1 and e == len(read) - 1 and a == read[-15:] and b == dr[-15:] and dr.find(b) != 0
    read2 = b"ACGTTGCAGGATCTTACGATCGGATCAG" + dr[:14]       # a repeat START at the read end: the case updateStartStops adds
    r, s, e, a, b = orc.smith_waterman(read2, dr, 10, len(read2) - 10)
    assert r == 1 and (s, e) == (28, len(read2) - 1) and a == b == dr[:14] and dr.find(b) == 0
    # the reference cuts a_ret with a length that includes aStartSearch (SmithWaterman.cpp:279): with a search window that
    # starts inside the read, a_ret runs on to the end of the read and the similarity is taken over that longer string
    read3 = b"TTATTATATTATTTATATATTATTATAT" + dr[:14] + b"TTATATTAT"
    r0, s0, e0, a0, b0 = orc.smith_waterman(read3, dr, 10, len(read3) - 10, 0.0)
    assert r0 == 1 and (s0, e0) == (28, 41) and a0 == read3[28:] and b0 == dr[:14]
    assert orc.smith_waterman(read3, dr, 10, len(read3) - 10) == (0, 0, 0, b"", b"")     # not similar enough: ("", "") and zeros


# file: (true DRs by GID, reads per group)  — SURVEY.md §8c
KNOWN = {
    "Ill100.fx.gz": ({1: b"CGGTTCATCCCCGCGCCTGCGGGGAACGC"}, {1: 4312}),
    "CN_gDC.fa.gz": ({1: b"CTTTTAATCGCACCTATTTGGAATTGAAAC"}, None),
    "front_offset_bug.fa.gz": ({10: b"CGCTCTGGCCGGTCTCCGACCGAGCCAGCCC", 20: None, 44: None, 45: None}, {10: 303, 20: 102, 44: 35, 45: 28}),
}


@pytest.mark.parametrize("fname", sorted(KNOWN))
def test_true_dr_known_answers(fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs, hdrs = [r[2] for r in recs], [r[0] for r in recs]
    res = orc.pipeline(seqs, hdrs)
    con = orc.consensus(seqs, res)
    assert con.error == 0
    drs, counts = KNOWN[fname]
    assert con.gids == sorted(drs)
    for gid, dr, n_reads in zip(con.gids, con.true_drs, con.group_read_counts()):
        if drs[gid] is not None:
            assert dr == drs[gid]
        if counts is not None:
            assert n_reads == counts[gid]
        assert dr <= orc_revcomp(dr)                           # laurenized
    # structural invariants of the repaired records
    for k in range(len(con.rec_alive)):
        if con.rec_alive[k] and con.rec_token[k]:
            ss = con.ss(k)
            assert len(ss) % 2 == 0 and ss == sorted(ss)


def orc_revcomp(s):
    import ctypes as C
    out = C.create_string_buffer(len(s))
    orc.lib().orc_revcomp(s, len(s), out)
    return out.raw


@pytest.mark.parametrize("fname", ["Ill.nr.miss.fa.gz", "poor_dr_ext.fa.gz"])
def test_other_reference_inputs_run_clean(fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs, hdrs = [r[2] for r in recs], [r[0] for r in recs]
    con = orc.consensus(seqs, orc.pipeline(seqs, hdrs))
    assert con.error == 0 and len(con.gids) >= 1


class WithGap:
    """a search result whose mDR2GIDMap has a NULL entry at GID 1 (every original group moves up by one)"""

    def __init__(self, res):
        self.__dict__.update(res.__dict__)
        self.groups = [[]] + [list(g) for g in res.groups]


def test_a_gid_without_a_group_is_skipped():
    """WorkHorse::findConsensusDRs `continue`s over a NULL group (WorkHorse.cpp:592-595): same true DRs one GID higher"""
    recs = fastx.read_fastx(os.path.join(DATA, "front_offset_bug.fa.gz"))
    seqs, hdrs = [r[2] for r in recs], [r[0] for r in recs]
    res = orc.pipeline(seqs, hdrs)
    a, b = orc.consensus(seqs, res), orc.consensus(seqs, WithGap(res))
    assert a.error == 0 and b.error == 0
    assert b.gids == [g + 1 for g in a.gids] and b.true_drs == a.true_drs and b.groups == a.groups
    assert b.next_free_gid == a.next_free_gid + 1
