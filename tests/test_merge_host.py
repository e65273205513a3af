"""Host merge (token order + createNonRedundantSet) vs the oracle, and the multi-rank
exchange step on CPU (gloo, world_size 2).  No GPU needed: crass_merge_create is pure host."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import orc, fastx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    from crass_amd import build
    build.build()
    crass_amd.load()
    return crass_amd


def oracle_candidates(res):
    """the representative DR of every pass-1 record, in read order"""
    return [res.tokens[t - 2] for t in res.rec_token[:res.n_pass1]]


@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "front_offset_bug.fa.gz", "CN_gDC.fa.gz"])
def test_merge_matches_oracle_on_reference_inputs(ca, fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    ref = orc.pipeline([r[2] for r in recs], [r[0] for r in recs], do_pass2=False)
    chars, lens = ca.dr_slots(oracle_candidates(ref))
    m = ca.merge_host(chars, lens)
    assert m.tokens == ref.tokens
    assert m.cand_token.tolist() == ref.rec_token[:ref.n_pass1].tolist()
    assert m.groups == ref.groups
    assert m.next_free_gid == ref.n_groups + 1
    assert m.n_patterns == ref.n_patterns
    for g in range(1, ref.n_groups + 1):
        a = [p for p, gg in zip(m.patterns, m.pat_group) if gg == g]
        b = [p for p, gg in zip(ref.patterns, ref.pat_group) if gg == g]
        assert sorted(a) == sorted(b)
        # per group: survivors followed by their reverse complements (WorkHorse.cpp:690-697)
        h = len(a) // 2
        rc = bytes.maketrans(b"ACGTN", b"TGCAN")
        assert [x.translate(rc)[::-1] for x in a[:h]] == a[h:]


def test_merge_synthetic_large(ca):
    spec = ca.synth_spec(read_len=150, crispr_per_million=100000)
    n = 60000
    w = ca.synth_packed(spec, 0, n)
    asc = ca.unpack_ascii(w, 10, 150, n)
    off = np.arange(0, (n + 1) * 150, 150, dtype=np.uint64)
    ref = orc.pipeline((asc, off), do_pass2=False)
    assert ref.n_groups >= 40
    chars, lens = ca.dr_slots(oracle_candidates(ref))
    m = ca.merge_host(chars, lens)
    assert m.tokens == ref.tokens and m.groups == ref.groups
    assert sorted(m.patterns) == sorted(ref.patterns)


def test_merge_many_variants_and_non_acgt_kmers(ca):
    """enough DR variants that every multi-threaded phase of the host merge really runs on several
    threads, plus DRs holding an N (their k-mers take the string-keyed path of clusterDRReads)."""
    import random
    rng = random.Random(99)

    def rs(n, alphabet=b"ACGT"):
        return bytes(rng.choice(alphabet) for _ in range(n))
    drs = [rs(rng.randint(26, 38)) for _ in range(120)]
    for i in range(0, 120, 7):                       # some consensus DRs contain an N
        d = bytearray(drs[i]); d[rng.randint(3, len(d) - 4)] = ord("N"); drs[i] = bytes(d)
    seqs = []
    for _ in range(9000):
        d = bytearray(rng.choice(drs))
        if rng.random() < 0.7:                       # sequencing-error variant, identical in every repeat of the read
            k = rng.randrange(len(d))
            r = rng.random()
            if r < 0.7: d[k] = rng.choice(b"ACGT")
            elif r < 0.85: del d[k]
            else: d.insert(k, rng.choice(b"ACGT"))
        d = bytes(d)
        s = rs(rng.randint(0, 12))
        while len(s) < 190:
            s += d + rs(rng.randint(28, 38))
        seqs.append(s[:rng.randint(150, 190)])
    ref = orc.pipeline(seqs, do_pass2=False)
    assert ref.n_pass1 > 5000 and len(ref.tokens) > 2500
    assert any(b"N" in t for t in ref.tokens)
    chars, lens = ca.dr_slots(oracle_candidates(ref), stride=64)
    for _ in range(3):                               # repeated: the workers are awake after the first call
        m = ca.merge_host(chars, lens)
        assert m.tokens == ref.tokens
        assert m.cand_token.tolist() == ref.rec_token[:ref.n_pass1].tolist()
        assert m.groups == ref.groups
        # the ORDER inside a group is libstdc++'s std::sort/std::partition order in the reference (the oracle is
        # plain C); pass 2 depends on the set only
        assert list(m.pat_group) == list(ref.pat_group)
        assert sorted(zip(list(m.pat_group), m.patterns)) == sorted(zip(list(ref.pat_group), ref.patterns))


WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import crass_amd as ca
from crass_amd.distributed import allgather_candidates
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
data = np.load(%(npz)r)
chars, lens, cut = data["chars"], data["lens"], int(data["cut"])
lo, hi = (0, cut) if rank == 0 else (cut, chars.shape[0])
g_chars, g_lens = allgather_candidates(chars[lo:hi], lens[lo:hi], dist)
m = ca.merge_host(g_chars, g_lens)
# compact exchange: each rank sends only its distinct strings (first-occurrence order)
from crass_amd.distributed import allgather_distinct
seen = {}
for k in range(lo, hi):
    seen.setdefault(bytes(chars[k, :lens[k]]), k)
order = sorted(seen.values())
d_chars, d_lens, my_off = allgather_distinct(chars[order], lens[order], dist)
md = ca.merge_host(d_chars, d_lens)
assert md.tokens == m.tokens and md.groups == m.groups and md.patterns == m.patterns
assert my_off == (0 if rank == 0 else int(d_chars.shape[0]) - len(order))
out = dict(n=int(g_chars.shape[0]), same_chars=bool(np.array_equal(g_chars, chars)), same_lens=bool(np.array_equal(g_lens, lens)),
           tokens=[t.decode() for t in m.tokens], groups=m.groups, patterns=[p.decode() for p in m.patterns])
json.dump(out, open(%(out)r %% rank, "w"))
dist.destroy_process_group()
"""


def test_two_rank_gloo_exchange_builds_identical_pattern_sets(ca, tmp_path):
    recs = fastx.read_fastx(os.path.join(DATA, "front_offset_bug.fa.gz"))
    ref = orc.pipeline([r[2] for r in recs], [r[0] for r in recs], do_pass2=False)
    chars, lens = ca.dr_slots(oracle_candidates(ref))
    cut = len(lens) // 3                      # uneven shards
    npz = str(tmp_path / "cand.npz")
    np.savez(npz, chars=chars, lens=lens, cut=cut)
    outpat = str(tmp_path / "out%d.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, npz=npz, out=outpat))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    import json
    o0, o1 = json.load(open(outpat % 0)), json.load(open(outpat % 1))
    assert o0 == o1                                             # every rank builds the same tables
    assert o0["same_chars"] and o0["same_lens"] and o0["n"] == len(lens)
    single = ca.merge_host(chars, lens)
    assert o0["tokens"] == [t.decode() for t in single.tokens]
    assert o0["groups"] == single.groups
    assert o0["patterns"] == [p.decode() for p in single.patterns]
    assert o0["groups"] == ref.groups


@pytest.mark.parametrize("fname", ["front_offset_bug.fa.gz", "Ill100.fx.gz", "CN_gDC.fa.gz"])
def test_host_view_rebuild_from_per_token_results(ca, fname):
    """what the engine does after the device merge (tokens / groups / pattern list from GID + dropped flag per
    token, reference sort/partition order, lazily indexed token table), fed with the host merge's own results:
    it must reproduce that merge field for field"""
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    ref = orc.pipeline([r[2] for r in recs], [r[0] for r in recs], do_pass2=False)
    cands = oracle_candidates(ref)
    chars, lens = ca.dr_slots(cands)
    m = ca.merge_host(chars, lens)
    # distinct strings in first-occurrence order and every candidate's index among them
    first, order = {}, []
    cand_distinct = np.empty(len(cands), np.uint32)
    for k, s in enumerate(cands):
        if s not in first:
            first[s] = len(order)
            order.append(k)
        cand_distinct[k] = first[s]
    assert [cands[k] for k in order] == m.tokens
    gid_of = np.zeros(len(order), np.uint32)
    for g, toks in enumerate(m.groups):
        for t in toks:
            gid_of[t - 2] = g + 1
    kept = set(m.patterns)
    dropped = np.array([0 if t in kept else 1 for t in m.tokens], np.uint8)
    r = ca.merge_rebuild(chars[order], lens[order], cand_distinct, gid_of, dropped, len(m.groups))
    assert r.tokens == m.tokens and r.groups == m.groups
    assert r.patterns == m.patterns and list(r.pat_group) == list(m.pat_group)
    assert r.cand_token.tolist() == m.cand_token.tolist()
    assert r.next_free_gid == m.next_free_gid
    # inconsistent input is refused, not trusted
    bad = gid_of.copy()
    bad[0] = len(m.groups) + 5
    with pytest.raises(ca.CrassError):
        ca.merge_rebuild(chars[order], lens[order], cand_distinct, bad, dropped, len(m.groups))


def test_include_substring_with_non_involutive_complement():
    """removeRedundantRepeats blanks b when b contains a OR reverseComplement(a) (includeSubstring, WorkHorse.cpp:78-86).
    comp_tab maps U -> A but A -> T (SeqUtils.cpp:50-59), so for a DR with a 'U' "rc(a) in b" and "a in rc(b)" differ:
    the host merge must test the reference's two needles literally."""
    import crass_amd as ca
    comp = {ord(x): ord(y) for x, y in zip("ACGTU", "TGCAA")}
    a = b"ACGGTCATTCAAGGCTAGCTTGACU"                          # 25 bases, ends in U
    rc_a = bytes(comp[c] for c in reversed(a))                # starts with A (the complement of U)
    b = b"GG" + rc_a + b"CC"                                  # contains rc(a) but not a; rc(b) does not contain a either
    rc_b = bytes(comp[c] for c in reversed(b))
    assert a not in b and rc_a in b and a not in rc_b
    chars, lens = ca.dr_slots([a, b])
    m = ca.merge_host(chars, lens)
    assert m.n_groups == 1
    assert sorted(m.patterns) == sorted([a, rc_a])            # b was dropped, as the reference drops it
