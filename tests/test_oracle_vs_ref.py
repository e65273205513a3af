"""Oracle leaf functions vs the *compiled reference sources* (oracle/_ref, built by
oracle/Makefile from /root/reference: PatternMatcher.cpp, StringCheck.cpp, kseq.cpp,
aho-corasick/*.c).  Skipped where the reference build is unavailable."""
import ctypes as C
import os
import random
import struct

import pytest

from tests import orc, fastx

ref = orc.ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (no /root/reference)")

DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


def rs(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(alpha) for _ in range(n))


def test_bmp_search_random_and_edges():
    L = orc.lib()
    rng = random.Random(1)
    for _ in range(20000):
        alpha = rng.choice([b"ACGT", b"AC", b"ACGTN"])
        t = rs(rng, rng.randint(0, 60), alpha)
        p = rs(rng, rng.randint(0, 10), alpha)
        a = L.orc_bmp_search(t, len(t), p, len(p))
        b = ref.ref_bmp_search(t, len(t), p, len(p))
        assert a == b, (t, p)
        if p and len(p) <= len(t):
            assert a == t.find(p)


def test_levenshtein_and_similarity():
    L = orc.lib()
    rng = random.Random(2)
    pairs = [(b"ABCDEF", b"ABDCEF"), (b"ABCDEF", b"BACDEF"), (b"", b"ACGT"), (b"AC", b"ACGT"),
             (b"ACG", b"ACG"), (b"A" * 50, b"A" * 41 + b"C" * 9)]
    for _ in range(3000):
        n, m = rng.randint(0, 70), rng.randint(0, 70)
        s = rs(rng, n)
        if rng.random() < 0.5 and n:
            # mutated copy: exercises the transposition term
            t = bytearray(s)
            for _k in range(rng.randint(0, 6)):
                if len(t) > 2:
                    i = rng.randrange(len(t) - 1)
                    op = rng.random()
                    if op < 0.4:
                        t[i], t[i + 1] = t[i + 1], t[i]
                    elif op < 0.7:
                        del t[i]
                    else:
                        t.insert(i, rng.choice(b"ACGT"))
            t = bytes(t)
        else:
            t = rs(rng, m)
        pairs.append((s, t))
    for s, t in pairs:
        assert L.orc_levenshtein(s, len(s), t, len(t)) == ref.ref_levenshtein(s, len(s), t, len(t)), (s, t)
        a = L.orc_similarity(s, len(s), t, len(t))
        b = ref.ref_similarity(s, len(s), t, len(t))
        assert struct.pack("f", a) == struct.pack("f", b), (s, t, a, b)
    # documented quirk
    assert L.orc_levenshtein(b"ABCDEF", 6, b"ABDCEF", 6) == 1
    assert L.orc_levenshtein(b"ABCDEF", 6, b"BACDEF", 6) == 2
    # d=9, max=50 -> 0.819999993 (float) which is NOT > 0.82
    s, t = b"A" * 50, b"A" * 41 + b"C" * 9
    assert L.orc_levenshtein(s, 50, t, 50) == 9
    assert float(L.orc_similarity(s, 50, t, 50)) <= 0.82


def test_first_match_semantics_vs_acism():
    rng = random.Random(3)
    for trial in range(300):
        alpha = rng.choice([b"ACGT", b"AC"])
        base = [rs(rng, rng.randint(4, 14), alpha) for _ in range(rng.randint(1, 6))]
        pats = []
        for b in base:
            pats.append(b)
            # variants: substrings / extensions / duplicates, like DR variant sets
            for _ in range(rng.randint(0, 4)):
                i = rng.randint(0, len(b) - 2)
                j = rng.randint(i + 2, len(b))
                pats.append(rs(rng, rng.randint(0, 2), alpha) + b[i:j] + rs(rng, rng.randint(0, 2), alpha))
        if rng.random() < 0.3:
            pats.append(pats[0])
        A = orc.PatternSet(pats, "oracle")
        R = orc.PatternSet(pats, "ref")
        for _ in range(60):
            txt = bytearray(rs(rng, rng.randint(0, 80), alpha + (b"N" if rng.random() < 0.3 else b"")))
            if rng.random() < 0.7 and len(txt) > 20:
                p = rng.choice(pats)
                k = rng.randint(0, len(txt) - 1)
                txt[k:k] = p
            txt = bytes(txt)
            a, b = A.first(txt), R.first(txt)
            assert a == b, (pats, txt, a, b)
            if a is not None:
                e, l = a
                # min end position, ties -> longest
                ends = [(i + len(p), -len(p)) for p in set(pats) for i in range(len(txt) - len(p) + 1)
                        if txt[i:i + len(p)] == p]
                assert (e, -l) == min(ends)
        A.close()
        R.close()


def test_stringcheck_tokens():
    rng = random.Random(4)
    pool = [rs(rng, rng.randint(23, 47)) for _ in range(50)]
    seq = [rng.choice(pool) for _ in range(500)]
    arr, lens = orc._strarr(seq)
    out = (C.c_int32 * len(seq))()
    ref.ref_stringcheck_tokens(arr, lens, len(seq), out)
    # oracle/host rule: first token 2, discovery order
    seen = {}
    for s, t in zip(seq, out):
        if s not in seen:
            seen[s] = len(seen) + 2
        assert seen[s] == t


@pytest.mark.parametrize("fname", sorted(os.listdir(DATA)))
def test_python_fastx_reader_matches_kseq(fname):
    path = os.path.join(DATA, fname)
    out = C.POINTER(C.c_ubyte)()
    ol = C.c_size_t()
    lr = C.c_int()
    n = ref.ref_kseq_dump(path.encode(), C.byref(out), C.byref(ol), C.byref(lr))
    raw = C.string_at(out, ol.value)
    ref.ref_free(out)
    pos = 0
    recs = []
    for _ in range(n):
        f = []
        for _k in range(4):
            (l,) = struct.unpack_from("<I", raw, pos)
            pos += 4
            if l == 0xFFFFFFFF:
                f.append(None)
            else:
                f.append(raw[pos:pos + l])
                pos += l
        recs.append(tuple(f))
    assert recs == fastx.read_fastx(path)
