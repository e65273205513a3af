"""SURVEY.md Appendix C read stream (the one the compiled reference's known answer was recorded on:
1 M reads -> 5 575 pass-1 reads, 1 988 DR variants, 52 groups, 700 patterns, 9 931 reads after pass 2).

The recipe is a pure-Python `random.seed(42)` loop (`generate_reference_loop`, ~100 s for 1 M reads).
`generate` produces the SAME byte stream ~20x faster by replaying CPython's Mersenne-Twister word
stream through numpy: `random.choice("ACGT")` is `_randbelow(4)` = top 3 bits of a 32-bit word,
redrawn while >= 4; `randint` / `choice(list)` are `_randbelow(n)` on the top n.bit_length() bits;
`random.random()` takes two words.  tests/test_oracle_pipeline.py checks both generators agree.
"""
import random

import numpy as np

L = 150
ACGT = np.frombuffer(b"ACGT", np.uint8)


def generate_reference_loop(n_reads, seed=42):
    """the recipe as written in SURVEY.md Appendix C (the DR is drawn BEFORE the prefix)"""
    rnd = random.Random(seed)

    def rand_seq(n):
        return "".join(rnd.choice("ACGT") for _ in range(n))
    drs = [rand_seq(rnd.randint(28, 37)) for _ in range(50)]
    seqs = []
    for _ in range(n_reads):
        if rnd.random() < 0.01:
            dr = rnd.choice(drs)
            s = rand_seq(rnd.randint(0, 40))
            while len(s) < L + 60:
                s += dr + rand_seq(rnd.randint(30, 38))
            off = rnd.randint(0, 40)
            s = s[off:off + L]
        else:
            s = rand_seq(L)
        seqs.append(s.encode())
    return seqs


class _Words:
    """CPython's MT19937 output words for random.Random(seed), in blocks"""

    def __init__(self, seed, block=1 << 24):
        st = random.Random(seed).getstate()[1]
        self.bg = np.random.MT19937()
        self.bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[:-1], dtype=np.uint32), "pos": int(st[-1])}}
        self.block = block
        self.w = np.zeros(0, np.uint32)
        self.i = 0
        self._prep()

    def _prep(self):
        self.r3 = (self.w >> 29).astype(np.uint8)
        self.acc = np.cumsum(self.r3 < 4, dtype=np.int64)

    def ensure(self, need):
        if len(self.w) - self.i >= need:
            return
        more = self.bg.random_raw(max(self.block, need)).astype(np.uint32)
        self.w = np.concatenate([self.w[self.i:], more])
        self.i = 0
        self._prep()

    def randbelow(self, n):
        k = n.bit_length()
        while True:
            self.ensure(1)
            r = int(self.w[self.i]) >> (32 - k)
            self.i += 1
            if r < n:
                return r

    def randint(self, a, b):
        return a + self.randbelow(b - a + 1)

    def rand(self):
        self.ensure(2)
        a, b = int(self.w[self.i]) >> 5, int(self.w[self.i + 1]) >> 6
        self.i += 2
        return (a * 67108864.0 + b) / 9007199254740992.0

    def rand_seq(self, n):
        """n accepted draws of _randbelow(4) as ASCII bytes"""
        if n == 0:
            return b""
        self.ensure(64 * n + 64)                      # acceptance rate 1/2: far more than enough
        before = self.acc[self.i - 1] if self.i else 0
        end = int(np.searchsorted(self.acc, before + n, side="left"))      # index of the n-th accepted word
        seg = self.r3[self.i:end + 1]
        self.i = end + 1
        return ACGT[seg[seg < 4]].tobytes()


def generate(n_reads, seed=42):
    """list[bytes], identical to generate_reference_loop(n_reads, seed)"""
    wd = _Words(seed)
    drs = [wd.rand_seq(wd.randint(28, 37)) for _ in range(50)]
    seqs = []
    for _ in range(n_reads):
        if wd.rand() < 0.01:
            dr = drs[wd.randbelow(50)]
            s = wd.rand_seq(wd.randint(0, 40))
            while len(s) < L + 60:
                s += dr + wd.rand_seq(wd.randint(30, 38))
            off = wd.randint(0, 40)
            s = s[off:off + L]
        else:
            s = wd.rand_seq(L)
        seqs.append(s)
    return seqs


KNOWN_1M = (5575, 1988, 52, 700, 9931)      # pass-1 reads, variants, groups, patterns, reads after pass 2 (SURVEY.md:407)
