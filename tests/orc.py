"""ctypes bindings for the test-only checkers under oracle/.

* ``liboracle.so``  — the plain-C restatement (oracle/crass_oracle.c)
* ``_ref/libcrass_ref.so`` — the reference's own leaf sources compiled where they lie
  (only present where oracle/Makefile found /root/reference; optional)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


class Params(C.Structure):
    _fields_ = [("lowDRsize", C.c_uint32), ("highDRsize", C.c_uint32),
                ("lowSpacerSize", C.c_uint32), ("highSpacerSize", C.c_uint32),
                ("searchWindowLength", C.c_uint32), ("minNumRepeats", C.c_uint32),
                ("kmer_clust_size", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        p = cls(23, 47, 26, 50, 8, 2, 6)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def astuple(self):
        return tuple(getattr(self, f[0]) for f in self._fields_)


class View(C.Structure):
    _fields_ = [("n_pass1", C.c_uint64), ("n_pass2", C.c_uint64),
                ("n_tokens", C.c_uint32), ("n_groups", C.c_uint32),
                ("n_patterns", C.c_uint32), ("max_read_len", C.c_uint32),
                ("error", C.c_int32),
                ("rec_read", C.POINTER(C.c_uint64)), ("rec_lowlexi", C.POINTER(C.c_uint8)),
                ("rec_token", C.POINTER(C.c_uint32)), ("rec_replen", C.POINTER(C.c_uint32)),
                ("rec_nss", C.POINTER(C.c_uint32)), ("rec_ss_off", C.POINTER(C.c_uint64)),
                ("ss_pool", C.POINTER(C.c_uint32)),
                ("tok_chars", C.POINTER(C.c_char)), ("tok_off", C.POINTER(C.c_uint64)),
                ("grp_tokens", C.POINTER(C.c_uint32)), ("grp_off", C.POINTER(C.c_uint64)),
                ("pat_chars", C.POINTER(C.c_char)), ("pat_off", C.POINTER(C.c_uint64)),
                ("pat_group", C.POINTER(C.c_uint32))]


class ConsInput(C.Structure):
    _fields_ = [("seqs", C.c_void_p), ("seq_off", C.c_void_p), ("n_reads", C.c_uint64),
                ("n_rec", C.c_uint64), ("rec_read", C.c_void_p), ("rec_lowlexi", C.c_void_p), ("rec_token", C.c_void_p),
                ("rec_nss", C.c_void_p), ("rec_ss_off", C.c_void_p), ("ss_pool", C.c_void_p),
                ("n_tokens", C.c_uint32), ("tok_chars", C.c_void_p), ("tok_off", C.c_void_p),
                ("n_groups", C.c_uint32), ("grp_tokens", C.c_void_p), ("grp_off", C.c_void_p),
                ("max_read_len", C.c_uint32)]


class ConsView(C.Structure):
    _fields_ = [("error", C.c_int32), ("next_free_gid", C.c_int32), ("n_tokens", C.c_uint32),
                ("tok_chars", C.POINTER(C.c_char)), ("tok_off", C.POINTER(C.c_uint64)),
                ("n_groups", C.c_uint32), ("grp_gid", C.POINTER(C.c_int32)), ("dr_chars", C.POINTER(C.c_char)),
                ("dr_off", C.POINTER(C.c_uint64)), ("grp_tokens", C.POINTER(C.c_uint32)), ("grp_off", C.POINTER(C.c_uint64)),
                ("n_rec", C.c_uint64), ("rec_alive", C.POINTER(C.c_uint8)), ("rec_rc", C.POINTER(C.c_uint8)),
                ("rec_token", C.POINTER(C.c_uint32)), ("rec_nss", C.POINTER(C.c_uint32)), ("rec_ss_off", C.POINTER(C.c_uint64)),
                ("ss_pool", C.POINTER(C.c_uint32)), ("tokread_off", C.POINTER(C.c_uint64)), ("tokread_idx", C.POINTER(C.c_uint64)),
                ("tok_has_list", C.POINTER(C.c_uint8))]


_lib = None
_ref = None
ORACLE_LIB_OVERRIDE = None      # tools/longread_phases.py: its own build of the oracle with the work counters compiled in


def lib():
    global _lib
    if _lib is None:
        path = ORACLE_LIB_OVERRIDE or os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_bmp_search.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_bmp_search.restype = C.c_int
        L.orc_levenshtein.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_levenshtein.restype = C.c_int
        L.orc_similarity.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_similarity.restype = C.c_float
        L.orc_revcomp.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_is_low_complexity.argtypes = [C.c_char_p, C.c_int]
        L.orc_scan_right.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int,
                                     C.c_char_p, C.c_int, C.c_uint32, C.c_uint32]
        L.orc_extend_pre_repeat.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
        L.orc_extend_pre_repeat.restype = C.c_uint32
        L.orc_qc_found_repeats.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
        L.orc_search_core.argtypes = [C.c_char_p, C.c_int, C.POINTER(Params), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint32)]
        L.orc_has_lattice_hit.argtypes = [C.c_char_p, C.c_int, C.POINTER(Params)]
        L.orc_dr_low_lexi.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_char_p,
                                      C.POINTER(C.c_int)]
        L.orc_ac_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32]
        L.orc_ac_create.restype = C.c_void_p
        L.orc_ac_destroy.argtypes = [C.c_void_p]
        L.orc_ac_first_match.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32)]
        L.orc_pipeline_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.POINTER(Params), C.c_int]
        L.orc_pipeline_run.restype = C.c_void_p
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_result_view.argtypes = [C.c_void_p, C.POINTER(View)]
        L.orc_pipeline_time.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Params)] + \
            [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_uint64)] * 2 + [C.POINTER(C.c_uint32)]
        L.orc_ksw_align.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + \
            [C.POINTER(C.c_int)] * 5
        L.orc_ksw_align.restype = None
        L.orc_smith_waterman.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                                         C.c_double] + [C.POINTER(C.c_int)] * 4
        L.orc_consensus_run.argtypes = [C.POINTER(ConsInput), C.POINTER(Params)]
        L.orc_consensus_run.restype = C.c_void_p
        L.orc_consensus_view.argtypes = [C.c_void_p, C.POINTER(ConsView)]
        L.orc_consensus_free.argtypes = [C.c_void_p]
        L.orc_calib_bmp.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Params), C.POINTER(C.c_uint64)]
        L.orc_calib_bmp.restype = C.c_double
        L.orc_calib_ac.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_calib_ac.restype = C.c_double
        _lib = L
    return _lib


def ref():
    """The compiled reference leaf library, or None when it is not available."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libcrass_ref.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build()
            if not os.path.exists(path):
                return None
        R = C.CDLL(path)
        R.ref_bmp_search.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        R.ref_levenshtein.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        R.ref_similarity.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        R.ref_similarity.restype = C.c_float
        R.ref_acism_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32]
        R.ref_acism_build.restype = C.c_void_p
        R.ref_acism_first.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
        R.ref_acism_free.argtypes = [C.c_void_p]
        R.ref_stringcheck_tokens.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32,
                                             C.POINTER(C.c_int32)]
        R.ref_kseq_dump.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_int)]
        R.ref_kseq_dump.restype = C.c_long
        R.ref_free.argtypes = [C.c_void_p]
        if hasattr(R, "ref_ksw_align"):
            R.ref_ksw_align.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + \
                [C.POINTER(C.c_int)] * 5
            R.ref_ksw_align.restype = None
        if hasattr(R, "ref_calib_bmp"):
            R.ref_calib_bmp.argtypes = [C.c_void_p, C.c_uint64, C.c_int] + [C.c_uint] * 5 + [C.POINTER(C.c_uint64)]
            R.ref_calib_bmp.restype = C.c_double
            R.ref_calib_acism.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
            R.ref_calib_acism.restype = C.c_double
        _ref = R
    return _ref


def _strarr(strs):
    arr = (C.c_char_p * len(strs))(*strs)
    lens = (C.c_uint32 * len(strs))(*[len(s) for s in strs])
    return arr, lens


class PatternSet:
    """first-match searcher over a pattern list; impl = 'oracle' or 'ref'."""

    def __init__(self, patterns, impl="oracle"):
        self.impl = impl
        self._keep = _strarr(list(patterns))
        if impl == "oracle":
            self.h = lib().orc_ac_create(self._keep[0], self._keep[1], len(patterns))
        else:
            self.h = ref().ref_acism_build(self._keep[0], self._keep[1], len(patterns))

    def first(self, text):
        e, l = C.c_uint32(), C.c_uint32()
        if self.impl == "oracle":
            r = lib().orc_ac_first_match(self.h, text, len(text), C.byref(e), C.byref(l))
        else:
            r = ref().ref_acism_first(self.h, text, len(text), C.byref(e), C.byref(l))
        return (e.value, l.value) if r else None

    def close(self):
        if self.h:
            (lib().orc_ac_destroy if self.impl == "oracle" else ref().ref_acism_free)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_core(seq, params=None):
    p = params or Params.default()
    cap = len(seq) + 8
    ss = (C.c_uint32 * cap)()
    nss = C.c_int(0)
    rl = C.c_uint32(0)
    r = lib().orc_search_core(seq, len(seq), C.byref(p), ss, C.byref(nss), cap, C.byref(rl))
    return r, list(ss[:nss.value]), rl.value


def concat(items):
    """list of bytes -> (uint8 array, uint64 offsets[n+1])"""
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        off[1:] = np.cumsum([len(s) for s in items], dtype=np.uint64)
    buf = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if items else np.zeros(0, np.uint8)
    return buf, off


class PipelineResult:
    """numpy copies of an orc_view"""

    def __init__(self, v):
        n = int(v.n_pass1 + v.n_pass2)
        self.n_pass1, self.n_pass2 = int(v.n_pass1), int(v.n_pass2)
        self.n_tokens, self.n_groups, self.n_patterns = v.n_tokens, v.n_groups, v.n_patterns
        self.max_read_len, self.error = v.max_read_len, v.error

        def arr(ptr, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt, copy=True)
        self.rec_read = arr(v.rec_read, n, np.uint64)
        self.rec_lowlexi = arr(v.rec_lowlexi, n, np.uint8)
        self.rec_token = arr(v.rec_token, n, np.uint32)
        self.rec_replen = arr(v.rec_replen, n, np.uint32)
        self.rec_nss = arr(v.rec_nss, n, np.uint32)
        self.rec_ss_off = arr(v.rec_ss_off, n, np.uint64)
        nss_total = int(self.rec_nss.sum())
        self.ss_pool = arr(v.ss_pool, nss_total, np.uint32)
        self.tok_off = arr(v.tok_off, v.n_tokens + 1, np.uint64)
        tc = C.string_at(v.tok_chars, int(self.tok_off[-1])) if v.n_tokens else b""
        self.tokens = [tc[int(self.tok_off[i]):int(self.tok_off[i + 1])] for i in range(v.n_tokens)]
        self.grp_off = arr(v.grp_off, v.n_groups + 1, np.uint64)
        gt = arr(v.grp_tokens, int(self.grp_off[-1]) if v.n_groups else 0, np.uint32)
        self.groups = [gt[int(self.grp_off[i]):int(self.grp_off[i + 1])].tolist() for i in range(v.n_groups)]
        self.pat_off = arr(v.pat_off, v.n_patterns + 1, np.uint64)
        pc = C.string_at(v.pat_chars, int(self.pat_off[-1])) if v.n_patterns else b""
        self.patterns = [pc[int(self.pat_off[i]):int(self.pat_off[i + 1])] for i in range(v.n_patterns)]
        self.pat_group = arr(v.pat_group, v.n_patterns, np.uint32)

    def ss(self, k):
        o = int(self.rec_ss_off[k])
        return self.ss_pool[o:o + int(self.rec_nss[k])].tolist()


def pipeline(seqs, headers=None, params=None, do_pass2=True):
    """seqs: list[bytes] or (uint8 array, uint64 offsets)."""
    p = params or Params.default()
    if isinstance(seqs, (list, tuple)) and (not seqs or isinstance(seqs[0], (bytes, bytearray))):
        sbuf, soff = concat(list(seqs))
    else:
        sbuf, soff = seqs
    n = len(soff) - 1
    if headers is not None:
        hbuf, hoff = concat(list(headers))
        hp, hop = hbuf.ctypes.data, hoff.ctypes.data
    else:
        hp, hop = None, None
    r = lib().orc_pipeline_run(sbuf.ctypes.data, soff.ctypes.data, n, hp, hop, C.byref(p), int(do_pass2))
    v = View()
    lib().orc_result_view(r, C.byref(v))
    out = PipelineResult(v)
    lib().orc_result_free(r)
    return out


def pipeline_time(sbuf, soff, params=None):
    p = params or Params.default()
    t = [C.c_double() for _ in range(3)]
    n1, n2, npat = C.c_uint64(), C.c_uint64(), C.c_uint32()
    err = lib().orc_pipeline_time(sbuf.ctypes.data, soff.ctypes.data, len(soff) - 1, C.byref(p),
                                  C.byref(t[0]), C.byref(t[1]), C.byref(t[2]),
                                  C.byref(n1), C.byref(n2), C.byref(npat))
    return dict(error=err, t_pass1=t[0].value, t_merge=t[1].value, t_pass2=t[2].value,
                n_pass1=n1.value, n_pass2=n2.value, n_patterns=npat.value)


def pipeline_time_keep(sbuf, soff, params=None):
    """pipeline_time that also returns the run's PipelineResult under "result" (the timed run IS the checker's run)"""
    p = params or Params.default()
    t = [C.c_double() for _ in range(3)]
    L = lib()
    L.orc_pipeline_time_keep.restype = C.c_void_p
    r = L.orc_pipeline_time_keep(C.c_void_p(sbuf.ctypes.data), C.c_void_p(soff.ctypes.data), C.c_uint64(len(soff) - 1), C.byref(p),
                                 C.byref(t[0]), C.byref(t[1]), C.byref(t[2]))
    r = C.c_void_p(r)
    v = View()
    L.orc_result_view(r, C.byref(v))
    out = PipelineResult(v)
    L.orc_result_free(r)
    return dict(error=out.error, t_pass1=t[0].value, t_merge=t[1].value, t_pass2=t[2].value,
                n_pass1=out.n_pass1, n_pass2=out.n_pass2, n_patterns=out.n_patterns, result=out)


def calibrate(asc, n, L, patterns, params=None):
    """Speed of the oracle's two hot leaf loops against the compiled reference's on the same reads
    (asc: uint8 [n*L], patterns: list[bytes]).  None when oracle/_ref is not available.
    ratio > 1: the oracle is faster than the reference by that factor."""
    R = ref()
    if R is None or not hasattr(R, "ref_calib_bmp") or not patterns:
        return None
    p = params or Params.default()
    Lb = lib()
    ck_o, ck_r = C.c_uint64(), C.c_uint64()
    t_o_bmp = Lb.orc_calib_bmp(asc.ctypes.data, n, L, C.byref(p), C.byref(ck_o))
    t_r_bmp = R.ref_calib_bmp(asc.ctypes.data, n, L, p.lowDRsize, p.highDRsize, p.lowSpacerSize, p.highSpacerSize,
                              p.searchWindowLength, C.byref(ck_r))
    assert ck_o.value == ck_r.value, "bmpSearch checksums differ"
    po = PatternSet(patterns, "oracle")
    pr = PatternSet(patterns, "ref")
    t_o_ac = Lb.orc_calib_ac(po.h, asc.ctypes.data, n, L, C.byref(ck_o))
    t_r_ac = R.ref_calib_acism(pr.h, asc.ctypes.data, n, L, C.byref(ck_r))
    assert ck_o.value == ck_r.value, "first-match checksums differ"
    po.close()
    pr.close()
    return dict(bmp_ratio=round(t_r_bmp / t_o_bmp, 3), ac_ratio=round(t_r_ac / t_o_ac, 3), reads=int(n), read_len=int(L),
                patterns=len(patterns), ref_bmp_s=round(t_r_bmp, 4), oracle_bmp_s=round(t_o_bmp, 4),
                ref_acism_s=round(t_r_ac, 4), oracle_ac_s=round(t_o_ac, 4))


# ---- the stage behind the hot path: findConsensusDRs ----
KSW_XSTART, KSW_XSUBO = 0x80000, 0x40000
ALIGNER_MAT = np.array([1, -3, -3, -3, 0, -3, 1, -3, -3, 0, -3, -3, 1, -3, 0, -3, -3, -3, 1, 0, 0, 0, 0, 0, 0], np.int8)   # Aligner.h:124-131
ALIGNER_XTRA = KSW_XSTART | KSW_XSUBO | 5


def ksw_align(query, target, impl="oracle", mat=ALIGNER_MAT, gapo=5, gape=2, xtra=ALIGNER_XTRA):
    """query / target: sequences of codes 0..4 -> (score, te, qe, tb, qb) as Aligner::getOffsetAgainstMaster sees them"""
    q = np.array(list(query), np.uint8)
    t = np.array(list(target), np.uint8)
    out = [C.c_int() for _ in range(5)]
    fn = lib().orc_ksw_align if impl == "oracle" else ref().ref_ksw_align
    fn(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, mat.ctypes.data, gapo, gape, xtra, *[C.byref(o) for o in out])
    assert q.tolist() == list(query) and t.tolist() == list(target)        # restored
    return tuple(o.value for o in out)


def smith_waterman(a, b, start, length, similarity=0.85):
    o = [C.c_int() for _ in range(6)]
    r = lib().orc_smith_waterman(a, len(a), b, len(b), C.byref(o[0]), C.byref(o[1]), start, length, similarity,
                                 C.byref(o[2]), C.byref(o[3]), C.byref(o[4]), C.byref(o[5]))
    return r, o[0].value, o[1].value, a[o[2].value:o[2].value + o[3].value], b[o[4].value:o[4].value + o[5].value]


class ConsResult:
    """numpy / python copies of an orc_cons_view (same field names as crass_amd's ConsensusResult)"""

    def __init__(self, v):
        def arr(ptr, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(int(cnt),)).astype(dt, copy=True)
        self.error, self.next_free_gid, self.n_tokens = int(v.error), int(v.next_free_gid), int(v.n_tokens)
        off = arr(v.tok_off, v.n_tokens + 1, np.uint64)
        tc = C.string_at(v.tok_chars, int(off[-1])) if v.n_tokens else b""
        self.tokens = [tc[int(off[i]):int(off[i + 1])] for i in range(v.n_tokens)]
        ng = int(v.n_groups)
        self.gids = arr(v.grp_gid, ng, np.int32).tolist()
        doff = arr(v.dr_off, ng + 1, np.uint64)
        dc = C.string_at(v.dr_chars, int(doff[-1])) if ng else b""
        self.true_drs = [dc[int(doff[i]):int(doff[i + 1])] for i in range(ng)]
        goff = arr(v.grp_off, ng + 1, np.uint64)
        gt = arr(v.grp_tokens, int(goff[-1]) if ng else 0, np.uint32)
        self.groups = [gt[int(goff[i]):int(goff[i + 1])].tolist() for i in range(ng)]
        n = int(v.n_rec)
        self.rec_alive = arr(v.rec_alive, n, np.uint8)
        self.rec_rc = arr(v.rec_rc, n, np.uint8)
        self.rec_token = arr(v.rec_token, n, np.uint32)
        self.rec_nss = arr(v.rec_nss, n, np.uint32)
        self.rec_ss_off = arr(v.rec_ss_off, n, np.uint64)
        self.ss_pool = arr(v.ss_pool, int(self.rec_nss.sum()), np.uint32)
        toff = arr(v.tokread_off, v.n_tokens + 1, np.uint64)
        tidx = arr(v.tokread_idx, int(toff[-1]) if v.n_tokens else 0, np.uint64)
        has = arr(v.tok_has_list, v.n_tokens, np.uint8)
        self.reads_of = [tidx[int(toff[i]):int(toff[i + 1])].tolist() if has[i] else None for i in range(v.n_tokens)]

    def ss(self, k):
        o = int(self.rec_ss_off[k])
        return self.ss_pool[o:o + int(self.rec_nss[k])].tolist()

    def group_read_counts(self):
        return [sum(len(self.reads_of[t - 2] or []) for t in g) for g in self.groups]


def consensus(seqs, res, params=None):
    """findConsensusDRs over a search result (an orc.PipelineResult or crass_amd's PipelineResult: same field names).
    seqs: list[bytes] or (uint8 array, uint64 offsets)."""
    p = params or Params.default()
    if isinstance(seqs, (list, tuple)) and (not seqs or isinstance(seqs[0], (bytes, bytearray))):
        sbuf, soff = concat(list(seqs))
    else:
        sbuf, soff = seqs
    n = res.n_pass1 + res.n_pass2
    keep = []

    def a(x, dt):
        y = np.ascontiguousarray(np.asarray(x)[:n] if len(x) >= n else x, dtype=dt)
        keep.append(y)
        return y.ctypes.data
    tbuf, toff = concat(list(res.tokens))
    goff = np.zeros(len(res.groups) + 1, np.uint64)
    goff[1:] = np.cumsum([len(g) for g in res.groups], dtype=np.uint64)
    gt = np.array([t for g in res.groups for t in g], np.uint32)
    ssp = np.ascontiguousarray(res.ss_pool, np.uint32)
    i = ConsInput(sbuf.ctypes.data, soff.ctypes.data, len(soff) - 1, n, a(res.rec_read, np.uint64), a(res.rec_lowlexi, np.uint8),
                  a(res.rec_token, np.uint32), a(res.rec_nss, np.uint32), a(res.rec_ss_off, np.uint64), ssp.ctypes.data,
                  len(res.tokens), tbuf.ctypes.data, toff.ctypes.data, len(res.groups), gt.ctypes.data, goff.ctypes.data,
                  int(res.max_read_len))
    h = lib().orc_consensus_run(C.byref(i), C.byref(p))
    v = ConsView()
    lib().orc_consensus_view(h, C.byref(v))
    out = ConsResult(v)
    lib().orc_consensus_free(h)
    return out


RC_TABLE = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def graph_groups(recs, ref, con):
    """the hand-off of the consensus stage as crass_build_outputs / oracle/crass_graph.py take it: [(gid, true DR, [(header,
    comment, RH_Seq, start_stops), ...])] in ascending GID, reads in WorkHorse::buildGraph's order (WorkHorse.cpp:454-505).
    recs: tests/fastx records (name, comment, seq, qual); ref: pipeline result; con: consensus result of this module."""
    groups = []
    for gid, dr, toks in zip(con.gids, con.true_drs, con.groups):
        reads = []
        for t in toks:
            for k in (con.reads_of[t - 2] or []):
                i = int(ref.rec_read[k])
                s = recs[i][2]
                if int(con.rec_rc[k]):
                    s = s.translate(RC_TABLE)[::-1]
                reads.append((recs[i][0], recs[i][1], s, con.ss(k)))
        groups.append((int(gid), dr, reads))
    return groups
