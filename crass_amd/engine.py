"""Python host layer over the C ABI (include/crass_hip.h).

Used by the tests, bench.py and the multi-GPU driver (torch.distributed is plumbing only).
The compiled C++ adapter with the reference's own function shapes (searchFile /
createNonRedundantSet / findSingletons / addReadHolder) lives in csrc/adapter/ — see
INTEGRATION.md.  Nothing here computes search results on the CPU.
"""
import ctypes as C

import numpy as np

from . import _abi


class CrassError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = _abi.load().crass_hip_strerror(status).decode()
        super().__init__("%s: %s (status %d)" % (where or "crass_hip", msg, status))


def _chk(status, where):
    if status != 0:
        raise CrassError(status, where)


def default_params(**kw):
    p = _abi.Params()
    _abi.load().crass_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).view(dtype).copy()


def _npv(addr, n, dtype):
    """n elements of dtype at the raw address addr (a c_void_p field) as a numpy copy"""
    if n == 0 or not addr:
        return np.zeros(0, dtype)
    ct = {np.uint8: C.c_uint8, np.uint32: C.c_uint32, np.uint64: C.c_uint64}[dtype]
    return np.ctypeslib.as_array(C.cast(C.c_void_p(addr), C.POINTER(ct)), shape=(int(n),)).view(dtype).copy()


def concat(items):
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        off[1:] = np.cumsum([len(s) for s in items], dtype=np.uint64)
    buf = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if items else np.zeros(0, np.uint8)
    return buf, off


class PackedReads:
    """2-bit packed reads + exception list produced by the C++ packer (crass_pack_reads)."""

    def __init__(self, seqs, pad_uniform=False):
        lib = _abi.load()
        if isinstance(seqs, (list, tuple)) and (not seqs or isinstance(seqs[0], (bytes, bytearray))):
            buf, off = concat(list(seqs))
        else:
            buf, off = seqs
        self._src = (buf, off)
        self.p = _abi.Packed()
        _chk(lib.crass_pack_reads(buf.ctypes.data, off.ctypes.data, len(off) - 1, int(pad_uniform), C.byref(self.p)),
             "crass_pack_reads")
        self.header_id = None

    @property
    def reads(self):
        return self.p.reads

    @property
    def n_reads(self):
        return int(self.p.reads.n_reads)

    def packed_array(self):
        r = self.p.reads
        if r.stride_words:
            n = int(r.n_reads) * int(r.stride_words)
        else:
            raise ValueError("ragged layout")
        return np.ctypeslib.as_array(C.cast(r.packed, _abi.u32p), shape=(n,))

    def close(self):
        if self.p.owner:
            _abi.load().crass_free_packed(C.byref(self.p))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastxFile:
    """FASTA/FASTQ(.gz) records with kseq_read semantics (C++ reader, crass_read_fastx)."""

    def __init__(self, path):
        lib = _abi.load()
        f = _abi.Fastx()
        _chk(lib.crass_read_fastx(str(path).encode(), C.byref(f)), "crass_read_fastx(%s)" % path)
        n = int(f.n_reads)
        self.n_reads = n
        self.max_len = int(f.max_len)
        self.last_ret = int(f.last_ret)
        self.seq_off = _np(f.seq_off, n + 1, np.uint64)
        self.seq = _np(f.seq, int(self.seq_off[-1]), np.uint8)
        self.name_off = _np(f.name_off, n + 1, np.uint64)
        self.name = _np(f.name, int(self.name_off[-1]), np.uint8)
        self.comment_off = _np(f.comment_off, n + 1, np.uint64)
        self.comment = _np(f.comment, int(self.comment_off[-1]), np.uint8)
        self.has_comment = _np(f.has_comment, n, np.uint8)
        self.qual_off = _np(f.qual_off, n + 1, np.uint64)
        self.qual = _np(f.qual, int(self.qual_off[-1]), np.uint8)
        self.has_qual = _np(f.has_qual, n, np.uint8)
        self.header_id = _np(f.header_id, n, np.uint64)
        lib.crass_free_fastx(C.byref(f))

    def _field(self, buf, off, i):
        return buf[int(off[i]):int(off[i + 1])].tobytes()

    def record(self, i):
        return (self._field(self.name, self.name_off, i),
                self._field(self.comment, self.comment_off, i) if self.has_comment[i] else None,
                self._field(self.seq, self.seq_off, i),
                self._field(self.qual, self.qual_off, i) if self.has_qual[i] else None)

    def records(self):
        return [self.record(i) for i in range(self.n_reads)]

    def unique_headers(self):
        return bool(np.all(self.header_id == np.arange(self.n_reads, dtype=np.uint64)))


class FastxIndex:
    """crass_index_fastx(_files): the input(s) kept mapped (plain text) or inflated once (gzip), the reads 2-bit packed at once, the
    records' text parsed on request; a list of paths = one read set in (file, read) order with header ids across the files.
    Raises CrassError(status 2, unsupported) for files that mix records with / without comment or quality."""

    def __init__(self, path):
        self.lib = _abi.load()
        h = C.c_void_p()
        if isinstance(path, (list, tuple)):
            arr = (C.c_char_p * len(path))(*[str(p).encode() for p in path])
            _chk(self.lib.crass_index_fastx_files(arr, len(path), C.byref(h)), "crass_index_fastx_files(%s)" % (path,))
        else:
            _chk(self.lib.crass_index_fastx(str(path).encode(), C.byref(h)), "crass_index_fastx(%s)" % path)
        self.h = h
        self.reads = _abi.Reads()
        ml, lr = C.c_uint32(), C.c_int()
        _chk(self.lib.crass_fastx_index_reads(self.h, C.byref(self.reads), C.byref(ml), C.byref(lr)), "crass_fastx_index_reads")
        self.max_len, self.last_ret, self.n_reads = int(ml.value), int(lr.value), int(self.reads.n_reads)

    def layout(self):
        """the packed reads as python objects: dict(stride, uniform_len, words per read (list of arrays), lengths, exceptions, header_id)"""
        r = self.reads
        n = self.n_reads
        stride, uni = int(r.stride_words), int(r.uniform_len)
        lengths = np.full(n, uni, np.uint32) if uni else _npv(r.lengths, n, np.uint32).copy()
        if stride:
            words = _npv(r.packed, n * stride, np.uint32).reshape(n, stride).copy() if n else np.zeros((0, stride), np.uint32)
            per = [words[i, :(int(lengths[i]) + 15) // 16] for i in range(n)]
        else:
            off = _npv(r.word_off, n + 1, np.uint64)
            allw = _npv(r.packed, int(off[n]), np.uint32).copy() if n else np.zeros(0, np.uint32)
            per = [allw[int(off[i]):int(off[i]) + (int(lengths[i]) + 15) // 16] for i in range(n)]
        ne = int(r.n_exceptions)
        exc = {}
        if ne:
            er, eo = _npv(r.exc_read, ne, np.uint64), _npv(r.exc_off, ne + 1, np.uint64)
            eb = _npv(r.exc_bytes, int(eo[ne]), np.uint8)
            exc = {int(er[k]): eb[int(eo[k]):int(eo[k + 1])].tobytes() for k in range(ne)}
        hid = _npv(r.header_id, n, np.uint64).tolist() if r.header_id else list(range(n))
        return dict(stride=stride, uniform_len=uni, words=per, lengths=lengths.tolist(), exceptions=exc, header_id=hid)

    def fetch(self, idx):
        """records (name, comment | None, seq, qual | None) of the read indices idx, in that order"""
        a = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64))
        f = _abi.Fastx()
        _chk(self.lib.crass_fastx_index_fetch(self.h, a.ctypes.data, len(a), C.byref(f)), "crass_fastx_index_fetch")
        n = int(f.n_reads)
        so, no, co, qo = (_np(x, n + 1, np.uint64) for x in (f.seq_off, f.name_off, f.comment_off, f.qual_off))
        sq, nm = _np(f.seq, int(so[-1]), np.uint8), _np(f.name, int(no[-1]), np.uint8)
        cm, ql = _np(f.comment, int(co[-1]), np.uint8), _np(f.qual, int(qo[-1]), np.uint8)
        hc, hq = _np(f.has_comment, n, np.uint8), _np(f.has_qual, n, np.uint8)
        out = [(nm[int(no[i]):int(no[i + 1])].tobytes(), cm[int(co[i]):int(co[i + 1])].tobytes() if hc[i] else None,
                sq[int(so[i]):int(so[i + 1])].tobytes(), ql[int(qo[i]):int(qo[i + 1])].tobytes() if hq[i] else None) for i in range(n)]
        self.lib.crass_free_fastx(C.byref(f))
        return out

    def drop_text(self):
        """give the plain-text inputs' mappings back (crass_fastx_index_drop_text); later fetches read from the files"""
        self.lib.crass_fastx_index_drop_text(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.crass_fastx_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_fastx(path, chunk_bytes=0, with_names=True):
    """The records of a FASTA/FASTQ(.gz) file through the chunked reader (crass_fastx_stream_*): a list of (name, comment, seq, qual)
    like FastxFile.records(), the header ids (job-level index of the first read with the same name), kseq_read's final return value
    and the number of chunks it took."""
    lib = _abi.load()
    tab = C.c_void_p(lib.crass_name_table_create()) if with_names else C.c_void_p()
    h = C.c_void_p()
    _chk(lib.crass_fastx_stream_open(str(path).encode(), int(chunk_bytes), tab, 0, C.byref(h)), "crass_fastx_stream_open(%s)" % path)
    recs, hid, last, chunks = [], [], -1, 0
    try:
        while True:
            f = _abi.Fastx()
            _chk(lib.crass_fastx_stream_next(h, C.byref(f)), "crass_fastx_stream_next")
            last = int(f.last_ret)
            n = int(f.n_reads)
            if n == 0:
                break
            chunks += 1
            so, no = _np(f.seq_off, n + 1, np.uint64), _np(f.name_off, n + 1, np.uint64)
            co, qo = _np(f.comment_off, n + 1, np.uint64), _np(f.qual_off, n + 1, np.uint64)
            sq, nm = _np(f.seq, int(so[-1]), np.uint8), _np(f.name, int(no[-1]), np.uint8)
            cm, ql = _np(f.comment, int(co[-1]), np.uint8), _np(f.qual, int(qo[-1]), np.uint8)
            hc, hq = _np(f.has_comment, n, np.uint8), _np(f.has_qual, n, np.uint8)
            hid.extend(_np(f.header_id, n, np.uint64).tolist())
            for i in range(n):
                recs.append((nm[int(no[i]):int(no[i + 1])].tobytes(), cm[int(co[i]):int(co[i + 1])].tobytes() if hc[i] else None,
                             sq[int(so[i]):int(so[i + 1])].tobytes(), ql[int(qo[i]):int(qo[i + 1])].tobytes() if hq[i] else None))
        assert int(lib.crass_fastx_stream_reads_done(h)) == len(recs)
    finally:
        lib.crass_fastx_stream_close(h)
        if with_names:
            lib.crass_name_table_destroy(tab)
    return recs, hid, last, chunks


def synth_spec(**kw):
    s = _abi.SynthSpec()
    _abi.load().crass_synth_default(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def synth_packed(spec, first_read, n_reads, out=None, n_threads=0):
    """Deterministic synthetic reads [first_read, first_read+n_reads) as packed uint32 words
    (uniform stride ceil(L/16)).  `out` may be a preallocated (e.g. pinned) uint32 array."""
    W = (spec.read_len + 15) // 16
    if out is None:
        out = np.empty(int(n_reads) * W, dtype=np.uint32)
    assert out.dtype == np.uint32 and out.size >= int(n_reads) * W
    _chk(_abi.load().crass_synth_packed(C.byref(spec), int(first_read), int(n_reads), out.ctypes.data, int(n_threads)),
         "crass_synth_packed")
    return out


def unpack_ascii(packed, stride_words, read_len, n_reads):
    out = np.empty(int(n_reads) * int(read_len), dtype=np.uint8)
    _chk(_abi.load().crass_unpack_ascii(packed.ctypes.data, int(stride_words), int(read_len), int(n_reads),
                                        out.ctypes.data), "crass_unpack_ascii")
    return out


class CandidateSet:
    def __init__(self, v):
        n = int(v.n)
        self.n = n
        self.read_idx = _np(v.read_idx, n, np.uint64)
        self.low_lexi = _np(v.low_lexi, n, np.uint8)
        self.repeat_len = _np(v.repeat_len, n, np.uint32)
        self.n_ss = _np(v.n_ss, n, np.uint32)
        self.ss_off = _np(v.ss_off, n, np.uint64)
        # the pool may be fixed-stride slots (ss_off[k] = k*cap) or tightly packed: size it from the offsets
        pool_len = int((self.ss_off + self.n_ss).max()) if n else 0
        self.ss_pool = _np(v.ss_pool, pool_len, np.uint32)
        self.dr_stride = int(v.dr_stride)
        self.dr_len = _np(v.dr_len, n, np.uint16)
        self.dr_chars = _np(C.cast(v.dr_chars, _abi.u8p), n * self.dr_stride, np.uint8)
        self.max_read_len = int(v.max_read_len)

    def ss(self, k):
        o = int(self.ss_off[k])
        return self.ss_pool[o:o + int(self.n_ss[k])].tolist()

    def dr(self, k):
        o = k * self.dr_stride
        return self.dr_chars[o:o + int(self.dr_len[k])].tobytes()


class MergeResult:
    def __init__(self, v):
        self.n_tokens = int(v.n_tokens)
        self.tok_off = _np(v.tok_off, self.n_tokens + 1, np.uint64)
        tc = C.string_at(v.tok_chars, int(self.tok_off[-1])) if self.n_tokens else b""
        self.tokens = [tc[int(self.tok_off[i]):int(self.tok_off[i + 1])] for i in range(self.n_tokens)]
        self.cand_token = _np(v.cand_token, int(v.n_candidates), np.uint32)
        self.n_groups = int(v.n_groups)
        self.grp_off = _np(v.grp_off, self.n_groups + 1, np.uint64)
        gt = _np(v.grp_tokens, int(self.grp_off[-1]) if self.n_groups else 0, np.uint32)
        self.groups = [gt[int(self.grp_off[i]):int(self.grp_off[i + 1])].tolist() for i in range(self.n_groups)]
        self.n_patterns = int(v.n_patterns)
        self.pat_off = _np(v.pat_off, self.n_patterns + 1, np.uint64)
        pc = C.string_at(v.pat_chars, int(self.pat_off[-1])) if self.n_patterns else b""
        self.patterns = [pc[int(self.pat_off[i]):int(self.pat_off[i + 1])] for i in range(self.n_patterns)]
        self.pat_group = _np(v.pat_group, self.n_patterns, np.uint32)
        self.next_free_gid = int(v.next_free_gid)


def merge_host(dr_chars, dr_len, kmer_clust_size=6):
    """createNonRedundantSet on the host without a GPU context (crass_merge_create).
    dr_chars: uint8 [n, stride]; dr_len: uint16 [n]."""
    lib = _abi.load()
    dr_chars = np.ascontiguousarray(dr_chars, dtype=np.uint8)
    dr_len = np.ascontiguousarray(dr_len, dtype=np.uint16)
    n = dr_chars.shape[0]
    stride = dr_chars.shape[1] if dr_chars.ndim == 2 and n else 16
    h = C.c_void_p()
    _chk(lib.crass_merge_create(dr_chars.ctypes.data, dr_len.ctypes.data, int(stride), int(n), int(kmer_clust_size),
                                C.byref(h)), "crass_merge_create")
    try:
        v = _abi.MergeView()
        _chk(lib.crass_merge_get(h, C.byref(v)), "crass_merge_get")
        return MergeResult(v)
    finally:
        lib.crass_merge_destroy(h)


def merge_rebuild(dx_chars, dx_len, cand_distinct, gid_of, dropped, n_groups):
    """host view of a merge from per-token results (crass_merge_rebuild): what the engine does after the device
    merge, callable without a GPU.  dx_chars: uint8 [n_distinct, stride] in token order."""
    lib = _abi.load()
    dx_chars = np.ascontiguousarray(dx_chars, dtype=np.uint8)
    dx_len = np.ascontiguousarray(dx_len, dtype=np.uint16)
    cand = np.ascontiguousarray(cand_distinct, dtype=np.uint32)
    gid = np.ascontiguousarray(gid_of, dtype=np.uint32)
    drop = np.ascontiguousarray(dropped, dtype=np.uint8)
    nd = dx_chars.shape[0]
    stride = dx_chars.shape[1] if nd else 16
    h = C.c_void_p()
    _chk(lib.crass_merge_rebuild(dx_chars.ctypes.data, dx_len.ctypes.data, int(stride), int(nd), cand.ctypes.data, int(len(cand)),
                                 gid.ctypes.data, drop.ctypes.data, int(n_groups), C.byref(h)), "crass_merge_rebuild")
    try:
        v = _abi.MergeView()
        _chk(lib.crass_merge_get(h, C.byref(v)), "crass_merge_get")
        return MergeResult(v)
    finally:
        lib.crass_merge_destroy(h)


def dr_slots(strings, stride=48):
    """list[bytes] -> (uint8 [n, stride], uint16 [n]) in the C ABI's fixed-slot layout"""
    n = len(strings)
    chars = np.zeros((n, stride), np.uint8)
    lens = np.zeros(n, np.uint16)
    for i, s in enumerate(strings):
        chars[i, :len(s)] = np.frombuffer(s, np.uint8)
        lens[i] = len(s)
    return chars, lens


class RecruitSet:
    def __init__(self, v):
        n = int(v.n)
        self.n = n
        self.read_idx = _np(v.read_idx, n, np.uint64)
        self.low_lexi = _np(v.low_lexi, n, np.uint8)
        self.start = _np(v.start, n, np.uint32)
        self.end = _np(v.end, n, np.uint32)
        self.dr_stride = int(v.dr_stride)
        self.dr_len = _np(v.dr_len, n, np.uint16)
        self.dr_chars = _np(C.cast(v.dr_chars, _abi.u8p), n * self.dr_stride, np.uint8)
        self.token = _np(v.token, n, np.uint32)

    def dr(self, k):
        o = k * self.dr_stride
        return self.dr_chars[o:o + int(self.dr_len[k])].tobytes()


class SearchEngine:
    """One context per GPU (crass_hip_create).  Call order mirrors WorkHorse::parseSeqFiles
    (WorkHorse.cpp:321-414): load_reads -> seed_scan (searchFile) -> merge
    (createNonRedundantSet) -> recruit (findSingletons)."""

    def __init__(self, params=None, device=0):
        self.lib = _abi.load()
        self.params = params or default_params()
        h = C.c_void_p()
        _chk(self.lib.crass_hip_create(C.byref(self.params), int(device), C.byref(h)), "crass_hip_create")
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.crass_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- reads ----
    def load_reads(self, packed, header_id=None, read_index_base=0):
        r = _abi.Reads()
        C.memmove(C.byref(r), C.byref(packed.reads), C.sizeof(r))
        hid = None
        if header_id is not None:
            hid = np.ascontiguousarray(header_id, dtype=np.uint64)
            r.header_id = hid.ctypes.data
        r.read_index_base = int(read_index_base)
        _chk(self.lib.crass_hip_load_reads(self.h, C.byref(r)), "crass_hip_load_reads")
        self._keep = (packed, hid)

    def load_packed_uniform(self, words, n_reads, read_len, read_index_base=0):
        """Host uint32 array of uniform-stride packed reads (synthetic generator output)."""
        r = _abi.Reads()
        r.n_reads = int(n_reads)
        r.packed = words.ctypes.data
        r.stride_words = (int(read_len) + 15) // 16
        r.uniform_len = int(read_len)
        r.read_index_base = int(read_index_base)
        _chk(self.lib.crass_hip_load_reads(self.h, C.byref(r)), "crass_hip_load_reads")

    def attach_device_tensor(self, tensor, n_reads, read_len, read_index_base=0):
        """Zero-copy: a torch int32 CUDA tensor holding uniform-stride packed reads."""
        r = _abi.Reads()
        r.n_reads = int(n_reads)
        r.packed = int(tensor.data_ptr())
        r.stride_words = (int(read_len) + 15) // 16
        r.uniform_len = int(read_len)
        r.read_index_base = int(read_index_base)
        _chk(self.lib.crass_hip_attach_device_reads(self.h, C.byref(r)), "crass_hip_attach_device_reads")
        self._keep = tensor

    # ---- passes ----
    def seed_scan(self, fetch=True):
        _chk(self.lib.crass_hip_seed_scan(self.h), "crass_hip_seed_scan")
        return self.candidates() if fetch else None

    def candidates(self):
        v = _abi.Candidates()
        _chk(self.lib.crass_hip_get_candidates(self.h, C.byref(v)), "crass_hip_get_candidates")
        return CandidateSet(v)

    def candidate_dr_view(self):
        """(uint8 array [n, dr_stride], uint16 lengths) without copying the other fields."""
        v = _abi.Candidates()
        _chk(self.lib.crass_hip_get_candidates(self.h, C.byref(v)), "crass_hip_get_candidates")
        n = int(v.n)
        chars = _np(C.cast(v.dr_chars, _abi.u8p), n * int(v.dr_stride), np.uint8).reshape(n, int(v.dr_stride))
        return chars, _np(v.dr_len, n, np.uint16)

    def merge(self, dr_chars=None, dr_len=None, fetch=True):
        """dr_chars: uint8 array [n, stride] of ALL candidates in global read order (multi-GPU),
        or None for this context's own candidates."""
        if dr_chars is None:
            _chk(self.lib.crass_hip_merge(self.h, None, None, 0, 0), "crass_hip_merge")
        else:
            dr_chars = np.ascontiguousarray(dr_chars, dtype=np.uint8)
            dr_len = np.ascontiguousarray(dr_len, dtype=np.uint16)
            n = dr_chars.shape[0]
            stride = dr_chars.shape[1] if n else 16
            _chk(self.lib.crass_hip_merge(self.h, dr_chars.ctypes.data, dr_len.ctypes.data, int(stride), int(n)),
                 "crass_hip_merge")
        return self.merge_view() if fetch else None

    def distinct(self):
        """(uint8 [n_distinct, stride], uint16 [n_distinct], uint32 cand->distinct map) of this
        context's pass-1 candidates, first-occurrence order (the compact multi-GPU payload)."""
        v = _abi.Distinct()
        _chk(self.lib.crass_hip_get_distinct(self.h, C.byref(v)), "crass_hip_get_distinct")
        nd, st = int(v.n_distinct), int(v.dr_stride)
        chars = _np(C.cast(v.dr_chars, _abi.u8p), nd * st, np.uint8).reshape(nd, st)
        return chars, _np(v.dr_len, nd, np.uint16), _np(v.cand_distinct, int(v.n_candidates), np.uint32)

    def merge_distinct(self, g_chars, g_lens, my_offset, fetch=True):
        """merge from the rank-ordered concatenation of every rank's distinct list"""
        g_chars = np.ascontiguousarray(g_chars, dtype=np.uint8)
        g_lens = np.ascontiguousarray(g_lens, dtype=np.uint16)
        n = g_chars.shape[0]
        stride = g_chars.shape[1] if n else 16
        _chk(self.lib.crass_hip_merge_distinct(self.h, g_chars.ctypes.data, g_lens.ctypes.data, int(stride), int(n),
                                               int(my_offset)), "crass_hip_merge_distinct")
        return self.merge_view() if fetch else None

    def distinct_device(self):
        """(device pointer of the distinct strings, device pointer of their lengths, n, stride) or None when
        pass 1 did not leave the list on the device"""
        v = _abi.DistinctDev()
        st = self.lib.crass_hip_get_distinct_device(self.h, C.byref(v))
        if st != 0:
            return None
        return int(v.d_chars or 0), int(v.d_len or 0), int(v.n_distinct), int(v.dr_stride)

    def merge_distinct_device(self, d_chars_ptr, d_len_ptr, stride, n_global, my_offset, fetch=True):
        """merge from a DEVICE-resident rank-ordered concatenation of every rank's distinct list"""
        _chk(self.lib.crass_hip_merge_distinct_device(self.h, C.c_void_p(d_chars_ptr), C.c_void_p(d_len_ptr), int(stride), int(n_global),
                                                      int(my_offset)), "crass_hip_merge_distinct_device")
        return self.merge_view() if fetch else None

    def exchange_setup(self, world, rank, cap_rows):
        """one-collective exchange: returns (device pointer of the send buffer, its size in bytes)"""
        x = _abi.Exchange()
        _chk(self.lib.crass_hip_exchange_setup(self.h, int(world), int(rank), int(cap_rows), C.byref(x)), "crass_hip_exchange_setup")
        return int(x.d_send), int(x.send_bytes)

    def exchange_set_deferred(self, on=True):
        """seed_scan() returns with pass 1 still queued; the collective goes on the engine's stream (crass_hip_exchange_set_deferred)"""
        _chk(self.lib.crass_hip_exchange_set_deferred(self.h, 1 if on else 0), "crass_hip_exchange_set_deferred")

    def merge_gathered(self, d_recv_ptr, fetch=True):
        """merge from the all-gathered send buffers (device pointer).  Returns None, or the number of rows the
        exchange needs when some rank's list did not fit (set up again, repeat seed scan + collective)."""
        st = self.lib.crass_hip_merge_gathered(self.h, C.c_void_p(d_recv_ptr))
        if st == _abi.ERR_OVERFLOW:
            return int(self.lib.crass_hip_exchange_needed_rows(self.h))
        _chk(st, "crass_hip_merge_gathered")
        return self.merge_view() if fetch else None

    def merge_view(self):
        v = _abi.MergeView()
        _chk(self.lib.crass_hip_get_merge(self.h, C.byref(v)), "crass_hip_get_merge")
        return MergeResult(v)

    def set_patterns(self, patterns):
        arr = (C.c_char_p * len(patterns))(*patterns)
        lens = (C.c_uint32 * len(patterns))(*[len(p) for p in patterns])
        _chk(self.lib.crass_hip_set_patterns(self.h, arr, lens, len(patterns)), "crass_hip_set_patterns")

    def recruit(self, extra_found=None, fetch=True):
        if extra_found is not None and len(extra_found):
            ef = np.ascontiguousarray(extra_found, dtype=np.uint64)
            _chk(self.lib.crass_hip_recruit(self.h, ef.ctypes.data, len(ef)), "crass_hip_recruit")
        else:
            _chk(self.lib.crass_hip_recruit(self.h, None, 0), "crass_hip_recruit")
        return self.recruits() if fetch else None

    def recruits(self):
        v = _abi.Recruits()
        _chk(self.lib.crass_hip_get_recruits(self.h, C.byref(v)), "crass_hip_get_recruits")
        return RecruitSet(v)

    def fetch_abi(self):
        """the three getters of the ABI (crass_hip_get_candidates / _get_merge / _get_recruits) and nothing else: what an adapter
        pays on top of a step to see the wide per-record arrays (the step itself ends with compact blobs in pinned memory).
        Returns (n_candidates, n_tokens, n_recruits)."""
        c, m, q = _abi.Candidates(), _abi.MergeView(), _abi.Recruits()
        _chk(self.lib.crass_hip_get_candidates(self.h, C.byref(c)), "crass_hip_get_candidates")
        _chk(self.lib.crass_hip_get_merge(self.h, C.byref(m)), "crass_hip_get_merge")
        _chk(self.lib.crass_hip_get_recruits(self.h, C.byref(q)), "crass_hip_get_recruits")
        return int(c.n), int(m.n_tokens), int(q.n)

    def stream_wait_event(self, event_handle):
        """order the engine's stream behind a HIP event (raw hipEvent_t handle, e.g. torch.cuda.Event().cuda_event)"""
        _chk(self.lib.crass_hip_stream_wait_event(self.h, C.c_void_p(int(event_handle))), "crass_hip_stream_wait_event")

    def set_host_view(self, light):
        """1: after a device merge this context only builds its own candidates' tokens (ranks other than 0 of a multi-rank job:
        tokens, groups and patterns are identical on every rank); 0: the full host view (crass_hip_set_host_view)"""
        _chk(self.lib.crass_hip_set_host_view(self.h, 1 if light else 0), "crass_hip_set_host_view")

    def set_timing_focus(self, kernels):
        """level 1: bit 0 seed scan, bit 1 survivors, bit 2 pass-2 scan (crass_hip_set_timing_focus)"""
        _chk(self.lib.crass_hip_set_timing_focus(self.h, int(kernels)), "crass_hip_set_timing_focus")

    def set_stage_timing(self, level):
        """0 none (default), 1 the three large kernels, 2 every stage — see crass_hip_set_stage_timing."""
        _chk(self.lib.crass_hip_set_stage_timing(self.h, int(level)), "crass_hip_set_stage_timing")

    def reload_env(self):
        """re-read the environment's A/B and test switches (they are read once, at creation)"""
        _chk(self.lib.crass_hip_reload_env(self.h), "crass_hip_reload_env")

    def counters(self):
        c = _abi.Counters()
        _chk(self.lib.crass_hip_get_counters(self.h, C.byref(c)), "crass_hip_get_counters")
        return c.asdict()

    def stream_handle(self):
        return self.lib.crass_hip_stream(self.h)

    def levenshtein_batch(self, pairs, want_similarity=True):
        """pairs: list of (bytes, bytes) -> (int32 distances, float32 similarities)"""
        items = []
        a_off, a_len, b_off, b_len = [], [], [], []
        pos = 0
        for a, b in pairs:
            a_off.append(pos); a_len.append(len(a)); items.append(a); pos += len(a)
            b_off.append(pos); b_len.append(len(b)); items.append(b); pos += len(b)
        chars = np.frombuffer(b"".join(items) + b"\0", dtype=np.uint8).copy()
        a_off = np.array(a_off, np.uint64); b_off = np.array(b_off, np.uint64)
        a_len = np.array(a_len, np.uint32); b_len = np.array(b_len, np.uint32)
        dist = np.zeros(len(pairs), np.int32)
        sim = np.zeros(len(pairs), np.float32)
        _chk(self.lib.crass_hip_levenshtein_batch(self.h, chars.ctypes.data, pos, a_off.ctypes.data, a_len.ctypes.data,
                                                  b_off.ctypes.data, b_len.ctypes.data, len(pairs), dist.ctypes.data,
                                                  sim.ctypes.data if want_similarity else None),
             "crass_hip_levenshtein_batch")
        return dist, sim


class SearchGroup:
    """Several GPUs from one process (crass_hip_group_*): contiguous read shards, one RCCL all-gather of the distinct
    candidate DR strings per step issued by the engine, one host view for the group.  devices: list of device
    indices; local_copies=True replaces the collective by device copies (tests: several contexts on one GPU)."""

    GROUP_LOCAL_COPIES = 1

    def __init__(self, devices, params=None, local_copies=False):
        self.lib = _abi.load()
        self.params = params or default_params()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        st = self.lib.crass_hip_group_create(C.byref(self.params), devs, len(devices), self.GROUP_LOCAL_COPIES if local_copies else 0,
                                             C.byref(h))
        if st != 0:
            raise CrassError(st, "crass_hip_group_create [%s]" % self.lib.crass_hip_group_last_error().decode())
        self.h = h
        self.n = len(devices)
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.crass_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st, where):
        if st != 0:
            raise CrassError(st, "%s [%s]" % (where, self.lib.crass_hip_group_last_error().decode()))

    @property
    def rccl_ranks(self):
        return int(self.lib.crass_hip_group_rccl_ranks(self.h))

    def load_reads(self, packed, header_id=None, read_index_base=0):
        r = _abi.Reads()
        C.memmove(C.byref(r), C.byref(packed.reads), C.sizeof(r))
        hid = None
        if header_id is not None:
            hid = np.ascontiguousarray(header_id, dtype=np.uint64)
            r.header_id = hid.ctypes.data
        r.read_index_base = int(read_index_base)
        self._chk(self.lib.crass_hip_group_load_reads(self.h, C.byref(r)), "crass_hip_group_load_reads")

    def load_packed_uniform(self, words, n_reads, read_len, read_index_base=0):
        r = _abi.Reads()
        r.n_reads = int(n_reads)
        r.packed = words.ctypes.data
        r.stride_words = (int(read_len) + 15) // 16
        r.uniform_len = int(read_len)
        r.read_index_base = int(read_index_base)
        self._chk(self.lib.crass_hip_group_load_reads(self.h, C.byref(r)), "crass_hip_group_load_reads")

    def seed_scan(self):
        self._chk(self.lib.crass_hip_group_seed_scan(self.h), "crass_hip_group_seed_scan")

    def merge(self):
        self._chk(self.lib.crass_hip_group_merge(self.h), "crass_hip_group_merge")

    def recruit(self, extra_found=None):
        if extra_found is not None and len(extra_found):
            ef = np.ascontiguousarray(extra_found, dtype=np.uint64)
            self._chk(self.lib.crass_hip_group_recruit(self.h, ef.ctypes.data, len(ef)), "crass_hip_group_recruit")
        else:
            self._chk(self.lib.crass_hip_group_recruit(self.h, None, 0), "crass_hip_group_recruit")

    def set_patterns(self, patterns):
        arr = (C.c_char_p * len(patterns))(*patterns)
        lens = (C.c_uint32 * len(patterns))(*[len(p) for p in patterns])
        self._chk(self.lib.crass_hip_group_set_patterns(self.h, arr, lens, len(patterns)), "crass_hip_group_set_patterns")

    def step(self):
        self._chk(self.lib.crass_hip_group_step(self.h), "crass_hip_group_step")

    def candidates(self):
        v = _abi.Candidates()
        self._chk(self.lib.crass_hip_group_get_candidates(self.h, C.byref(v)), "crass_hip_group_get_candidates")
        return CandidateSet(v)

    def merge_view(self):
        v = _abi.MergeView()
        self._chk(self.lib.crass_hip_group_get_merge(self.h, C.byref(v)), "crass_hip_group_get_merge")
        return MergeResult(v)

    def recruits(self):
        v = _abi.Recruits()
        self._chk(self.lib.crass_hip_group_get_recruits(self.h, C.byref(v)), "crass_hip_group_get_recruits")
        return RecruitSet(v)

    def rank_counters(self, rank):
        c = _abi.Counters()
        _chk(self.lib.crass_hip_get_counters(C.c_void_p(self.lib.crass_hip_group_ctx(self.h, int(rank))), C.byref(c)), "crass_hip_get_counters")
        return c.asdict()

    def rank_set_stage_timing(self, rank, level):
        _chk(self.lib.crass_hip_set_stage_timing(C.c_void_p(self.lib.crass_hip_group_ctx(self.h, int(rank))), int(level)), "crass_hip_set_stage_timing")

    def rank_set_timing_focus(self, rank, kernels):
        _chk(self.lib.crass_hip_set_timing_focus(C.c_void_p(self.lib.crass_hip_group_ctx(self.h, int(rank))), int(kernels)), "crass_hip_set_timing_focus")

    def result(self):
        """the whole job's hand-off as a PipelineResult (same fields as search_pipeline's)"""
        cand, merge, rec = self.candidates(), self.merge_view(), self.recruits()
        res = PipelineResult(cand, merge, rec, cand.max_read_len)
        res.counters = [self.rank_counters(r) for r in range(self.n)]
        return res


def search_pipeline_group(seqs, devices, headers=None, params=None, local_copies=False, pad_uniform=0, fused=True):
    """search_pipeline over a group of contexts (one per entry of `devices`)"""
    packed = PackedReads(seqs, pad_uniform)
    header_id = None
    if headers is not None:
        first = {}
        header_id = np.empty(len(headers), np.uint64)
        for i, h in enumerate(headers):
            header_id[i] = first.setdefault(h, i)
        if np.all(header_id == np.arange(len(headers), dtype=np.uint64)):
            header_id = None
    g = SearchGroup(devices, params, local_copies)
    try:
        g.load_reads(packed, header_id)
        if fused:
            g.step()
        else:
            g.seed_scan(); g.merge(); g.recruit()
        return g.result()
    finally:
        g.close()
        packed.close()


class PipelineResult:
    """Same field names as tests/orc.PipelineResult so parity tests compare attribute by attribute."""

    def __init__(self, cand, merge, rec, max_read_len):
        self.n_pass1, self.n_pass2 = cand.n, rec.n
        self.n_tokens, self.n_groups, self.n_patterns = merge.n_tokens, merge.n_groups, merge.n_patterns
        self.max_read_len = max_read_len
        self.error = 0
        self.rec_read = np.concatenate([cand.read_idx, rec.read_idx])
        self.rec_lowlexi = np.concatenate([cand.low_lexi, rec.low_lexi])
        self.rec_token = np.concatenate([merge.cand_token[:cand.n] if len(merge.cand_token) == cand.n
                                         else merge.cand_token, rec.token]).astype(np.uint32)
        self.rec_replen = np.concatenate([cand.repeat_len, np.zeros(rec.n, np.uint32)])
        self.rec_nss = np.concatenate([cand.n_ss, np.full(rec.n, 2, np.uint32)])
        rec_ss = np.stack([rec.start, rec.end], axis=1).reshape(-1) if rec.n else np.zeros(0, np.uint32)
        self.ss_pool = np.concatenate([cand.ss_pool, rec_ss]).astype(np.uint32)
        off1 = cand.ss_off
        base = len(cand.ss_pool)
        off2 = base + 2 * np.arange(rec.n, dtype=np.uint64)
        self.rec_ss_off = np.concatenate([off1, off2]).astype(np.uint64)
        self.tokens, self.groups, self.patterns, self.pat_group = merge.tokens, merge.groups, merge.patterns, merge.pat_group
        self.cand, self.merge, self.rec = cand, merge, rec

    def ss(self, k):
        o = int(self.rec_ss_off[k])
        return self.ss_pool[o:o + int(self.rec_nss[k])].tolist()


def search_pipeline(seqs, headers=None, params=None, device=0, do_pass2=True, engine=None, pad_uniform=0):
    """pass 1 -> merge -> pass 2 on one GPU for host reads (list[bytes]); returns PipelineResult.
    pad_uniform: crass_pack_reads' layout switch (0 tight, 1 one stride, 2 automatic)."""
    packed = PackedReads(seqs, pad_uniform)
    header_id = None
    if headers is not None:
        first = {}
        header_id = np.empty(len(headers), np.uint64)
        for i, h in enumerate(headers):
            header_id[i] = first.setdefault(h, i)
        if np.all(header_id == np.arange(len(headers), dtype=np.uint64)):
            header_id = None
    own = engine is None
    eng = engine or SearchEngine(params, device)
    if not own:
        eng.reload_env()                # a reused context re-reads the A/B switches (they are read once per context)
    try:
        eng.load_reads(packed, header_id)
        cand = eng.seed_scan()
        merge = eng.merge()
        if do_pass2:
            rec = eng.recruit()
            merge = eng.merge_view()        # pass 2 may add tokens (addReadHolder)
        else:
            rec = RecruitSet(_abi.Recruits())
        res = PipelineResult(cand, merge, rec, cand.max_read_len)
        res.counters = eng.counters()
        return res
    finally:
        if own:
            eng.close()
        packed.close()


class ConsensusResult:
    """crass_cons_view as numpy / python objects (same field names as the oracle's result in tests/orc.py)"""

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
        self.counters = v.counters.asdict()

    def ss(self, k):
        o = int(self.rec_ss_off[k])
        return self.ss_pool[o:o + int(self.rec_nss[k])].tolist()

    def group_read_counts(self):
        return [sum(len(self.reads_of[t - 2] or []) for t in g) for g in self.groups]


def consensus(seqs, res, params=None, device=0):
    """crass_hip_consensus (WorkHorse::findConsensusDRs) over a search result: a PipelineResult of this module or any object
    with its fields (rec_read, rec_lowlexi, rec_token, rec_nss, rec_ss_off, ss_pool, tokens, groups, max_read_len,
    n_pass1, n_pass2).  seqs: list[bytes] or (uint8 array, uint64 offsets) — the input reads."""
    lib = _abi.load()
    p = params or default_params()
    if isinstance(seqs, (list, tuple)) and (not seqs or isinstance(seqs[0], (bytes, bytearray))):
        sbuf, soff = concat(list(seqs))
    else:
        sbuf, soff = seqs
    n = int(res.n_pass1 + res.n_pass2)
    keep = []

    def a(x, dt):
        y = np.ascontiguousarray(np.asarray(x)[:n], dtype=dt)
        keep.append(y)
        return y.ctypes.data
    tbuf, toff = concat(list(res.tokens))
    goff = np.zeros(len(res.groups) + 1, np.uint64)
    goff[1:] = np.cumsum([len(g) for g in res.groups], dtype=np.uint64)
    gt = np.array([t for g in res.groups for t in g], np.uint32)
    ssp = np.ascontiguousarray(res.ss_pool, np.uint32)
    i = _abi.ConsInput(sbuf.ctypes.data, soff.ctypes.data, len(soff) - 1, n, a(res.rec_read, np.uint64), a(res.rec_lowlexi, np.uint8),
                       a(res.rec_token, np.uint32), a(res.rec_nss, np.uint32), a(res.rec_ss_off, np.uint64), ssp.ctypes.data,
                       len(res.tokens), tbuf.ctypes.data, toff.ctypes.data, len(res.groups), gt.ctypes.data, goff.ctypes.data,
                       int(res.max_read_len))
    h = C.c_void_p()
    import time as _time
    _t0 = _time.perf_counter()
    _chk(lib.crass_hip_consensus(C.byref(p), int(device), C.byref(i), C.byref(h)), "crass_hip_consensus")
    consensus.last_call_s = _time.perf_counter() - _t0       # the C call alone (what an adapter pays), without this wrapper's conversions
    try:
        v = _abi.ConsView()
        _chk(lib.crass_hip_consensus_view(h, C.byref(v)), "crass_hip_consensus_view")
        return ConsensusResult(v)
    finally:
        lib.crass_hip_consensus_free(h)


def build_outputs(groups, out_dir="./", timestamp="", command_line="", cwd="", log_to_screen=True, cov_cutoff=0, write_to=None):
    """crass_build_outputs (WorkHorse::buildGraph ... outputResults): groups = [(gid, true_dr bytes, [(header, comment or None, seq,
    start_stops), ...])] in ascending GID, the reads in buildGraph's order.  Returns (files {name: bytes}, kept gids, stdout text);
    write_to: also write the files into that directory (crass_outputs_write)."""
    lib = _abi.load()
    gid = np.array([g for g, _, _ in groups], np.int32)
    dr_buf, dr_off = concat([d for _, d, _ in groups])
    recs = [r for _, _, rs in groups for r in rs]
    goff = np.zeros(len(groups) + 1, np.uint64)
    if groups:
        goff[1:] = np.cumsum([len(rs) for _, _, rs in groups], dtype=np.uint64)
    hb, ho = concat([r[0] for r in recs])
    cb, co = concat([(r[1] or b"") for r in recs])
    sb, so = concat([r[2] for r in recs])
    nss = np.array([len(r[3]) for r in recs], np.uint32)
    ssoff = np.zeros(len(recs), np.uint64)
    if len(recs) > 1:
        ssoff[1:] = np.cumsum(nss[:-1], dtype=np.uint64)
    pool = np.array([v for r in recs for v in r[3]] or [0], np.uint32)

    def ptr(a):
        return a.ctypes.data if a.size else None
    gi = _abi.GraphInput(len(groups), ptr(gid), ptr(dr_buf), dr_off.ctypes.data, goff.ctypes.data, len(recs), ptr(hb), ho.ctypes.data,
                         ptr(cb) if cb.size else None, co.ctypes.data, ptr(sb), so.ctypes.data, ptr(nss), ptr(ssoff), pool.ctypes.data)
    if cb.size == 0:
        cz = np.zeros(1, np.uint8)                  # (no comment bytes at all: an empty but valid buffer)
        gi.com_chars = cz.ctypes.data
    oo = _abi.OutputOpts(out_dir.encode() if isinstance(out_dir, str) else out_dir, timestamp.encode(), command_line.encode(), cwd.encode(),
                         1 if log_to_screen else 0, int(cov_cutoff), 0, 0, 0)
    h = C.c_void_p()
    _chk(lib.crass_build_outputs(C.byref(gi), C.byref(oo), C.byref(h)), "crass_build_outputs")
    try:
        v = _abi.OutputsView()
        _chk(lib.crass_outputs_get(h, C.byref(v)), "crass_outputs_get")
        files = {v.name[i].decode(): C.string_at(v.data[i], int(v.size[i])) for i in range(v.n_files)}
        kept = [int(v.kept_gid[i]) for i in range(v.n_groups_kept)]
        text = v.stdout_text.decode()
        if write_to is not None:
            _chk(lib.crass_outputs_write(h, str(write_to).encode()), "crass_outputs_write")
        return files, kept, text
    finally:
        lib.crass_outputs_free(h)
