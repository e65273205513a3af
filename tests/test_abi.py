"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/crass_hip.h declares; host-only utilities (packer, FASTX reader, generator, merge
behaviour through the oracle-independent properties).  No compute entry point is called."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import fastx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    from crass_amd import build
    build.build()
    crass_amd.load()
    return crass_amd


def test_header_symbols_all_exported(ca):
    hdr = open(os.path.join(ROOT, "include", "crass_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(crass_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = C.CDLL(ca.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(ca.SYMBOLS), declared ^ set(ca.SYMBOLS)
    assert ca.load().crass_hip_abi_version() == 3


def test_no_kernel_shifts_by_the_last_vgpr_of_its_allocation(ca):
    """engine_internal.h / crass_amd/vgpr_guard.py: on the MI355X pool a wave that shares its SIMD mis-executes a 64-bit shift whose
    AMOUNT sits in the last register of its VGPR allocation (profiles/ubench/vgpr_edge3.hip).  The guard disassembles the shipped
    library: no kernel may hold that pattern; the kernel that exposed it (24 registers, amount in v23) carries the one floor left"""
    from crass_amd import vgpr_guard
    counts, off, multiples, strays = vgpr_guard.analyse(ca.LIB_PATH)
    assert len(counts) > 100 and off == [] and strays == []
    fin = [v for k, v in counts.items() if "k_recruit_finishILb0E" in k]
    assert fin and all(v % 8 != 0 for v in fin)
    assert vgpr_guard.check(ca.LIB_PATH, verbose=False) == len(counts)


def test_the_guard_finds_the_pattern_where_it_exists(tmp_path):
    """the guard against a binary that HOLDS the pattern: the round-3 micro-benchmark's kernels with the amount in v23 of 24
    registers are reported, its controls (amount in v21; data pair ending in v23; 32-bit ops on v23) are not"""
    import shutil
    import subprocess
    from crass_amd import vgpr_guard
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "edge3")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "profiles", "ubench", "vgpr_edge3.hip")])
    counts, off, multiples, strays = vgpr_guard.analyse(exe)
    bad = {o[0] for o in off}
    assert any("k_lshr64_last" in k for k in bad) and any("lshl" in k for k in bad) and any("ashr" in k for k in bad)
    assert not any("ctrl" in k or "data_hi" in k for k in bad)
    assert all(o[3] == "v23" for o in off)
    with pytest.raises(RuntimeError, match="amount from the last VGPR"):
        vgpr_guard.check(exe)


def test_the_guard_fails_closed_on_a_failed_or_partial_disassembly(ca, monkeypatch):
    """a disassembly that fails (tool exit code) or that lacks a kernel's label must refuse the build — it used to
    file every kernel under 'ends on a granule, no edge shift', which is only a warning"""
    from crass_amd import vgpr_guard
    monkeypatch.setattr(vgpr_guard, "_shift_amounts", lambda co: {})          # the object "has no disassembly"
    with pytest.raises(RuntimeError, match="no disassembly label"):
        vgpr_guard.analyse(ca.LIB_PATH)
    monkeypatch.undo()
    import subprocess
    real = subprocess.run

    def failing(cmd, *a, **k):
        r = real(cmd, *a, **k)
        if os.path.basename(cmd[0]) == "llvm-objdump":
            r.returncode = 1
        return r
    monkeypatch.setattr(vgpr_guard.subprocess, "run", failing)
    with pytest.raises(RuntimeError, match="llvm-objdump -d failed"):
        vgpr_guard.analyse(ca.LIB_PATH)


def test_the_guard_reads_the_amount_of_lshl_add_u64():
    """v_lshl_add_u64 dst, src0, AMOUNT, src2: the fourth 64-bit shift that can shift by a VGPR on gfx950"""
    from crass_amd import vgpr_guard
    m = vgpr_guard.LSHL_ADD64.match("\tv_lshl_add_u64 v[2:3], v[4:5], v23, s[0:1]   // 0000: D2080002")
    assert m and m.group(2) == "v23"
    m = vgpr_guard.LSHL_ADD64.match("\tv_lshl_add_u64 v[2:3], v[4:5], 3, v[6:7]")
    assert m and m.group(2) == "3"
    m = vgpr_guard.LSHL_ADD64.match("\tv_lshl_add_u64 v[2:3], s[4:5], 0, v[6:7]")
    assert m and m.group(2) == "0"


def test_no_gpu_means_loud_failure_not_fallback(ca):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ca.CrassError) as e:
        ca.SearchEngine()
    assert e.value.status == 3


def test_param_validation(ca):
    import torch
    if torch.cuda.is_available():
        pytest.skip("validated on the CPU box only")
    lib = ca.load()
    from crass_amd import _abi
    h = C.c_void_p()
    for kw in (dict(searchWindowLength=5), dict(searchWindowLength=10), dict(minNumRepeats=1),
               dict(lowDRsize=47), dict(lowSpacerSize=50)):
        p = ca.default_params(**kw)
        assert lib.crass_hip_create(C.byref(p), 0, C.byref(h)) == 1
    assert lib.crass_hip_create(C.byref(ca.default_params(highDRsize=500)), 0, C.byref(h)) == 2
    assert lib.crass_hip_create(C.byref(ca.default_params(lowDRsize=8, searchWindowLength=9)), 0, C.byref(h)) == 2
    assert b"Fatal error in search algorithm" in lib.crass_hip_strerror(7)


def test_group_argument_validation_needs_no_gpu(ca):
    """crass_hip_group_create refuses bad argument lists before it touches a device; a device listed twice needs the
    local-copy stand-in (RCCL wants one rank per device) and says so in crass_hip_group_last_error()"""
    lib = ca.load()
    p = ca.default_params()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.crass_hip_group_create(C.byref(p), devs, 0, 0, C.byref(h)) == 1
    assert lib.crass_hip_group_create(C.byref(p), None, 2, 0, C.byref(h)) == 1
    assert lib.crass_hip_group_create(C.byref(p), devs, 2, 8, C.byref(h)) == 1          # unknown flag
    assert lib.crass_hip_group_create(C.byref(p), devs, 2, 0, C.byref(h)) == 1
    assert b"listed twice" in lib.crass_hip_group_last_error()
    assert lib.crass_hip_group_size(None) == 0 and lib.crass_hip_group_rccl_ranks(None) == 0
    assert lib.crass_hip_group_step(None) == 1 and lib.crass_hip_group_recruit(None, None, 0) == 1
    assert b"RCCL" in lib.crass_hip_strerror(10)
    import torch
    if not torch.cuda.is_available():                   # valid arguments, no device: the contexts cannot be created
        assert lib.crass_hip_group_create(C.byref(p), devs, 2, 1, C.byref(h)) == 3


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """a group whose RCCL library cannot be loaded must come back with CRASS_ERR_RCCL and a text, before any device is
    touched (the loader used to read dlerror() twice: the second read is NULL).  In a child process: the binding is per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import crass_amd as ca\n"
        "lib = ca.load(); p = ca.default_params(); h = C.c_void_p(); devs = (C.c_int * 2)(0, 1)\n"
        "st = lib.crass_hip_group_create(C.byref(p), devs, 2, 0, C.byref(h))\n"
        "print(st, lib.crass_hip_group_last_error().decode())\n"
        "st2 = lib.crass_hip_group_create(C.byref(p), devs, 2, 0, C.byref(h))\n"
        "print(st2)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, CRASS_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("10 RCCL not found:") and "no_such_librccl" in lines[0]
    assert lines[1] == "10"


def test_outputs_stage_argument_validation(ca):
    from crass_amd import _abi
    lib = ca.load()
    h = C.c_void_p()
    assert lib.crass_build_outputs(None, None, C.byref(h)) == 1
    gi, oo = _abi.GraphInput(), _abi.OutputOpts()
    gi.n_groups = 1                                      # a group without its arrays
    assert lib.crass_build_outputs(C.byref(gi), C.byref(oo), C.byref(h)) == 1
    files, kept, text = ca.build_outputs([])            # no groups: an empty but well-formed result
    assert kept == [] and files["crass.crispr"].endswith(b'<crispr version="1.1"/>\n') and "0 CRISPRs found!" in text


@pytest.mark.parametrize("fname", sorted(os.listdir(DATA)))
def test_fastx_reader_matches_kseq_semantics(ca, fname):
    path = os.path.join(DATA, fname)
    assert ca.FastxFile(path).records() == fastx.read_fastx(path)


def test_packer_roundtrip_and_exceptions(ca):
    recs = fastx.read_fastx(os.path.join(DATA, "CN_gDC.fa.gz"))
    seqs = [r[2] for r in recs]
    p = ca.PackedReads(seqs)
    r = p.reads
    assert r.n_reads == len(seqs) and r.uniform_len == 150 and r.stride_words == 10
    exc = set(np.ctypeslib.as_array(C.cast(r.exc_read, C.POINTER(C.c_uint64)), shape=(int(r.n_exceptions),)).tolist())
    assert exc == {i for i, s in enumerate(seqs) if set(s) - set(b"ACGT")}
    words = p.packed_array()
    asc = ca.unpack_ascii(words, 10, 150, len(seqs))
    for i, s in enumerate(seqs):
        if i not in exc:
            assert asc[i * 150:(i + 1) * 150].tobytes() == s
    # ragged input -> per-read offsets
    rag = ca.PackedReads([b"ACGT", b"ACGTACGTACGTACGTACG", b"", b"TTTT"])
    assert rag.reads.stride_words == 0 and rag.reads.uniform_len == 0


def test_synthetic_generator_is_deterministic_and_shardable(ca):
    spec = ca.synth_spec(read_len=150)
    a = ca.synth_packed(spec, 0, 20000)
    b = np.concatenate([ca.synth_packed(spec, 0, 7000, n_threads=1), ca.synth_packed(spec, 7000, 13000, n_threads=3)])
    assert np.array_equal(a, b)
    asc = ca.unpack_ascii(a, 10, 150, 20000).reshape(20000, 150)
    assert set(np.unique(asc).tolist()) == set(b"ACGT")
    # padding bases (150..159) are zero
    assert np.all((a.reshape(20000, 10)[:, 9] >> 12) == 0)
    gc = ca.synth_packed(ca.synth_spec(read_len=150, gc_classes=4), 0, 4000)
    gasc = ca.unpack_ascii(gc, 10, 150, 4000).reshape(4000, 150)
    frac = ((gasc == ord("G")) | (gasc == ord("C"))).mean(axis=1)
    assert frac.min() < 0.38 and frac.max() > 0.62


def _write_both(tmp_path, name, text):
    import gzip
    plain = tmp_path / name
    plain.write_bytes(text)
    gz = tmp_path / (name + ".gz")
    with gzip.open(gz, "wb") as f:
        f.write(text)
    return str(plain), str(gz)


def _fastx_cases():
    import random
    rng = random.Random(11)

    def seq(n):
        return "".join(rng.choice("ACGTN" if rng.random() < 0.05 else "ACGT") for _ in range(n))

    cases = {}
    # a. single-line FASTA, comments on some headers only (stale-buffer semantics), duplicate names
    recs = []
    for i in range(3000):
        nm = "r%d" % (i if i % 97 else i // 2)
        recs.append(">%s%s\n%s\n" % (nm, (" c%d x" % i) if (i > 40 and i % 3 == 0) else "", seq(rng.randint(20, 180))))
    cases["fa_single"] = "".join(recs)
    # b. multi-line FASTA
    recs = []
    for i in range(1500):
        s = seq(rng.randint(50, 400))
        recs.append(">m%d desc\n%s\n" % (i, "\n".join(s[k:k + 60] for k in range(0, len(s), 60))))
    cases["fa_multi"] = "".join(recs)
    # c. FASTQ, quality lines that start with '@' or '+'
    recs = []
    for i in range(2500):
        s = seq(rng.randint(30, 150))
        q = "".join(rng.choice("@+IIIHG5#!~") for _ in s)
        if i % 5 == 0:
            q = "@" + q[1:]
        if i % 7 == 0:
            q = "+" + q[1:]
        recs.append("@q%d%s\n%s\n+\n%s\n" % (i, " lane=%d" % (i % 4) if i % 2 else "", s, q))
    cases["fq"] = "".join(recs)
    # c2. FASTQ odds and ends: an empty sequence, a quality string wrapped over lines, blanks inside one (skipped by kseq), bytes
    # beyond 127 in one, a quality line longer than the sequence (the rest is skipped as the next record is looked for)
    recs = []
    for i in range(1200):
        s = seq(rng.randint(10, 120))
        q = "".join(rng.choice("IIIHG5#!~") for _ in s)
        k = i % 12
        if k == 3:
            s, q = "", ""
        elif k == 5 and len(q) > 20:
            q = q[:7] + "\n" + q[7:15] + "\n" + q[15:]
        elif k == 7 and len(q) > 4:
            q = q[:3] + " " + q[3:]
        elif k == 9:
            q = q + "IIII"
        elif k == 11 and len(q) > 4:
            q = q[:2] + "\t" + q[2:]
        recs.append("@e%d\n%s\n+\n%s\n" % (i, s, q))
    cases["fq_edge"] = "".join(recs)
    # d. the same, truncated inside the last quality line
    cases["fq_trunc"] = cases["fq"][:-40]
    # e. '>' inside a sequence line (ends the record there in kseq): the pieces cannot line up, one-piece parse
    cases["fa_odd"] = cases["fa_single"][:20000] + ">odd\nACGT>ACGTAC\nGGGG\n" + cases["fa_single"][20000:]
    # f. CRLF, g. no trailing newline
    cases["fa_crlf"] = cases["fa_single"].replace("\n", "\r\n")
    cases["fa_noeol"] = cases["fa_multi"].rstrip("\n")
    return {k: v.encode() for k, v in cases.items()}


@pytest.mark.parametrize("case", ["fa_single", "fa_multi", "fq", "fq_edge", "fq_trunc", "fa_odd", "fa_crlf", "fa_noeol"])
@pytest.mark.parametrize("chunk", ["256", "5000", "serial"])
def test_parallel_fastx_reader_is_kseq_exact(ca, tmp_path, case, chunk):
    """the block-parallel reader against the byte-at-a-time kseq reference, with pieces far smaller than in
    production so that every boundary condition occurs, on plain and gzipped copies of the same text"""
    text = _fastx_cases()[case]
    plain, gz = _write_both(tmp_path, case + ".txt", text)
    ref = fastx.read_fastx(gz)
    first = {}
    ref_ids = [first.setdefault(r[0], i) for i, r in enumerate(ref)]
    os.environ.pop("CRASS_FASTX_CHUNK", None)
    os.environ.pop("CRASS_FASTX_SERIAL", None)
    if chunk == "serial":
        os.environ["CRASS_FASTX_SERIAL"] = "1"
    else:
        os.environ["CRASS_FASTX_CHUNK"] = chunk
    try:
        for path in (plain, gz):
            f = ca.FastxFile(path)
            assert f.records() == ref
            assert f.header_id.tolist() == ref_ids
            assert f.max_len == max((len(r[2]) for r in ref), default=0)
    finally:
        os.environ.pop("CRASS_FASTX_CHUNK", None)
        os.environ.pop("CRASS_FASTX_SERIAL", None)
    assert len(ref) > 1000


@pytest.mark.parametrize("case", ["fa_single", "fa_multi", "fq", "fq_edge", "fq_trunc", "fa_odd", "fa_crlf", "fa_noeol"])
@pytest.mark.parametrize("chunk", [300, 4096, 65536, 0])
def test_streaming_fastx_reader_is_kseq_exact(ca, tmp_path, case, chunk):
    """the chunked reader (crass_fastx_stream_*, bounded host memory) against the byte-at-a-time kseq reference: chunks from far
    smaller than a record (the stream grows them) to larger than the file, plain and gzipped; records, stale comment / quality
    carry-over across chunk ends, job-level header ids through the 128-bit name table, kseq_read's final return value"""
    text = _fastx_cases()[case]
    plain, gz = _write_both(tmp_path, case + ".txt", text)
    ref = fastx.read_fastx(gz)
    first = {}
    ref_ids = [first.setdefault(r[0], i) for i, r in enumerate(ref)]
    whole = ca.FastxFile(plain)
    for path in (plain, gz):
        recs, hid, last, chunks = ca.stream_fastx(path, chunk_bytes=chunk)
        assert recs == ref
        assert hid == ref_ids
        assert last == whole.last_ret
        if chunk and chunk < len(text) // 4:
            assert chunks > 3
    assert len(ref) > 1000


def test_streaming_reader_grows_geometrically_for_a_record_far_longer_than_the_chunk(ca, tmp_path):
    """a record k blocks long was copied and parsed k times (O(k^2)); the stream now asks for twice the bytes before
    it parses again.  3 MB records through 300-byte chunks (10^4 blocks each) finish in seconds and come back whole"""
    import random, time
    rng = random.Random(5)
    recs = [(">r%d c%d" % (i, i), "".join(rng.choice("ACGT") for _ in range(n))) for i, n in enumerate([3_000_000, 70, 1_500_000, 3, 90])]
    text = "".join("%s\n%s\n" % (h, "\n".join(sq[k:k + 70] for k in range(0, len(sq), 70))) for h, sq in recs)
    plain, gz = _write_both(tmp_path, "long.fa", text.encode())
    ref = fastx.read_fastx(gz)
    t0 = time.perf_counter()
    for path in (plain, gz):
        got, hid, last, chunks = ca.stream_fastx(path, chunk_bytes=300)
        assert got == ref
        assert hid == list(range(len(recs)))
    assert time.perf_counter() - t0 < 30


@pytest.mark.parametrize("bits", ["1", "8", "12", "14"])
def test_indexed_reader_header_ids_with_any_shard_count(ca, tmp_path, bits):
    """the header table of the indexed reader is sharded by the name hash's top bits (12 for jobs of millions of reads, 8 below);
    CRASS_HDR_SHARD_BITS forces a count: names repeated within and across pieces get the first record of their name either way"""
    import random
    rng = random.Random(int(bits))
    names = ["n%d" % rng.randrange(3000) for _ in range(9000)]
    text = "".join(">%s\n%s\n" % (nm, "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 90)))) for nm in names).encode()
    plain, _ = _write_both(tmp_path, "dups.fa", text)
    first = {}
    ref_ids = [first.setdefault(nm, i) for i, nm in enumerate(names)]
    os.environ["CRASS_HDR_SHARD_BITS"] = bits
    os.environ["CRASS_FASTX_CHUNK"] = "20000"
    try:
        ix = ca.FastxIndex(plain)
    finally:
        os.environ.pop("CRASS_HDR_SHARD_BITS", None)
        os.environ.pop("CRASS_FASTX_CHUNK", None)
    assert ix.n_reads == len(names)
    assert ix.layout()["header_id"] == ref_ids


@pytest.mark.parametrize("avx2", [True, False])
def test_indexed_reader_packs_every_line_length(ca, tmp_path, avx2):
    """the index packs the common record's sequence line in one pass of 32-byte steps that also finds the line's end (AVX2; the
    scalar path with CRASS_NO_AVX2): every length 1 .. 200 — block boundaries, the last block's masked lanes, the record at the very
    end of the file — and lines with one byte outside ACGT at any place (the general path takes them: exception reads)"""
    import random
    rng = random.Random(77)
    seqs = []
    for L in range(1, 201):
        seqs.append("".join(rng.choice("ACGT") for _ in range(L)))
        bad = list("".join(rng.choice("ACGT") for _ in range(L)))
        bad[rng.randrange(L)] = rng.choice("NacgtRY-")
        seqs.append("".join(bad))
    rng.shuffle(seqs)
    text = "".join(">s%d\n%s\n" % (i, sq) for i, sq in enumerate(seqs)).encode()
    plain, _ = _write_both(tmp_path, "lens.fa", text)
    if not avx2:
        os.environ["CRASS_NO_AVX2"] = "1"
    try:
        import subprocess, sys, json
        # (the switch is read once per process: a process of its own)
        code = ("import sys, json; sys.path.insert(0, %r); import crass_amd as ca; ca.load(); ix = ca.FastxIndex(%r); lay = ix.layout(); "
                "print(json.dumps({'n': int(ix.n_reads), 'lengths': [int(x) for x in lay['lengths']], 'exc': sorted(int(x) for x in lay['exceptions']), "
                "'words': [[int(v) for v in w] for w in lay['words']]}))") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), plain)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
        got = json.loads(out.strip().splitlines()[-1])
    finally:
        os.environ.pop("CRASS_NO_AVX2", None)
    want = _packed_layout(ca, [sq.encode() for sq in seqs])
    assert got["n"] == len(seqs) and got["lengths"] == want["lengths"]
    assert got["exc"] == sorted(want["exceptions"]) and len(got["exc"]) >= 150
    for i in range(len(seqs)):
        if i not in want["exceptions"]:
            assert got["words"][i] == [int(v) for v in want["words"][i]], (i, len(seqs[i]))


def _packed_layout(ca, seqs):
    """what crass_pack_reads (mode 2) makes of these sequences, in FastxIndex.layout()'s form"""
    pk = ca.PackedReads(list(seqs), pad_uniform=2)
    r = pk.reads
    n = int(r.n_reads)
    from crass_amd.engine import _npv as _np
    stride, uni = int(r.stride_words), int(r.uniform_len)
    lengths = [uni] * n if uni else _np(r.lengths, n, np.uint32).tolist()
    if stride:
        words = _np(r.packed, n * stride, np.uint32).reshape(n, stride) if n else np.zeros((0, stride), np.uint32)
        per = [words[i, :(lengths[i] + 15) // 16] for i in range(n)]
    else:
        off = _np(r.word_off, n + 1, np.uint64)
        allw = _np(r.packed, int(off[n]), np.uint32) if n else np.zeros(0, np.uint32)
        per = [allw[int(off[i]):int(off[i]) + (lengths[i] + 15) // 16] for i in range(n)]
    ne = int(r.n_exceptions)
    exc = {}
    if ne:
        er, eo = _np(r.exc_read, ne, np.uint64), _np(r.exc_off, ne + 1, np.uint64)
        eb = _np(r.exc_bytes, int(eo[ne]), np.uint8)
        exc = {int(er[k]): eb[int(eo[k]):int(eo[k + 1])].tobytes() for k in range(ne)}
    return dict(stride=stride, uniform_len=uni, words=per, lengths=lengths, exceptions=exc)


@pytest.mark.parametrize("case", ["fa_multi", "fq", "fq_edge", "fq_trunc", "fa_odd_nocomment", "fa_crlf_nocomment", "fa_noeol", "fa_uniform", "fa_trimmed"])
@pytest.mark.parametrize("chunk", ["256", "5000", "serial"])
def test_indexed_fastx_reader_is_the_whole_file_reader(ca, tmp_path, case, chunk):
    """crass_index_fastx (the input kept mapped, reads packed at once, text on request) against the whole-file reader + crass_pack_reads
    and the byte-at-a-time kseq reference: layout (stride / ragged / padded), every read's words, lengths, exception reads with their
    bytes, header ids, and the text of fetched records — pieces far smaller than in production; uniform-length, trimmed (padded to
    one stride) and ragged sets, multi-line FASTA, FASTQ incl. a truncated last record, '>' inside a sequence line, CRLF"""
    import random
    cases = _fastx_cases()
    rng = random.Random(5)
    strip = lambda t: b"".join(ln.split(b" ")[0] + (b"\r\n" if ln.endswith(b"\r\n") else b"\n") if ln[:1] == b">" else ln for ln in t.splitlines(keepends=True))
    cases["fa_odd_nocomment"] = strip(cases["fa_odd"])
    cases["fa_crlf_nocomment"] = strip(cases["fa_crlf"])
    # FASTQ with a comment on EVERY header (the shared cases put one on every other record: refused, see the next test), quality
    # lines that start with '@' or '+'
    recs = []
    for i in range(2500):
        sq = "".join(rng.choice("ACGTN" if rng.random() < 0.05 else "ACGT") for _ in range(rng.randint(30, 150)))
        q = "".join(rng.choice("@+IIIHG5#!~") for _ in sq)
        q = ("@" + q[1:]) if i % 5 == 0 else (("+" + q[1:]) if i % 7 == 0 else q)
        recs.append("@q%d lane=%d\n%s\n+\n%s\n" % (i, i % 4, sq, q))
    cases["fq"] = "".join(recs).encode()
    cases["fq_trunc"] = cases["fq"][:-40]
    cases["fa_uniform"] = "".join(">u%d\n%s\n" % (i % 1900, "".join(rng.choice("ACGT") for _ in range(150))) for i in range(2000)).encode()
    cases["fa_trimmed"] = "".join(">t%d\n%s\n" % (i, "".join(rng.choice("ACGTN" if i % 50 == 0 else "ACGT") for _ in range(rng.randint(90, 150)))) for i in range(2000)).encode()
    text = cases[case]
    plain, gz = _write_both(tmp_path, case + ".txt", text)
    ref = fastx.read_fastx(gz)
    first = {}
    ref_ids = [first.setdefault(r[0], i) for i, r in enumerate(ref)]
    os.environ.pop("CRASS_FASTX_CHUNK", None)
    os.environ.pop("CRASS_FASTX_SERIAL", None)
    if chunk == "serial":
        os.environ["CRASS_FASTX_SERIAL"] = "1"
    else:
        os.environ["CRASS_FASTX_CHUNK"] = chunk
    try:
        ix = ca.FastxIndex(plain)
        try:
            C.CDLL("libdeflate.so.0")
            ixz = ca.FastxIndex(gz)                                 # gzip'd: the same index over the inflated image (libdeflate)
        except OSError:
            ixz = None                                              # (no libdeflate on this host: refused, as below)
        whole = ca.FastxFile(plain)
        os.environ["CRASS_NO_LIBDEFLATE"] = "1"
        try:
            with pytest.raises(ca.CrassError):
                ca.FastxIndex(gz)                                   # ... without the library: the other readers' job
        finally:
            os.environ.pop("CRASS_NO_LIBDEFLATE", None)
    finally:
        os.environ.pop("CRASS_FASTX_CHUNK", None)
        os.environ.pop("CRASS_FASTX_SERIAL", None)
    assert ix.n_reads == len(ref) and ix.max_len == whole.max_len and ix.last_ret == whole.last_ret
    lay = ix.layout()
    want = _packed_layout(ca, [r[2] for r in ref])
    assert lay["stride"] == want["stride"] and lay["uniform_len"] == want["uniform_len"] and lay["lengths"] == want["lengths"]
    assert lay["exceptions"] == want["exceptions"]
    for i in range(len(ref)):
        if i not in want["exceptions"]:                             # (an exception read's slot content is ignored)
            assert np.array_equal(lay["words"][i], want["words"][i]), i
    assert lay["header_id"] == ref_ids
    pick = list(range(0, len(ref), 7)) + [len(ref) - 1, 0, 3, 3]
    assert ix.fetch(pick) == [ref[i] for i in pick]
    assert ix.fetch([]) == []
    ix.drop_text()                                                  # (the mapping given back: the same records, read from the file)
    assert ix.fetch(pick) == [ref[i] for i in pick]
    assert ix.fetch(list(range(len(ref)))) == ref
    ix.close()
    if ixz is not None:
        layz = ixz.layout()
        assert ixz.n_reads == len(ref) and ixz.max_len == whole.max_len and ixz.last_ret == whole.last_ret
        assert all(layz[k] == lay[k] for k in ("stride", "uniform_len", "lengths", "exceptions", "header_id"))
        assert all(np.array_equal(a, b) for a, b in zip(layz["words"], lay["words"]))
        assert ixz.fetch(pick) == [ref[i] for i in pick]
        ixz.close()
    assert len(ref) > 1000
    if case == "fa_trimmed":
        assert lay["stride"] == 10 and lay["uniform_len"] == 0 and len(lay["exceptions"]) >= 30
    if case == "fa_uniform":
        assert lay["stride"] == 10 and lay["uniform_len"] == 150 and lay["header_id"][1900] == 0


def test_indexed_reader_over_several_inputs(ca, tmp_path):
    """crass_index_fastx_files: paired-end style inputs — one read set in (file, read) order, header ids across the files (mates
    that share a name; readsFound is keyed by the header string whatever file it came from, libcrispr.cpp:138,411), a FASTQ next to
    a gzip'd FASTA next to a multi-line FASTA, fetch across the files; against the whole-file reader file by file"""
    import random
    rng = random.Random(31)
    def sq(n):
        return "".join(rng.choice("ACGTN" if rng.random() < 0.03 else "ACGT") for _ in range(n))
    t1 = "".join("@p%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in ((i, sq(rng.randint(40, 150))) for i in range(1500)))
    t2 = "".join(">p%d\n%s\n" % (i if i % 3 else 5000 + i, sq(rng.randint(40, 150))) for i in range(1500))        # two thirds share file 1's names
    s3 = [sq(rng.randint(100, 400)) for _ in range(700)]
    t3 = "".join(">m%d desc\n%s\n" % (i % 650, "\n".join(s[k:k + 60] for k in range(0, len(s), 60))) for i, s in enumerate(s3))
    p1, g1 = _write_both(tmp_path, "a.fq", t1.encode())
    _, p2 = _write_both(tmp_path, "b.fa", t2.encode())
    p3, g3 = _write_both(tmp_path, "c.fa", t3.encode())
    os.environ["CRASS_FASTX_CHUNK"] = "3000"
    try:
        ix = ca.FastxIndex([p1, p2, p3])
    finally:
        os.environ.pop("CRASS_FASTX_CHUNK", None)
    ref = fastx.read_fastx(g1) + fastx.read_fastx(p2) + fastx.read_fastx(g3)
    assert ix.n_reads == len(ref) == 3700
    first = {}
    ref_ids = [first.setdefault(r[0], i) for i, r in enumerate(ref)]
    lay = ix.layout()
    want = _packed_layout(ca, [r[2] for r in ref])
    assert lay["stride"] == want["stride"] and lay["uniform_len"] == want["uniform_len"] and lay["lengths"] == want["lengths"]
    assert lay["exceptions"] == want["exceptions"] and lay["header_id"] == ref_ids and sum(1 for i, h in enumerate(ref_ids) if h != i) > 900
    for i in range(len(ref)):
        if i not in want["exceptions"]:
            assert np.array_equal(lay["words"][i], want["words"][i]), i
    pick = list(range(0, len(ref), 11)) + [len(ref) - 1, 0, 1499, 1500, 2999, 3000]
    assert ix.fetch(pick) == [ref[i] for i in pick]
    ix.close()
    # one input among them that the index does not take (comments on some records only): the whole job is refused
    mixed, _ = _write_both(tmp_path, "mixed.fa", _fastx_cases()["fa_single"])
    with pytest.raises(ca.CrassError) as e:
        ca.FastxIndex([p1, mixed])
    assert e.value.status == 2


def test_readers_on_random_record_soup(ca, tmp_path):
    """300 small random FASTA / FASTQ-like texts — wrapped and unwrapped lines, blank lines, CRLF, lower case and N, '>' '+' '@' in odd
    places, comments on some / all / no headers, quality strings wrapped, too short, too long, a truncated tail — through the
    whole-file reader, the streamed reader (tiny chunks) and, where it takes the file, the index (tiny pieces; its records packed in
    place or through the general path): all == the byte-at-a-time kseq reference"""
    import random
    rng = random.Random(20251003)
    n_index = n_refused = 0
    for case in range(300):
        fq = rng.random() < 0.5
        com_mode = rng.choice(["none", "all", "some"])
        eol = "\r\n" if rng.random() < 0.15 else "\n"
        wrap = rng.choice([0, 0, 7, 60])
        nasty = rng.random() < 0.25                              # ('>' '+' '@' inside sequence lines end a record there: mostly files the index refuses)
        parts = []
        for i in range(rng.randint(1, 40)):
            L = rng.choice([0, 1, 5, 16, 17, 31, 64, rng.randint(1, 200)])
            alpha = rng.choice(["ACGT", "ACGT", "ACGT", "ACGTN", "acgtACGT", "ACGT>", "ACGT+", "ACGT@", "ACG T"] if nasty else ["ACGT", "ACGT", "ACGT", "ACGTN", "acgtACGT", "ACG T"])
            s = "".join(rng.choice(alpha) for _ in range(L))
            body = eol.join(s[k:k + wrap] for k in range(0, len(s), wrap)) if wrap and s else s
            com = " c%d x" % i if com_mode == "all" or (com_mode == "some" and rng.random() < 0.5) else ""
            if rng.random() < 0.1:
                body += eol                                         # a blank line behind the sequence
            if fq:
                ql = len(s.replace(" ", "")) + (rng.choice([0, 0, 0, 0, -1, 2]) if nasty else 0)
                q = "".join(rng.choice("IIIH5#!~@+>") for _ in range(max(0, ql)))
                if rng.random() < 0.2 and len(q) > 6:
                    q = q[:5] + eol + q[5:]
                parts.append("@r%d%s%s%s%s+%s%s%s" % (i, com, eol, body, eol, eol, q, eol))
            else:
                parts.append(">r%d%s%s%s%s" % (i % 13 if rng.random() < 0.3 else i, com, eol, body, eol))
        text = "".join(parts)
        if rng.random() < 0.2:
            text = text[:max(1, len(text) - rng.randint(1, 30))]
        if rng.random() < 0.1:
            text = "junk before the first record" + eol + text
        plain, gz = _write_both(tmp_path, "soup%d.txt" % case, text.encode())
        ref = fastx.read_fastx(gz)
        first = {}
        ref_ids = [first.setdefault(r[0], i) for i, r in enumerate(ref)]
        os.environ["CRASS_FASTX_CHUNK"] = str(rng.choice([64, 200, 1000]))
        try:
            whole = ca.FastxFile(plain)
            assert whole.records() == ref, (case, text)
            assert whole.header_id.tolist() == ref_ids, case
            recs, hid, last, _ = ca.stream_fastx(plain, chunk_bytes=rng.choice([256, 300, 1000, 0]))
            assert recs == ref and hid == ref_ids and last == whole.last_ret, (case, text)
            try:
                ix = ca.FastxIndex(plain)
            except ca.CrassError as e:
                assert e.status == 2, case                           # mixed comments / qualities: the ordered readers' file
                n_refused += 1
                continue
        finally:
            os.environ.pop("CRASS_FASTX_CHUNK", None)
        n_index += 1
        assert ix.n_reads == len(ref) and ix.last_ret == whole.last_ret, (case, text)
        lay = ix.layout()
        want = _packed_layout(ca, [r[2] for r in ref]) if ref else None
        if ref:
            assert lay["lengths"] == want["lengths"] and lay["exceptions"] == want["exceptions"] and lay["header_id"] == ref_ids, (case, text)
            assert lay["stride"] == want["stride"] and lay["uniform_len"] == want["uniform_len"], (case, text)
            for i in range(len(ref)):
                if i not in want["exceptions"]:
                    assert np.array_equal(lay["words"][i], want["words"][i]), (case, i, text)
            assert ix.fetch(list(range(len(ref)))) == ref, (case, text)
        ix.close()
    assert n_index >= 120 and n_refused >= 20, (n_index, n_refused)


def _write_bgzf(path, data, block=60000, eof_block=True):
    """BGZF as bgzip writes it (SAM spec 4.1): gzip members of <= 64 KB with the 'BC' extra field holding the member's size - 1"""
    import struct, zlib
    with open(path, "wb") as f:
        chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_block else [])
        for c in chunks:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = co.compress(c) + co.flush()
            bsize = 12 + 6 + len(d) + 8 - 1
            assert bsize < 65536
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + d +
                    struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c) & 0xFFFFFFFF))


def test_bgzf_inputs_are_inflated_member_by_member(ca, tmp_path):
    """a BGZF input (bgzip; the .fastq.gz of Illumina's converters): its members' sizes are read from their headers and trailers and the
    members inflated side by side — the same records as through the serial path (CRASS_NO_BGZF), zlib (CRASS_NO_LIBDEFLATE) and the
    reference reader; a damaged member and a member without the size field fall back to the serial path and its verdict"""
    import random
    rng = random.Random(3)
    recs = []
    for i in range(60000):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(80, 150)))
        recs.append("@b%d\n%s\n+\n%s\n" % (i, s, "".join(rng.choice("IH5#") for _ in s)))
    text = "".join(recs).encode()
    bg = str(tmp_path / "reads.fq.gz")
    _write_bgzf(bg, text, block=rng.choice([30000, 60000, 65000]))
    assert os.path.getsize(bg) > 70 * 20000                  # (more than 64 members: the side-by-side path)
    ref = fastx.read_fastx(bg)
    assert len(ref) == 60000
    got = {}
    for env in ({}, {"CRASS_NO_BGZF": "1"}, {"CRASS_NO_LIBDEFLATE": "1"}):
        os.environ.update(env)
        try:
            f = ca.FastxFile(bg)
            got[tuple(env)] = f.records()
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert got[tuple(env)] == ref
    ix = ca.FastxIndex(bg)
    assert ix.n_reads == 60000 and ix.fetch([0, 59999, 31234]) == [ref[0], ref[59999], ref[31234]]
    ix.close()
    # a member whose deflate data is damaged: the side-by-side path gives up, the serial path reports what zlib / gzip report
    raw = bytearray(open(bg, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ca.CrassError):
        ca.FastxFile(str(bad))
    # plain gzip in front of BGZF members (cat a.gz b.bgzf): not BGZF from the start, the serial path reads both
    import gzip
    mixed = tmp_path / "mixed.fq.gz"
    head = "".join(recs[:500]).encode()
    mixed.write_bytes(gzip.compress(head) + open(bg, "rb").read())
    assert ca.FastxFile(str(mixed)).records() == fastx.read_fastx(str(mixed))


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("kind", ["random_fasta", "repeats", "fastq", "wrapped"])
def test_one_gzip_member_inflated_by_several_threads(ca, tmp_path, kind, level):
    """csrc/pgzip.cpp: a gzip member entered at block starts found in the middle of the stream, the unknown 32 KB in front of a chunk
    carried as markers and resolved afterwards, accepted only with the trailer's length and CRC-32.  Thresholds lowered so that a
    few megabytes are cut into dozens of chunks, some shorter than a window: the same records as the serial path (zlib /
    libdeflate) for literal-heavy text, long-distance repeats, quality strings, wrapped lines; several members and a damaged
    stream go back to the serial path and its verdict"""
    import gzip, random
    rng = random.Random(hash((kind, level)) & 0xFFFF)
    recs = []
    if kind == "random_fasta":
        for i in range(40000):
            recs.append(">r%d\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(150))))
    elif kind == "repeats":
        pool = ["".join(rng.choice("ACGT") for _ in range(rng.randint(100, 250))) for _ in range(60)]
        for i in range(90000):
            recs.append(">r%d\n%s\n" % (i % 977, pool[(i * 7 + i // 11) % len(pool)]))
    elif kind == "fastq":
        for i in range(30000):
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(50, 150)))
            recs.append("@q%d l=%d\n%s\n+\n%s\n" % (i, i % 8, s, "".join(rng.choice("FFFFF:,#") for _ in s)))
    else:
        for i in range(8000):
            s = "".join(rng.choice("ACGTN") for _ in range(rng.randint(300, 1200)))
            recs.append(">w%d\n%s\n" % (i, "\n".join(s[k:k + 70] for k in range(0, len(s), 70))))
    text = "".join(recs).encode()
    gz = tmp_path / "one.gz"
    gz.write_bytes(gzip.compress(text, level))
    os.environ["CRASS_NO_PGZIP"] = "1"
    try:
        want = ca.FastxFile(str(gz)).records()
    finally:
        os.environ.pop("CRASS_NO_PGZIP", None)
    os.environ["CRASS_TIMING"] = "1"
    for chunk in ("16384", "100000", "400000"):
        os.environ.update({"CRASS_PGZIP_MIN_BYTES": "1000", "CRASS_PGZIP_CHUNK_BYTES": chunk})
        try:
            got = ca.FastxFile(str(gz)).records()
        finally:
            for k in ("CRASS_PGZIP_MIN_BYTES", "CRASS_PGZIP_CHUNK_BYTES"):
                os.environ.pop(k, None)
        assert got == want, chunk
    os.environ.pop("CRASS_TIMING", None)
    if kind == "random_fasta" and level == 6:
        os.environ.update({"CRASS_PGZIP_MIN_BYTES": "1000", "CRASS_PGZIP_CHUNK_BYTES": "100000"})
        try:
            two = tmp_path / "two.gz"
            two.write_bytes(gzip.compress(text[:len(text) // 2], level) + gzip.compress(text[len(text) // 2:], level))
            assert ca.FastxFile(str(two)).records() == want       # several members: the serial path reads them all
            raw = bytearray(gz.read_bytes())
            raw[len(raw) // 3] ^= 0x10
            bad = tmp_path / "bad.gz"
            bad.write_bytes(bytes(raw))
            with pytest.raises(ca.CrassError):
                ca.FastxFile(str(bad))
        finally:
            for k in ("CRASS_PGZIP_MIN_BYTES", "CRASS_PGZIP_CHUNK_BYTES"):
                os.environ.pop(k, None)


def test_several_thread_inflate_on_arbitrary_bytes(ca):
    """csrc/pgzip.cpp called directly (its C++ symbol) on data that is not sequence text: incompressible bytes (stored blocks), tiny
    alphabets and long runs (long matches at every distance, fixed-Huffman blocks at the ends), a mix — every zlib level and strategy,
    chunks down to 4 KB: byte-identical to zlib's inflate, or not taken"""
    import random, zlib
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(ca.__file__)), "libcrass_hip.so"))
    fn = getattr(lib, "_ZN5crass15parallel_gunzipEPKhmPPhPmj")
    fn.restype = C.c_bool
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = random.Random(77)
    def gz(data, level, strategy):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
        return co.compress(data) + co.flush()
    kinds = {
        "random": lambda n: rng.randbytes(n),
        "runs": lambda n: b"".join(bytes([rng.randrange(4)]) * rng.randrange(1, 5000) for _ in range(n // 2500)),
        "text": lambda n: b"".join(rng.choice([b"ACGT", b"GGGA", b"TTTTTTTT", b"\n>r\n"]) for _ in range(n // 5)),
        "mixed": lambda n: b"".join((rng.randbytes(rng.randrange(1, 70000)) if rng.random() < 0.3 else bytes([65 + rng.randrange(3)]) * rng.randrange(1, 90000)) for _ in range(n // 40000)),
    }
    taken = 0
    os.environ["CRASS_PGZIP_MIN_BYTES"] = "100"
    try:
        for kind, make in kinds.items():
            for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (0, zlib.Z_DEFAULT_STRATEGY)):
                data = make(3_000_000)
                blob = gz(data, level, strategy)
                for chunk in ("4096", "50000"):
                    os.environ["CRASS_PGZIP_CHUNK_BYTES"] = chunk
                    out, n = C.c_void_p(), C.c_size_t()
                    if fn(blob, len(blob), C.byref(out), C.byref(n), 4):
                        got = C.string_at(out, n.value)
                        libc.free(out)
                        assert got == data, (kind, level, strategy, chunk)
                        taken += 1
    finally:
        os.environ.pop("CRASS_PGZIP_MIN_BYTES", None); os.environ.pop("CRASS_PGZIP_CHUNK_BYTES", None)
    print("several-thread inflate taken in %d of 48 cases" % taken)
    assert taken >= 20, taken


def test_indexed_reader_refuses_mixed_comments(ca, tmp_path):
    """kseq's stale comment / quality buffers (libcrispr.cpp:124-131) make a record's fields depend on the records before it: a
    file that mixes records with and without a comment is left to the ordered readers"""
    text = _fastx_cases()["fa_single"]
    plain, _ = _write_both(tmp_path, "mixed.fa", text)
    with pytest.raises(ca.CrassError) as e:
        ca.FastxIndex(plain)
    assert e.value.status == 2


def test_streaming_reader_on_the_reference_inputs(ca):
    """the five regression inputs of the reference (tests/golden/data, byte-identical to /root/reference/test/*.gz) through the
    stream in small chunks = through the whole-file reader"""
    for fname in ["Ill100.fx.gz", "CN_gDC.fa.gz", "front_offset_bug.fa.gz", "Ill.nr.miss.fa.gz", "poor_dr_ext.fa.gz"]:
        path = os.path.join(DATA, fname)
        f = ca.FastxFile(path)
        recs, hid, last, chunks = ca.stream_fastx(path, chunk_bytes=20000)
        assert recs == f.records() and hid == f.header_id.tolist() and last == f.last_ret and (chunks > 1 or f.n_reads < 50)


def test_gzip_inflate_paths_agree(ca, tmp_path):
    """.gz inputs: the libdeflate fast path (whole-buffer, member by member), zlib's gzread (CRASS_NO_LIBDEFLATE) and the
    plain file give the same records — single member, several concatenated members (gzread semantics: one stream),
    an output far larger than the first size guess, and damaged data (both paths: the same error)."""
    import gzip, zlib
    text = _fastx_cases()["fq"]
    plain = tmp_path / "m.fq"
    plain.write_bytes(text)
    single = tmp_path / "single.fq.gz"
    single.write_bytes(gzip.compress(text, 1))
    cut = [0, len(text) // 3, len(text) // 3 + 1, len(text)]            # member boundaries in the middle of records
    multi = tmp_path / "multi.fq.gz"
    multi.write_bytes(b"".join(gzip.compress(text[a:b], 6) for a, b in zip(cut, cut[1:])))
    zeros = tmp_path / "ratio.fa.gz"                                     # compresses 1000:1: the first buffer guess is too small
    big = b">z\n" + b"A" * 3_000_000 + b"\n"
    zeros.write_bytes(gzip.compress(big, 9))
    ref = ca.FastxFile(plain).records()
    try:
        for env in (None, "1"):
            if env: os.environ["CRASS_NO_LIBDEFLATE"] = env
            else: os.environ.pop("CRASS_NO_LIBDEFLATE", None)
            assert ca.FastxFile(single).records() == ref
            assert ca.FastxFile(multi).records() == ref
            z = ca.FastxFile(zeros)
            assert z.n_reads == 1 and z.max_len == 3_000_000
            bad = bytearray(single.read_bytes())
            bad[len(bad) // 2] ^= 0x55
            broken = tmp_path / "broken.fq.gz"
            broken.write_bytes(bytes(bad))
            with pytest.raises(ca.CrassError):
                ca.FastxFile(broken)
    finally:
        os.environ.pop("CRASS_NO_LIBDEFLATE", None)


def test_header_ids_many_threads(ca, tmp_path):
    """Enough records that the name table is filled by several threads at once; a third of the names repeat
    (readsFound is keyed by the header string, crass WorkHorse.cpp:1017)."""
    import random
    rng = random.Random(5)
    n = 400_000
    names = [f"r{rng.randrange(n // 3)}" if i % 3 == 0 else f"u{i}" for i in range(n)]
    p = tmp_path / "dups.fa"
    with open(p, "w") as fh:
        fh.write("".join(f">{nm}\nACGTACGTAC\n" for nm in names))
    first = {}
    ref_ids = np.array([first.setdefault(nm, i) for i, nm in enumerate(names)], dtype=np.uint64)
    f = ca.FastxFile(str(p))
    assert f.n_reads == n
    assert np.array_equal(np.asarray(f.header_id), ref_ids)
    # the indexed reader's table: 256 shards by the hash's top byte, each filled by one thread in read order
    ix = ca.FastxIndex(str(p))
    from crass_amd.engine import _npv
    assert ix.n_reads == n and ix.reads.header_id
    assert np.array_equal(_npv(ix.reads.header_id, n, np.uint64), ref_ids)
    ix.close()
    # the streamed reader's table: 64 sub-tables, a chunk's names dealt to them in read order and every sub-table filled by one
    # thread; chunks of ~2 MB (a batch per chunk) and one chunk for the whole file
    for chunk in (2 << 20, 0):
        recs, hid, last, chunks = ca.stream_fastx(str(p), chunk_bytes=chunk)
        assert len(recs) == n and np.array_equal(np.array(hid, dtype=np.uint64), ref_ids)


@pytest.mark.parametrize("kind", ["fa_plain", "fa_all_comments", "fq_plain", "fq_all_comments", "fa_comments_stop", "fq_multi"])
@pytest.mark.parametrize("chunk", [4096, 100000, 0])
def test_streaming_reader_block_assembly(ca, tmp_path, kind, chunk):
    """the chunk assembly's block form (no record of the stream has a comment / quality, or every record has its own: four block
    copies per piece, the pieces side by side) against the record-by-record walk (CRASS_STREAM_SERIAL_ASSEMBLE) and kseq;
    a file whose comments stop half-way (later records inherit the last one: the walk) starts in the block form and leaves it"""
    import random
    rng = random.Random(23)
    recs = []
    for i in range(6000):
        sq = "".join(rng.choice("ACGT") for _ in range(rng.randint(20, 200)))
        com = " lane=%d x" % (i % 7) if kind in ("fa_all_comments", "fq_all_comments") or (kind == "fa_comments_stop" and i < 3000) else ""
        if kind.startswith("fa"):
            recs.append(">n%d%s\n%s\n" % (i, com, sq))
        else:
            q = "".join(rng.choice("IIIHG5#!~") for _ in sq)
            body = "\n".join(sq[k:k + 70] for k in range(0, len(sq), 70)) if kind == "fq_multi" else sq
            recs.append("@n%d%s\n%s\n+\n%s\n" % (i, com, body, q))
    text = "".join(recs).encode()
    plain, gz = _write_both(tmp_path, kind + ".txt", text)
    ref = fastx.read_fastx(gz)
    assert len(ref) == 6000
    for path in (plain, gz):
        got = ca.stream_fastx(path, chunk_bytes=chunk)
        os.environ["CRASS_STREAM_SERIAL_ASSEMBLE"] = "1"
        try:
            walk = ca.stream_fastx(path, chunk_bytes=chunk)
        finally:
            os.environ.pop("CRASS_STREAM_SERIAL_ASSEMBLE", None)
        assert got[0] == ref and walk[0] == ref
        assert got[1] == walk[1] == list(range(6000))
        assert got[2] == walk[2]


def test_command_line_option_checks_need_no_gpu(ca, tmp_path):
    """crass-hip validates its options the way crass.cpp:264-400 does before anything touches a device: usage, the missing input,
    bounds that cross, values crass replaces with its defaults (with crass's warnings) — all without a GPU; a run that gets as far as
    the search on a host without one fails loudly (there is no CPU path)"""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.abspath(ca.__file__)), "crass-hip")
    if not os.path.exists(cli):
        pytest.skip("crass-hip not built")
    run = lambda *a: subprocess.run([cli] + list(a), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    r = run("-h")
    assert r.returncode == 0 and "usage: crass-hip" in r.stdout
    r = run()
    assert r.returncode == 1 and "No input files were provided" in r.stderr
    r = run("-d", "50", "-D", "40", "x.fa")
    assert r.returncode == 1 and "lower direct repeat bound is bigger" in r.stderr
    r = run("-s", "60", "-S", "50", "x.fa")
    assert r.returncode == 1 and "lower spacer bound is bigger" in r.stderr
    r = run("-n", "1", "x.fa")
    assert r.returncode == 1 and "cannot be less than 2" in r.stderr
    fa = tmp_path / "r.fa"
    fa.write_text(">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    import torch
    if torch.cuda.device_count() == 0:
        r = run("-g", "-d", "5", "-w", "12", "-o", str(tmp_path), str(fa))      # (-d < 8 and -w outside 6..9: warned about and replaced)
        assert "changing to 23" in r.stderr and "Changing window length to 8" in r.stderr
        assert r.returncode != 0                             # no device: the search stage throws, the process reports it
