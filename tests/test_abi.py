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
    assert ca.load().crass_hip_abi_version() == 1


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
