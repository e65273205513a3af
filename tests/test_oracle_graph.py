"""SURVEY 8f rows f-4 / f-3 (spacer graph, crass.crispr, Group_*.fa): the product (crass_build_outputs in libcrass_hip.so,
host code) against the restatement oracle/crass_graph.py, byte for byte, on the reference's regression inputs (hand-off from
the oracle's search + consensus stages: everything here runs without a GPU) and on randomised CRISPR loci with sequencing
errors, variant spacers and partial coverage (forks, bubbles, cross nodes, flankers).  PARITY UNPINNED against the
reference itself (see the oracle's header): what is pinned are SURVEY 8c's per-group read counts, structural invariants and
the XML's well-formedness."""
import os
import random
import sys
import xml.etree.ElementTree as ET

import pytest

from tests import orc, fastx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import crass_graph as cg  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
OPTS = dict(timestamp="03_10_2026_120000", cmdline="crass -o out reads.fa ", cwd="/work/dir")


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    from crass_amd import build
    build.build()
    crass_amd.load()
    return crass_amd


def both(ca, groups, outdir="out/", log_to_screen=True):
    o = cg.run([(g, d, [cg.Read(*r) for r in rs]) for g, d, rs in groups], outdir=outdir, log_to_screen=log_to_screen, **OPTS)
    files, kept, text = ca.build_outputs(groups, out_dir=outdir, timestamp=OPTS["timestamp"], command_line=OPTS["cmdline"], cwd=OPTS["cwd"],
                                         log_to_screen=log_to_screen)
    assert kept == o["kept"]
    assert sorted(files) == sorted(o["files"])
    for name in files:
        assert files[name] == o["files"][name], name
    assert text.splitlines() == o["stdout"]
    # ... and crass_outputs_write puts the same bytes on disk (every file cut into slices that several threads write)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        files2, _, _ = ca.build_outputs(groups, out_dir=outdir, timestamp=OPTS["timestamp"], command_line=OPTS["cmdline"], cwd=OPTS["cwd"],
                                        log_to_screen=log_to_screen, write_to=td)
        assert sorted(os.listdir(td)) == sorted(files2)
        for name in files2:
            with open(os.path.join(td, name), "rb") as f:
                assert f.read() == files[name], name
    return o, files


def check_xml(files, groups, kept):
    root = ET.fromstring(files["crass.crispr"].decode("latin-1"))
    assert root.tag == "crispr" and root.attrib == {"version": "1.1"}
    gs = root.findall("group")
    assert [g.attrib["gid"] for g in gs] == ["G%d" % k for k in kept]
    dr_of = {g: d for g, d, _ in groups}
    hdrs_of = {g: {r[0] for r in rs} for g, _, rs in groups}
    for g, k in zip(gs, kept):
        assert g.attrib["drseq"] == dr_of[k].decode()
        assert [c.tag for c in g] == ["data", "metadata", "assembly"]
        data = g.find("data")
        assert [c.tag for c in data][:3] == ["sources", "drs", "spacers"]
        so = {s.attrib["soid"]: s.attrib["accession"] for s in data.find("sources")}
        assert set(so.values()) <= {h.decode("latin-1") for h in hdrs_of[k]}
        spids = set()
        for sp in data.find("spacers"):
            assert set(sp.attrib) == {"cov", "seq", "spid"} and int(sp.attrib["cov"]) >= 1
            assert all(s.attrib["soid"] in so for s in sp)
            spids.add(sp.attrib["spid"])
        fl = data.find("flankers")
        flids = {f.attrib["flid"] for f in fl} if fl is not None else set()
        # every spacer named in the assembly is a spacer or flanker of <data>, and sits in exactly one contig
        seen = []
        for c in g.find("assembly"):
            for cs in c:
                seen.append(cs.attrib["spid"])
                assert cs.attrib["spid"] in spids | flids
        assert len(seen) == len(set(seen))
        # the reads file holds reads of this group only, each once, in hand-off order
        fa = files["Group_%d_%s.fa" % (k, dr_of[k].decode())]
        names = [l[1:].split(b" ")[0] for l in fa.split(b"\n") if l.startswith(b">")]
        order = [r[0] for r in dict((g2, rs) for g2, _, rs in groups)[k]]
        assert set(names) <= set(order)
        it = iter(order)
        assert all(any(n == o for o in it) for n in names)          # a subsequence of the hand-off order


KNOWN_GROUP_READS = {"front_offset_bug.fa.gz": {10: 303, 20: 102, 44: 35, 45: 28}, "Ill100.fx.gz": {1: 4312}}


@pytest.mark.parametrize("fname", sorted(f for f in os.listdir(DATA) if f.endswith(".gz")))
def test_reference_inputs_product_equals_oracle(ca, fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs, hdrs = [x[2] for x in recs], [x[0] for x in recs]
    ref = orc.pipeline(seqs, hdrs)
    con = orc.consensus(seqs, ref)
    groups = orc.graph_groups(recs, ref, con)
    if fname in KNOWN_GROUP_READS:
        assert {g: len(rs) for g, _, rs in groups} == KNOWN_GROUP_READS[fname]
    o, files = both(ca, groups)
    check_xml(files, groups, o["kept"])
    both(ca, groups, outdir="/abs/dir/", log_to_screen=False)


def locus_reads(rng, n_spacers=25, n_reads=400, L=150, err=0.0, variants=0.0):
    """reads sampled from one synthetic CRISPR locus; start/stops are the true DR positions inside each read (full copies only)"""
    def rs(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    dr = rs(rng.randint(28, 36))
    spacers = [rs(rng.randint(30, 38)) for _ in range(n_spacers)]
    alt = {i: rs(len(spacers[i])) for i in range(n_spacers) if rng.random() < variants}
    reads = []
    for k in range(n_reads):
        use = [alt[i] if (i in alt and rng.random() < 0.4) else spacers[i] for i in range(n_spacers)]
        genome, pos = rs(60), []
        for sp in use:
            pos.append(len(genome))
            genome += dr + sp
        pos.append(len(genome))
        genome += dr + rs(60)
        a = rng.randrange(0, len(genome) - L)
        seq = bytearray(genome[a:a + L])
        ss = []
        for p in pos:
            if p >= a and p + len(dr) <= a + L:
                ss += [p - a, p - a + len(dr) - 1]
        if len(ss) < 2:
            continue
        for i in range(L):                                   # substitutions outside the repeats
            if rng.random() < err and not any(ss[j] <= i <= ss[j + 1] for j in range(0, len(ss), 2)):
                seq[i] = rng.choice(b"ACGT")
        reads.append((b"read%d" % k, b"c%d" % k if k % 7 == 0 else None, bytes(seq), ss))
    return dr, reads


@pytest.mark.parametrize("seed", range(12))
def test_random_loci_product_equals_oracle(ca, seed):
    rng = random.Random(seed)
    groups = []
    for g in range(rng.randint(1, 4)):
        dr, reads = locus_reads(rng, n_spacers=rng.randint(4, 30), n_reads=rng.randint(20, 500), err=rng.choice([0, 0.002, 0.01]),
                                variants=rng.choice([0, 0.1, 0.3]))
        groups.append((3 * g + 1, dr, reads))
    o, files = both(ca, groups)
    check_xml(files, groups, o["kept"])


def test_forks_and_bubbles_are_exercised(ca):
    """the randomised loci do reach the order-sensitive parts of the graph code"""
    rng = random.Random(5)
    dr, reads = locus_reads(rng, n_spacers=20, n_reads=600, err=0.01, variants=0.3)
    out = []
    nm = cg.NodeManager(dr, out)
    for r in reads:
        nm.add_read(cg.Read(*r))
    n_nodes = len(nm.nodes)
    nm.clean_graph()
    assert sum(1 for n in nm.nodes.values() if not n.attached) > 0 and n_nodes > 40
    nm.build_spacer_graph()
    assert max(s.rank() for s in nm.spacers.values()) > 2           # cross nodes before the spacer graph is cleaned
    nm.clean_spacer_graph()
    nm.split_into_contigs()
    assert nm.next_contig >= 1


def test_empty_and_tiny_inputs(ca):
    o, files = both(ca, [])
    assert o["kept"] == [] and files["crass.crispr"] == b'<?xml version="1.0" encoding="ISO8859-1" standalone="no" ?>\n<crispr version="1.1"/>\n'
    dr, reads = locus_reads(random.Random(1), n_spacers=2, n_reads=3)
    both(ca, [(1, dr, reads)])


def test_xml_escapes_and_attribute_order(ca):
    dr, reads = locus_reads(random.Random(2), n_spacers=12, n_reads=300)
    reads = [(h + b'&<">', c, s, ss) for h, c, s, ss in reads]
    o, files = both(ca, [(7, dr, reads)])
    assert o["kept"] == [7]
    x = files["crass.crispr"].decode("latin-1")
    assert 'accession="read' in x and "&amp;&lt;&quot;>" in x
    assert '<group drseq="%s" gid="G7">' % dr.decode() in x and '<dr drid="DR1" seq="' in x
    ET.fromstring(x)


def test_outputs_write_reports_a_directory_it_cannot_write_to(ca, tmp_path):
    """crass_outputs_write creates and writes every file on its own threads: a directory that does not exist is CRASS_ERR_IO (the
    reference's writers fail the same way, WorkHorse.cpp:1900-2038), and nothing is left half-written elsewhere"""
    groups = [(1, b"GTTTCAATCCACGCGCCCACGCGGGGCGCGAC", [])]
    with pytest.raises(ca.CrassError):
        ca.build_outputs(groups, out_dir="out/", timestamp=OPTS["timestamp"], command_line=OPTS["cmdline"], cwd=OPTS["cwd"],
                         log_to_screen=True, write_to=str(tmp_path / "no" / "such" / "dir"))
    files, kept, _ = ca.build_outputs(groups, out_dir="out/", timestamp=OPTS["timestamp"], command_line=OPTS["cmdline"], cwd=OPTS["cwd"],
                                      log_to_screen=True, write_to=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(files)
