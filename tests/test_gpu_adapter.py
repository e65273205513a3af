"""The C++ adapter (reference seam: searchFile / createNonRedundantSet / findSingletons) driven
through the `crass-hip` command line on the reference's own regression inputs; its hand-off dump
is compared field by field with the oracle (tokens, groups, k-mer counts, patterns, every
ReadHolder incl. the possibly reverse-complemented sequence, comment and quality)."""
import collections
import os
import subprocess

import pytest

from tests import orc, fastx

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
RC = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


@pytest.fixture(scope="module")
def cli():
    from crass_amd import build
    return build.build_adapter()


def parse_handoff(path):
    out = dict(tokens={}, groups={}, kmers=collections.defaultdict(dict), patterns=[], reads=collections.defaultdict(list), meta={})
    for line in open(path, "rb").read().split(b"\n"):
        if not line:
            continue
        f = line.split(b"\t")
        if f[0].startswith(b"#"):
            out["meta"][f[0][1:].decode()] = int(f[1])
        elif f[0] == b"T":
            out["tokens"][int(f[1])] = f[2]
        elif f[0] == b"G":
            out["groups"][int(f[1])] = [int(x) for x in f[2:]]
        elif f[0] == b"K":
            out["kmers"][int(f[1])][f[2]] = int(f[3])
        elif f[0] == b"P":
            out["patterns"].append(f[1])
        elif f[0] == b"R":
            ss = [int(x) for x in f[6].split(b",")]
            out["reads"][int(f[1])].append(dict(header=f[2], low=int(f[3]), replen=int(f[4]), fasta=int(f[5]), ss=ss,
                                                seq=f[7], comment=f[8], qual=f[9] if len(f) > 9 else b""))
    return out


# the one-call device path (default), crass's three calls (--seam), and both over a group of contexts sharing the one GPU
MODES = {"device": [], "seam": ["--seam"], "group3": ["--devices", "0,0,0", "--local-copies"],
         "seam-group2": ["--seam", "--devices", "0,0", "--local-copies"],
         # streamed ingest (bounded host memory): the inputs are read in chunks — here far smaller than in production, so that
         # records, stale comments and duplicate headers straddle chunk ends — and a second time for the hand-off's text
         "stream": [], "stream-group3": ["--devices", "0,0,0", "--local-copies"]}
MODE_ENV = {"stream": {"CRASS_INGEST": "stream", "CRASS_INGEST_CHUNK_BYTES": "30000"},
            "stream-group3": {"CRASS_INGEST": "stream", "CRASS_INGEST_CHUNK_BYTES": "7000"}}


def mode_env(mode):
    return dict(os.environ, **MODE_ENV.get(mode, {"CRASS_INGEST": "whole"}))


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "front_offset_bug.fa.gz", "CN_gDC.fa.gz", "poor_dr_ext.fa.gz"])
def test_cli_handoff_matches_oracle(cli, tmp_path, fname, mode):
    path = os.path.join(DATA, fname)
    r = subprocess.run([cli, "--dump-handoff", "-o", str(tmp_path)] + MODES[mode] + [path], capture_output=True, timeout=300, env=mode_env(mode))
    assert r.returncode == 0, r.stderr.decode()
    so = r.stdout.decode()
    assert "[crass_patternFinder]: Processed" in so and "[crass_clusterCore]:" in so and "[crass_singletonFinder]:" in so
    h = parse_handoff(os.path.join(str(tmp_path), "crass_hip_handoff.tsv"))
    recs = fastx.read_fastx(path)
    ref = orc.pipeline([x[2] for x in recs], [x[0] for x in recs])
    assert ("Found %d reads" % (ref.n_pass1 + ref.n_pass2)) in so
    assert ("%d non-redundant patterns." % ref.n_patterns) in so
    assert h["meta"]["max_read_length"] == ref.max_read_len
    assert h["meta"]["next_free_GID"] == ref.n_groups + 1
    assert [h["tokens"][t] for t in sorted(h["tokens"])] == ref.tokens and min(h["tokens"]) == 2
    assert [h["groups"][g] for g in sorted(h["groups"])] == ref.groups
    assert sorted(h["patterns"]) == sorted(ref.patterns)
    # group k-mer counts: laurenized 11-mers of every member (WorkHorse.cpp:1547-1560,1620-1625)
    for gi, toks in enumerate(ref.groups):
        want = collections.Counter()
        for t in toks:
            s = ref.tokens[t - 2]
            for i in range(len(s) - 10):
                km = s[i:i + 11]
                rc = km.translate(RC)[::-1]
                want[min(km, rc)] += 1
        assert h["kmers"][gi + 1] == dict(want)
    # mReads[token]: pass-1 reads in read order, then pass-2 recruits in read order
    want_reads = collections.defaultdict(list)
    for k in range(ref.n_pass1 + ref.n_pass2):
        i = int(ref.rec_read[k])
        name, comment, seq, qual = recs[i]
        low = int(ref.rec_lowlexi[k])
        want_reads[int(ref.rec_token[k])].append(dict(
            header=name, low=low, replen=int(ref.rec_replen[k]), fasta=0 if qual is not None else 1, ss=ref.ss(k),
            seq=seq if low else seq.translate(RC)[::-1], comment=comment or b"", qual=qual or b""))
    assert set(h["reads"]) == set(want_reads)
    for t in want_reads:
        assert h["reads"][t] == want_reads[t], t


def test_cli_option_validation(cli, tmp_path):
    path = os.path.join(DATA, "Ill.nr.miss.fa.gz")
    assert subprocess.run([cli, "-n", "1", path], capture_output=True).returncode == 1
    assert subprocess.run([cli, "-d", "50", "-D", "40", path], capture_output=True).returncode == 1
    assert subprocess.run([cli], capture_output=True).returncode == 1
    r = subprocess.run([cli, "-o", str(tmp_path), str(tmp_path / "missing.fa")], capture_output=True)
    assert r.returncode == 2 and b"Could not open FASTQ" in r.stderr
    # -d 8 -w 9 passes the reference's own validation but leaves its seed stride undefined (unsigned wrap, libcrispr.cpp:281):
    # warned about and clamped to 2w - 1 = 17; the run then equals an explicit -d 17 -w 9 run
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = subprocess.run([cli, "--dump-handoff", "-d", "8", "-w", "9", "-o", str(tmp_path / "a"), path], capture_output=True, timeout=300)
    rb = subprocess.run([cli, "--dump-handoff", "-d", "17", "-w", "9", "-o", str(tmp_path / "b"), path], capture_output=True, timeout=300)
    assert ra.returncode == 0 and rb.returncode == 0 and b"changing to 17" in ra.stderr and b"changing to" not in rb.stderr
    assert open(tmp_path / "a" / "crass_hip_handoff.tsv", "rb").read() == open(tmp_path / "b" / "crass_hip_handoff.tsv", "rb").read()


@pytest.mark.parametrize("mode", sorted(MODES))
def test_cli_two_files_cross_file_headers(cli, tmp_path, mode):
    """readsFound is shared across files (WorkHorse.cpp:329-393): a header found in file 1's pass 1
    suppresses recruitment of the same header in file 2."""
    f1 = os.path.join(DATA, "Ill.nr.miss.fa.gz")
    r = subprocess.run([cli, "--dump-handoff", "-o", str(tmp_path)] + MODES[mode] + [f1, f1], capture_output=True, timeout=300, env=mode_env(mode))
    assert r.returncode == 0, r.stderr.decode()
    h = parse_handoff(os.path.join(str(tmp_path), "crass_hip_handoff.tsv"))
    recs = fastx.read_fastx(f1)
    seqs = [x[2] for x in recs] * 2
    hdrs = [x[0] for x in recs] * 2
    ref = orc.pipeline(seqs, hdrs)
    n = sum(len(v) for v in h["reads"].values())
    assert n == ref.n_pass1 + ref.n_pass2
    assert [h["tokens"][t] for t in sorted(h["tokens"])] == ref.tokens
    assert [h["groups"][g] for g in sorted(h["groups"])] == ref.groups and sorted(h["patterns"]) == sorted(ref.patterns)
    got = {t: [(x["header"], x["low"], x["ss"]) for x in v] for t, v in h["reads"].items()}
    want = collections.defaultdict(list)
    for k in range(ref.n_pass1 + ref.n_pass2):
        want[int(ref.rec_token[k])].append((hdrs[int(ref.rec_read[k])], int(ref.rec_lowlexi[k]), ref.ss(k)))
    assert got == dict(want)


def parse_consensus(path):
    out = dict(tokens={}, groups={}, drs={}, reads=collections.defaultdict(list), meta={})
    for line in open(path, "rb").read().split(b"\n"):
        if not line:
            continue
        f = line.split(b"\t")
        if f[0].startswith(b"#"):
            out["meta"][f[0][1:].decode()] = int(f[1])
        elif f[0] == b"T":
            out["tokens"][int(f[1])] = f[2]
        elif f[0] == b"D":
            out["drs"][int(f[1])] = f[2]
        elif f[0] == b"G":
            out["groups"][int(f[1])] = [int(x) for x in f[2:]]
        elif f[0] == b"R":
            out["reads"][int(f[1])].append(dict(header=f[2], low=int(f[3]), ss=[int(x) for x in f[4].split(b",")] if f[4] else [], seq=f[5]))
    return out


@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "front_offset_bug.fa.gz", "CN_gDC.fa.gz", "poor_dr_ext.fa.gz"])
def test_cli_consensus_matches_oracle(cli, tmp_path, fname):
    """findConsensusDRs through the adapter (the reference's function shape over crass_hip_consensus): true DRs, groups,
    tokens and every surviving ReadHolder (orientation, repaired start/stops, sequence) against the oracle"""
    path = os.path.join(DATA, fname)
    r = subprocess.run([cli, "--dump-handoff", "-o", str(tmp_path), path], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    h = parse_consensus(os.path.join(str(tmp_path), "crass_hip_consensus.tsv"))
    recs = fastx.read_fastx(path)
    seqs, hdrs = [x[2] for x in recs], [x[0] for x in recs]
    ref = orc.pipeline(seqs, hdrs)
    con = orc.consensus(seqs, ref)
    assert con.error == 0
    assert ("[crass_consensus]: %d true direct repeats" % len(con.gids)) in r.stdout.decode()
    assert h["meta"]["next_free_GID"] == con.next_free_gid
    assert [h["tokens"][t] for t in sorted(h["tokens"])] == con.tokens
    assert h["drs"] == dict(zip(con.gids, con.true_drs))
    assert h["groups"] == dict(zip(con.gids, con.groups))
    want = collections.defaultdict(list)
    for t, lst in enumerate(con.reads_of):
        for k in lst or []:
            i = int(ref.rec_read[k])
            rc = int(con.rec_rc[k])
            want[t + 2].append(dict(header=hdrs[i], low=1 - rc, ss=con.ss(k), seq=seqs[i].translate(RC)[::-1] if rc else seqs[i]))
    assert {t: v for t, v in h["reads"].items() if v} == {t: v for t, v in want.items() if v}


@pytest.mark.parametrize("log_to_screen", [True, False])
@pytest.mark.parametrize("fname", ["Ill100.fx.gz", "front_offset_bug.fa.gz", "CN_gDC.fa.gz", "poor_dr_ext.fa.gz", "Ill.nr.miss.fa.gz"])
def test_cli_writes_crass_outputs_equal_to_the_oracle(cli, tmp_path, fname, log_to_screen):
    """SURVEY 8f rows f-4 / f-3 end to end: `crass-hip -o dir reads` leaves crass.crispr, Group_<gid>_<DR>.fa, the spacer graph
    files and the key file in dir — byte for byte what oracle/crass_graph.py derives from the oracle's search + consensus"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import crass_graph as cg
    path = os.path.join(DATA, fname)
    out = str(tmp_path / "o") + "/"
    os.mkdir(out)
    stamp = "03_10_2026_101500"
    cmd = [cli, "--timestamp", stamp, "-o", out] + (["-g"] if log_to_screen else []) + [path]
    r = subprocess.run(cmd, capture_output=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()
    recs = fastx.read_fastx(path)
    seqs, hdrs = [x[2] for x in recs], [x[0] for x in recs]
    ref = orc.pipeline(seqs, hdrs)
    con = orc.consensus(seqs, ref)
    groups = orc.graph_groups(recs, ref, con)
    want = cg.run([(g, d, [cg.Read(*x) for x in rs]) for g, d, rs in groups], outdir=out, timestamp=stamp, cmdline=" ".join(cmd) + " ",
                  cwd=str(tmp_path), log_to_screen=log_to_screen)
    got = {f: open(os.path.join(out, f), "rb").read() for f in os.listdir(out) if not f.endswith(".log")}
    assert sorted(got) == sorted(want["files"])
    for f in got:
        assert got[f] == want["files"][f], f
    assert ("[crass_graphBuilder]: %d CRISPRs found!" % len(want["kept"])) in r.stdout.decode()
    assert os.path.exists(os.path.join(out, "crass.%s.log" % stamp)) == (not log_to_screen)


def test_streamed_and_whole_ingest_write_the_same_files(cli, tmp_path):
    """a synthetic FASTQ with comments on some records, ragged lengths, N reads and duplicate headers: the complete command line
    (search, consensus, spacer graphs, .crispr + Group_*.fa) with streamed ingest (40 KB chunks) and with whole-file ingest —
    byte-identical output directories"""
    import crass_amd as ca
    import numpy as np
    ca.load()
    n, L = 30000, 150
    spec = ca.synth_spec(read_len=L, crispr_per_million=60000, n_dr=8)
    asc = ca.unpack_ascii(ca.synth_packed(spec, 0, n), (L + 15) // 16, L, n)
    rng = np.random.default_rng(5)
    lines = []
    for i in range(n):
        s = asc[i * L:(i + 1) * L].tobytes()
        if i % 11 == 0:
            s = s[:int(rng.integers(70, L))]
        if i % 501 == 0:
            s = s[:40] + b"N" + s[41:]
        name = b"r%d" % (i if i % 97 else i // 2)
        com = (b" c%d" % i) if (i > 30 and i % 4 == 0) else b""
        lines.append(b"@" + name + com + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    fq = tmp_path / "s.fq"
    fq.write_bytes(b"".join(lines))
    outs = {}
    for mode, env in (("whole", {"CRASS_INGEST": "whole"}), ("stream", {"CRASS_INGEST": "stream", "CRASS_INGEST_CHUNK_BYTES": "40000"})):
        d = tmp_path / mode
        d.mkdir()
        r = subprocess.run([cli, "-g", "--timestamp", "01_01_2026_000000", "-o", str(d), str(fq)], capture_output=True, timeout=600,
                           env=dict(os.environ, **env))                  # (a fixed mTimeStamp: it is part of a file name)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[mode] = {f: open(d / f, "rb").read() for f in sorted(os.listdir(d))}
        assert any(f.endswith(".crispr") for f in outs[mode]) and len(outs[mode]) >= 3
    assert outs["whole"].keys() == outs["stream"].keys()
    for f in outs["whole"]:
        a, b = outs["whole"][f], outs["stream"][f]
        # (the XML and the log record the output directory of the command line)
        a, b = a.replace(str(tmp_path / "whole").encode(), b"DIR"), b.replace(str(tmp_path / "stream").encode(), b"DIR")
        assert a == b, f


@pytest.mark.parametrize("kind", ["fastq_all_comments", "fasta_plain", "fasta_multiline", "fasta_gz", "fastq_gz", "fasta_two_files"])
def test_indexed_ingest_writes_the_same_files(cli, tmp_path, kind):
    """the indexed reader (crass_index_fastx: the input stays mapped, reads packed as their piece is parsed, the handed-on records'
    text parsed on request) against whole-file and streamed ingest through the complete command line: byte-identical output
    directories.  Ragged lengths, N reads, duplicate headers; FASTQ with a comment on every record, plain FASTA, multi-line FASTA;
    pieces of 20 KB so that the index is built from dozens of pieces.  Default mode (no CRASS_INGEST) takes the index for these."""
    import crass_amd as ca
    import numpy as np
    ca.load()
    n, L = 30000, 150
    spec = ca.synth_spec(read_len=L, crispr_per_million=60000, n_dr=8)
    asc = ca.unpack_ascii(ca.synth_packed(spec, 0, n), (L + 15) // 16, L, n)
    rng = np.random.default_rng(9)
    lines = []
    for i in range(n):
        s = asc[i * L:(i + 1) * L].tobytes()
        if i % 11 == 0:
            s = s[:int(rng.integers(70, L))]
        if i % 501 == 0:
            s = s[:40] + b"N" + s[41:]
        name = b"r%d" % (i if i % 97 else i // 2)
        if kind in ("fastq_all_comments", "fastq_gz"):
            lines.append(b"@" + name + (b" c%d" % i) + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
        elif kind in ("fasta_plain", "fasta_gz", "fasta_two_files"):
            lines.append(b">" + name + b"\n" + s + b"\n")
        else:
            lines.append(b">" + name + b"\n" + b"\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + b"\n")
    fx = tmp_path / (("s.fq" if kind.startswith("fastq") else "s.fa") + (".gz" if kind.endswith("_gz") else ""))
    inputs = [str(fx)]
    if kind == "fasta_two_files":                                # (paired-end style: one read set in (file, read) order, names shared across the files)
        import gzip
        fx2 = tmp_path / "s2.fa.gz"
        fx.write_bytes(b"".join(lines[:n // 2]))
        with gzip.open(fx2, "wb", compresslevel=1) as fh:
            fh.write(b"".join(lines[n // 2:]))
        inputs = [str(fx), str(fx2)]
    elif kind.endswith("_gz"):                                   # (gzip'd: the index is built over the inflated image)
        import gzip
        with gzip.open(fx, "wb", compresslevel=1) as fh:
            fh.write(b"".join(lines))
    else:
        fx.write_bytes(b"".join(lines))
    outs, errs = {}, {}
    for mode, env in (("whole", {"CRASS_INGEST": "whole"}), ("stream", {"CRASS_INGEST": "stream", "CRASS_INGEST_CHUNK_BYTES": "40000"}),
                      ("index", {"CRASS_INGEST": "index", "CRASS_FASTX_CHUNK": "20000"}), ("auto", {"CRASS_TIMING": "1"})):
        d = tmp_path / mode
        d.mkdir()
        e = dict(os.environ, **env)
        if mode == "auto":
            e.pop("CRASS_INGEST", None)
        r = subprocess.run([cli, "-g", "--timestamp", "01_01_2026_000000", "-o", str(d)] + inputs, capture_output=True, timeout=600, env=e)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[mode] = {f: open(d / f, "rb").read() for f in sorted(os.listdir(d))}
        errs[mode] = r.stderr.decode()
        assert any(f.endswith(".crispr") for f in outs[mode]) and len(outs[mode]) >= 3
    assert "indexed ingest" in errs["auto"]
    for mode in ("stream", "index", "auto"):
        assert outs["whole"].keys() == outs[mode].keys()
        for f in outs["whole"]:
            a, b = outs["whole"][f], outs[mode][f]
            a, b = a.replace(str(tmp_path / "whole").encode(), b"DIR"), b.replace(str(tmp_path / mode).encode(), b"DIR")
            assert a == b, (mode, f)

