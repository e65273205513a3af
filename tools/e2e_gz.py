#!/usr/bin/env python3
"""crass-hip end to end on a GZIP'd FASTA of N reads (default 10 M) and on the same reads as two files (plain + gzip'd), with the
default reader (the index over the inflated image / over both inputs) and the readers it replaced:  python tools/e2e_gz.py [reads]"""
import gzip, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
td = tempfile.mkdtemp(prefix="crass_e2e_gz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    spec = ca.synth_spec(read_len=L)
    fa, gz, h1, h2 = (os.path.join(td, x) for x in ("all.fa", "all.fa.gz", "half1.fa", "half2.fa.gz"))
    with open(fa, "wb") as f, open(h1, "wb") as f1:
        for first in range(0, n, 5_000_000):
            m = min(5_000_000, n - first)
            asc = ca.unpack_ascii(ca.synth_packed(spec, first, m), (L + 15) // 16, L, m).reshape(m, L)
            rec = np.empty((m, 10 + L + 1), np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("S8"), 8)
            rec[:, 0] = ord(">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
            rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
            b = rec.tobytes()
            f.write(b)
            if first < n // 2:
                f1.write(b)
    subprocess.check_call("gzip -1 -c %s > %s" % (fa, gz), shell=True)
    subprocess.check_call("tail -c +%d %s | gzip -1 -c > %s" % (os.path.getsize(h1) + 1, fa, h2), shell=True)
    # the same text as BGZF (bgzip's format; what Illumina's converters write): members of <= 64 KB that carry their own sizes
    import struct, zlib
    from concurrent.futures import ThreadPoolExecutor
    bg = os.path.join(td, "all.bgzf.fa.gz")
    def member(c):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        d = co.compress(c) + co.flush()
        return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
    with open(fa, "rb") as f, open(bg, "wb") as g, ThreadPoolExecutor(16) as ex:
        while True:
            blob = f.read(64 << 20)
            if not blob:
                break
            for m in ex.map(member, [blob[i:i + 60000] for i in range(0, len(blob), 60000)]):
                g.write(m)
        g.write(member(b""))
    cli = os.path.join(os.path.dirname(os.path.abspath(ca.__file__)), "crass-hip")
    cases = (("one gzip'd input", [gz], ("auto", "whole", "stream")), ("one BGZF input", [bg], ("auto", "whole")), ("two inputs (plain + gzip'd)", [h1, h2], ("auto", "whole", "stream")),
             ("the same reads, one plain input", [fa], ("auto",)))
    if os.environ.get("ONLY_AUTO"):                      # (big inputs: the default reader only)
        cases = tuple((l, i, ("auto",)) for l, i, m in cases)
    for label, inputs, modes in cases:
        for mode in modes:
            env = dict(os.environ, CRASS_TIMING="1")
            env.pop("CRASS_INGEST", None)
            if mode != "auto":
                env["CRASS_INGEST"] = mode
            walls, line = [], ""
            for _ in range(2):
                od = os.path.join(td, "out"); shutil.rmtree(od, ignore_errors=True); os.makedirs(od)
                t0 = time.perf_counter()
                r = subprocess.run([cli, "-g", "-o", od] + inputs, capture_output=True, env=env)
                walls.append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr.decode()[-1500:]
                line = [l for l in r.stderr.decode().splitlines() if "searchAndRecruit:" in l and "reads on" in l][0].split("searchAndRecruit: ")[1]
            print("%-34s reader %-6s wall %.2f s (%s)  %s" % (label, mode, min(walls), ", ".join("%.2f" % w for w in walls), line[:150]), flush=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
