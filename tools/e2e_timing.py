#!/usr/bin/env python3
"""End-to-end timing from a FASTA file on disk: read/parse (kseq semantics) -> 2-bit pack -> H2D -> pass 1 +
merge + pass 2 -> records on the host.  Reported per stage ('also report end-to-end incl.
parse/pack/H2D separately')."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = 150
spec = ca.synth_spec(read_len=L)
words = ca.synth_packed(spec, 0, n)
asc = ca.unpack_ascii(words, 10, L, n)
path = os.path.join(tempfile.gettempdir(), "e2e_%d.fa" % n)
t0 = time.time()
with open(path, "wb") as f:
    rec = np.empty((n, 8 + 1 + L + 1), np.uint8)          # ">rNNNNNN\nSEQ\n" fixed-width records
    ids = np.char.zfill(np.arange(n).astype("S7"), 7)
    rec[:, 0] = ord(">"); rec[:, 1] = ord("r")
    rec[:, 2:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, 7)[:, :7]
    rec[:, 9] = 10
    rec[:, 10:10 + L] = asc.reshape(n, L)
    rec[:, 10 + L] = 10
    f.write(rec.tobytes())
print("wrote %s (%.1f MB) in %.1fs" % (path, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
for it in range(2):
    t = [time.perf_counter()]
    fx = ca.FastxFile(path); t.append(time.perf_counter())
    seqs = None
    packed = fx.packed() if hasattr(fx, "packed") else None
    t.append(time.perf_counter())
    eng = ca.SearchEngine(device=0)
    if packed is None:
        recs = fx.records()
        packed = ca.PackedReads([r[2] for r in recs])
    eng.load_reads(packed, None); t.append(time.perf_counter())
    eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False); t.append(time.perf_counter())
    c = eng.counters()
    print("run %d: parse %.3fs  pack %.3fs  H2D+alloc %.3fs  search %.4fs  total %.3fs  -> %.2f M reads/s end to end; found %d + %d" % (
        it, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0], n / (t[4] - t[0]) / 1e6, c["n_pass1_found"], c["n_pass2_found"]), flush=True)
    eng.close()
os.remove(path)
