#!/usr/bin/env python3
"""crass_index_fastx alone (no GPU): a FASTA of N reads of bench.py's stream in tmpfs, indexed K times with CRASS_TIMING=1.
   python tools/index_time.py [reads] [repeats] [x = delete the file afterwards] [fq = FASTQ instead of FASTA]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRASS_TIMING"] = "1"
import numpy as np
import crass_amd as ca
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 150
fastq = "fq" in sys.argv[3:]
fa = "/dev/shm/crass_index_time_%d.%s" % (n, "fq" if fastq else "fa")
if not os.path.exists(fa):
    spec = ca.synth_spec(read_len=L)
    with open(fa, "wb") as f:
        for first in range(0, n, 5_000_000):
            m = min(5_000_000, n - first)
            w = ca.synth_packed(spec, first, m)
            asc = ca.unpack_ascii(w, (L + 15) // 16, L, m).reshape(m, L)
            ids = np.char.zfill(np.arange(first, first + m).astype("S8"), 8)
            rec = np.empty((m, 10 + L + 1 + ((2 + L + 1) if fastq else 0)), np.uint8)
            rec[:, 0] = ord("@" if fastq else ">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
            rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
            if fastq:
                rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10
                rec[:, 13 + L:13 + 2 * L] = 33 + (np.arange(L, dtype=np.uint8)[None, :] * 7 + (np.arange(m, dtype=np.uint32)[:, None] & 31).astype(np.uint8)) % 40
                rec[:, 13 + 2 * L] = 10
            f.write(rec.tobytes())
for _ in range(K):
    t0 = time.perf_counter()
    ix = ca.FastxIndex(fa)
    t1 = time.perf_counter()
    ix.close()
    print("index %.3f s, free %.3f s" % (t1 - t0, time.perf_counter() - t1), flush=True)
if "x" in sys.argv[3:]:
    os.unlink(fa)
