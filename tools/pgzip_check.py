#!/usr/bin/env python3
"""the several-thread inflate of one gzip member (csrc/pgzip.cpp) against the serial path on the same file: same records, and the time of
each:  python tools/pgzip_check.py [reads] [fq]"""
import os, sys, time, gzip, subprocess
sys.path.insert(0,'/root/repo')
os.environ["CRASS_TIMING"]="1"
import numpy as np
import crass_amd as ca
ca.load()
n=int(sys.argv[1]) if len(sys.argv)>1 else 3000000
L=150
fq = len(sys.argv)>2 and sys.argv[2]=="fq"
spec = ca.synth_spec(read_len=L)
p="/dev/shm/pgz_%d%s"%(n, ".fq" if fq else ".fa")
with open(p,"wb") as f:
    for first in range(0,n,5000000):
        m=min(5000000,n-first)
        asc = ca.unpack_ascii(ca.synth_packed(spec, first, m), (L + 15) // 16, L, m).reshape(m, L)
        rec = np.empty((m, 10 + L + 1 + ((2+L+1) if fq else 0)), np.uint8)
        ids = np.char.zfill(np.arange(first,first+m).astype("S8"), 8)
        rec[:, 0] = ord("@" if fq else ">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
        rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
        if fq:
            rec[:, 11+L]=ord("+"); rec[:,12+L]=10
            rec[:, 13+L:13+2*L] = 33 + (np.arange(L,dtype=np.uint8)[None,:]*7 + (np.arange(m,dtype=np.uint32)[:,None]&31).astype(np.uint8))%40
            rec[:, 13+2*L]=10
        f.write(rec.tobytes())
for lvl in (1,6):
    subprocess.check_call("gzip -%d -c %s > %s.gz"%(lvl,p,p), shell=True)
    for env in ({}, {"CRASS_NO_PGZIP":"1"}):
        os.environ.pop("CRASS_NO_PGZIP",None); os.environ.update(env)
        t0=time.perf_counter(); f=ca.FastxFile(p+".gz"); t1=time.perf_counter()
        print("level",lvl,env, f.n_reads, "%.2f s"%(t1-t0), flush=True)
        if not env: a=(f.seq.tobytes(), f.name.tobytes())
        else: assert a==(f.seq.tobytes(), f.name.tobytes()), "MISMATCH"
os.unlink(p); os.unlink(p+".gz")
