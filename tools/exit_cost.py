#!/usr/bin/env python3
"""_exit -> reaped for a process holding host memory / device memory / pinned memory / a HIP context (profiles/ubench/exit_cost.cpp)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/exit_cost"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "profiles/ubench/exit_cost.cpp"), "-o", exe])
cases = [("nothing", "0 0 0 0 0"), ("HIP context only", "0 0 0 0 1"), ("6 GB host, 4 KB pages", "6 0 0 0 0"), ("6 GB host, THP", "6 1 0 0 0"),
         ("3 GB device", "0 0 3 0 1"), ("12 GB device", "0 0 12 0 1"), ("1 GB pinned", "0 0 0 1 1"), ("6 GB host 4 KB + 3 GB device + 0.25 GB pinned", "6 0 3 0.25 1"),
         ("6 GB host 4 KB, dropped by 1 thread first", "6 0 0 0 0 1"), ("6 GB host 4 KB, dropped by 16 threads first", "6 0 0 0 0 16"),
         ("6 GB host THP, dropped by 16 threads first", "6 1 0 0 0 16"), ("6 GB host 4 KB, dropped by 64 threads first", "6 0 0 0 0 64"),
         ("HIP context, 1 stream with a queue", "0 0 0 0 1 0 1"), ("HIP context, 2 streams", "0 0 0 0 1 0 2"), ("HIP context, 4 streams", "0 0 0 0 1 0 4"),
         ("HIP context, 8 streams (GPU_MAX_HW_QUEUES caps the queues)", "0 0 0 0 1 0 8")]
for name, args in cases:
    best = None
    for _ in range(2):
        t0 = time.time()
        p = subprocess.Popen([exe] + args.split(), stdout=subprocess.PIPE)
        out = p.stdout.read().decode().strip(); p.wait(); t1 = time.time()
        if p.returncode != 0: print(name, "rc", p.returncode); break
        d = t1 - float(out); best = d if best is None else min(best, d)
        run = float(out) - t0
    print("%-50s _exit -> reaped %.3f s   (spawn -> _exit %.3f s)" % (name, best or -1, run), flush=True)
