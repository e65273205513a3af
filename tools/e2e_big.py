#!/usr/bin/env python3
"""End to end on (half of) the metric's own configuration: `crass-hip -g -o DIR` on a FASTA of N reads (default 50 M = 8 GB, in
tmpfs) of bench.py's synthetic stream with each of the command line's readers — auto (the indexed reader for a plain-text input),
whole-file, streamed — wall clock of the whole process, its stage lines and peak RSS.  bench.py's own e2e leg is the "auto" row.
   python tools/e2e_big.py [reads] [reader ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crass_amd as ca
import bench
ca.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
modes = tuple(sys.argv[2:]) or ("auto", "whole", "stream")
L = 150
spec = ca.synth_spec(read_len=L)
r = bench._e2e_cli(ca, spec, L, n, modes=modes)
if "error" in r:
    print(r); sys.exit(1)
print("%d reads, FASTA %.0f MB in %s (written in %.1f s)" % (r["reads"], r["fasta_mb"], r["input_dir"], r["input_written_s"]))
by = r.get("by_reader") or {modes[0]: r}
for m in modes:
    x = by[m]
    print("== reader %-6s  wall %.3f s (%s)  %.2f M reads/s  peak RSS %.0f MB  %s" % (m, x["wall_s"], ", ".join("%.3f" % w for w in x["walls_s"]), x["reads_per_s"] / 1e6,
                                                                                     x["peak_rss_mb"], x["found"]))
    for line in x["stages"]:
        print("      " + line)
print(json.dumps({k: v for k, v in r.items() if k != "by_reader"}))
