#!/usr/bin/env python3
"""Dump the pass-1 candidate DR strings of a synthetic shard (one per line, read order) so the host
merge can be profiled offline:  python tools/dump_candidates.py OUT [--reads N] [--n-dr K] [--gc-classes G]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--n-dr", type=int, default=50)
    ap.add_argument("--gc-classes", type=int, default=0)
    a = ap.parse_args()
    import crass_amd as ca
    ca.load()
    spec = ca.synth_spec(read_len=a.read_len, n_dr=a.n_dr, gc_classes=a.gc_classes)
    words = ca.synth_packed(spec, 0, a.reads)
    with ca.SearchEngine() as eng:
        eng.load_packed_uniform(words, a.reads, a.read_len)
        c = eng.seed_scan()
        with open(a.out, "wb") as f:
            for k in range(c.n):
                f.write(c.dr(k) + b"\n")
    print("wrote %d candidates" % c.n)


if __name__ == "__main__":
    main()
