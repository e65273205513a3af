#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into per-kernel HBM bytes per launch.

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request for wide
coalesced streaming reads, i.e. exactly half the bytes — doubled here.  Calibration in our own
access pattern: k_filter_fast_impl must read every packed word once = reads x 40 B; the doubled
counter gives that value to within 1.5 %.  WRITE_SIZE is used as reported.  Unit: KB."""
import csv
import json
import sys


def load(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "crass::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("crass::")[1].split("(")[0]
        out.setdefault(name, []).append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * f.get(k, 0.0)
        wr = w.get(k, 0.0)
        res[k] = {"hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr),
                  "fetch_size_raw_kb": round(f.get(k, 0.0) / 1024.0, 3), "write_size_raw_kb": round(w.get(k, 0.0) / 1024.0, 3)}
    json.dump({"workload": "bench.py --steps 1 --warmup 0 (10 M x 150 bp reads)", "correction": "FETCH_SIZE x2 (gfx950)",
               "per_launch": res}, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
