#!/usr/bin/env python3
"""Where do the merge kernels' microseconds go?  Runs pass 1 + the device merge of the 100 M-read workload with parts of the
merge kernels switched OFF (CRASS_DM_ABLATE bits, dmerge.hip — results are garbage, pass 2 is not run), three steps per setting,
under `rocprofv3 --kernel-trace`; `dm_ablate.py parse DIR` then prints every setting's k_dm_* durations (last step of the three).
  run:    rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- python3 tools/dm_ablate.py run [n_reads]
  parse:  python3 tools/dm_ablate.py parse DIR"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SETTINGS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 0]
NAMES = {0: "baseline", 1: "pack: no owner atomicMin", 2: "pack: no codes store", 4: "greedy: no needle-key claim", 8: "greedy: no spin",
         16: "redundant: no candidate compare", 32: "redundant: no probe", 64: "keys: no cnt atomicAdd", 128: "keys: no block atomics",
         256: "keys: no CAS", 512: "keys: nothing", 1024: "insert: no cuckoo insert", 2048: "insert: no block_reserve"}


def run(n):
    import crass_amd as ca
    L = 150
    spec = ca.synth_spec(read_len=L)
    eng = ca.SearchEngine(device=0)
    eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, L)
    for _ in range(3):
        eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    for a in SETTINGS:
        os.environ["CRASS_DM_ABLATE"] = str(a)
        for _ in range(3):
            eng.seed_scan(fetch=False)
            eng.merge(fetch=False)
        print("setting", a, eng.counters()["n_merge_fallbacks"], flush=True)
    os.environ.pop("CRASS_DM_ABLATE", None)
    eng.close()


def parse(d):
    ev = []
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].split("(")[0]
            if "k_dm_" in nm and "verify" not in nm:
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.replace("crass::", "").replace("void ", "")))
    ev.sort()
    merges, cur = [], None
    for s, e, nm in ev:
        if nm.startswith("k_dm_pack_codes") or (nm.startswith("k_dm_init") and cur is None):
            cur = []
            merges.append(cur)
        if cur is not None:
            cur.append((nm, (e - s) / 1e3, s, e))
    merges = merges[3:]                                   # the three warm-up steps
    cols = ["k_dm_pack_codes", "k_dm_greedy", "k_dm_rd_bases", "k_dm_rd_fill", "k_dm_redundant", "k_dm_keys", "k_dm_key_bases_insert", "k_dm_fill_finish"]
    print("%-34s" % "setting" + "".join("%9s" % c.replace("k_dm_", "")[:8] for c in cols) + "     span")
    for i, a in enumerate(SETTINGS):
        g = merges[3 * i + 2] if 3 * i + 2 < len(merges) else None
        if not g:
            continue
        dur = {nm: us for nm, us, _, _ in g}
        span = (max(x[3] for x in g) - min(x[2] for x in g)) / 1e3
        print("%-34s" % NAMES[a] + "".join("%9.1f" % dur.get(c, float("nan")) for c in cols) + "%9.1f" % span)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000)
    else:
        parse(sys.argv[2])
