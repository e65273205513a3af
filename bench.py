#!/usr/bin/env python3
"""bench.py — reads/sec through DR search + recruit on synthetic 150 bp reads.

One "step" = one pass of the whole hot path over one resident batch of synthetic reads:
pass 1 (seed scan + extend + QC, crass searchFile/searchCore) -> DR merge
(createNonRedundantSet, on the device; with N>1 ranks preceded by ONE RCCL all-gather of
every rank's distinct candidate DR strings) -> pass 2 (multi-pattern recruit,
findSingletons).  Packed reads are already resident in HBM when the timed region starts;
the timed region ends with the ordered candidate / recruit records, tokens, groups and
pattern list in host memory.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 10 M synthetic 150 bp reads per GPU, 50 seeded DRs,
1 % CRISPR reads (weak scaling: every rank holds its own 10 M-read shard of one global
stream).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--n-dr", type=int, default=50)
    ap.add_argument("--gc-classes", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads timed on the CPU baseline (0 = skip)")
    ap.add_argument("--check", action="store_true", help="also verify the GPU result against the oracle on the CPU sample")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for single-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import crass_amd as ca
    ca.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
    coll_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")

    L, n = args.read_len, args.reads
    W = (L + 15) // 16
    spec = ca.synth_spec(read_len=L, n_dr=args.n_dr, gc_classes=args.gc_classes)
    first = rank * n                                     # weak scaling: shard `rank` of one global stream
    t0 = time.time()
    words = ca.synth_packed(spec, first, n)
    t_gen = time.time() - t0

    eng = ca.SearchEngine(device=local_rank)
    eng.load_packed_uniform(words, n, L, read_index_base=first)     # H2D once; resident for every step

    xg = None
    if world > 1 and args.dist_backend == "nccl":
        from crass_amd.distributed import GatheredExchange
        xg = GatheredExchange(eng, dist, coll_dev)          # one RCCL all-gather of fixed-size device buffers per step

    def step():
        eng.seed_scan(fetch=False)
        if xg is not None:
            while not xg.step():                            # capacity raised (first steps only): repeat the seed scan
                eng.seed_scan(fetch=False)
        elif world > 1:
            # only the DISTINCT candidate DR strings travel (rank order == read order, so every rank
            # replays the same global token order and builds the same pattern set locally)
            # (gloo dry runs: the same exchange through host arrays)
            from crass_amd.distributed import allgather_distinct
            chars, lens, _ = eng.distinct()
            g_chars, g_lens, my_off = allgather_distinct(chars, lens, dist, coll_dev)
            eng.merge_distinct(g_chars, g_lens, my_off, fetch=False)
        else:
            eng.merge(fetch=False)
        eng.recruit(fetch=False)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # untimed scouting steps with every stage timed (level 2): the stage breakdown, and which of the three large
    # kernels is the dominant one
    eng.set_stage_timing(2)
    scout_keys = ("ms_filter", "ms_survivor", "ms_recruit", "ms_compact", "ms_pass1_total", "ms_merge_device", "ms_recruit_finish",
                  "ms_pass2_total")
    scout = {k: [] for k in scout_keys}
    for _ in range(3):
        step()
        cs = eng.counters()
        for k in scout:
            scout[k].append(cs[k])
    scout = {k: float(np.mean(v)) for k, v in scout.items()}
    stages = {k: round(scout[k], 4) for k in ("ms_compact", "ms_pass1_total", "ms_merge_device", "ms_recruit_finish", "ms_pass2_total")}
    names = {"seed_scan_filter": ("ms_filter", 1), "survivor": ("ms_survivor", 2), "recruit_scan": ("ms_recruit", 4)}
    dom = max(names, key=lambda k: scout[names[k][0]])
    # timed region: HIP events around the dominant kernel only (level 1 with a focus: every event record costs ~6 us
    # of stream time, and the other two kernels' durations are reported from the scouting steps)
    eng.set_stage_timing(1)
    eng.set_timing_focus(names[dom][1])
    for _ in range(2):                                   # (settle into the timed configuration)
        step()
    kern = {names[dom][0]: [], "ms_merge_host": [], "ms_sink_host": []}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        c = eng.counters()
        for k in kern:
            kern[k].append(c[k])
    sync()
    dt = time.perf_counter() - t0
    eng.set_timing_focus(7)
    tot_p1, tot_p2 = None, None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cc = eng.counters()
        tot = torch.tensor([cc["n_pass1_found"], cc["n_pass2_found"]], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot_p1, tot_p2 = int(tot[0].item()), int(tot[1].item())
    c = eng.counters()
    ms_per_step = dt * 1e3 / args.steps
    value = world * n * args.steps / dt

    # ---- roofline of the dominant kernel (HIP events on the engine's stream, see engine.cpp) ----
    avg = {k: float(np.mean(v)) for k, v in kern.items()}
    bytes_per_read_per_pass = (L + 3) // 4               # ceil(L/4) B per read per pass
    dom_ms = avg[names[dom][0]]                          # HIP events over the timed region
    if dom == "survivor":
        alg_bytes = c["n_filter_survivors"] * bytes_per_read_per_pass
    else:
        alg_bytes = n * bytes_per_read_per_pass
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # per-kernel view (the two streaming kernels touch every read; the survivor kernel only the ~2 % that
    # survive the filter, so its algorithmic bytes are tiny and it is latency/issue bound by nature)
    per_kernel = {}
    for name, units in (("seed_scan_filter", n), ("survivor", c["n_filter_survivors"]), ("recruit_scan", n)):
        ms = dom_ms if name == dom else scout[names[name][0]]
        b = units * bytes_per_read_per_pass
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        per_kernel[name] = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes": int(b), "achieved_GBps": round(gbs, 1),
                            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5),
                            "measured_in": "timed steps" if name == dom else "scouting steps (untimed)"}
    traffic = None
    try:        # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/summarize_pmc.py)
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["per_launch"]
        key = {"seed_scan_filter": "k_filter_fast_impl", "survivor": "k_survivor", "recruit_scan": "k_anchor_filter"}[dom]
        for k, v in pm.items():
            if k.startswith(key) and n == 10_000_000 and L == 150:
                traffic = v["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(dom_ms, 4),
                "per_kernel": per_kernel, "host_ms": {"merge": round(avg["ms_merge_host"], 3), "sink": round(avg["ms_sink_host"], 3)},
                "merge_device_ms": stages["ms_merge_device"], "device_merge": int(c.get("used_device_merge", 0)),
                "stages_ms_untimed_steps": stages}

    out = {
        "metric": "reads/sec through DR search+recruit, 150bp synthetic, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d synthetic %d bp reads per GPU, %d seeded DRs, 1%% CRISPR reads "
                               "(BASELINE.json configs[1]); pass1 + merge + pass2" % (n, L, args.n_dr),
                   "reads_per_gpu": n, "read_len": L, "n_dr": args.n_dr, "parallelism": "read-shards x%d" % world,
                   "pass1_found": int(c["n_pass1_found"]) if tot_p1 is None else tot_p1,
                   "pass2_found": int(c["n_pass2_found"]) if tot_p2 is None else tot_p2,
                   "patterns": int(c["n_patterns"]), "ac_states": int(c["ac_states"]),
                   "filter_survivors": int(c["n_filter_survivors"]),
                   "fast_filter": int(c["used_fast_filter"]), "lds_automaton": int(c["used_lds_automaton"]),
                   "synth_gen_s": round(t_gen, 2)},
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle (single core, same algorithm class as the reference) on a
    #      bounded prefix of the SAME stream; rank 0 at N=1 only ----
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from tests import orc
        m = min(args.cpu_sample, n)
        asc = ca.unpack_ascii(words, W, L, m)
        off = np.arange(0, (m + 1) * L, L, dtype=np.uint64)
        r = orc.pipeline_time(asc, off)
        cpu_s = r["t_pass1"] + r["t_merge"] + r["t_pass2"]
        out["cpu_baseline"] = {"value": round(m / cpu_s, 1), "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads of the same synthetic stream; pass1 %.2fs merge %.2fs pass2 %.2fs"
                                         % (m, r["t_pass1"], r["t_merge"], r["t_pass2"]),
                               "cpu": _cpu_model()}
        # the same restatement on every host core (SURVEY §8d "node's host cores" figure): the sample is sharded
        # over T threads, each runs its own pass 1 + merge + pass 2 (ctypes releases the GIL during the C call)
        try:
            import threading
            T = max(1, min(os.cpu_count() or 1, 64))
            per = m // T
            if T > 1 and per >= 10000:
                res = [None] * T

                def work(t):
                    lo = t * per
                    sub_off = (off[lo:lo + per + 1] - off[lo]).astype(np.uint64)
                    res[t] = orc.pipeline_time(asc[lo * L:(lo + per) * L], sub_off)
                th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
                t0 = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                dt_all = time.perf_counter() - t0
                out["cpu_baseline"]["all_cores"] = {"value": round(T * per / dt_all, 1), "unit": "reads/s", "cores": T,
                                                    "sample": "%d threads x %d reads (independent shards), wall %.2fs" % (T, per, dt_all)}
        except Exception as e:                      # the reported baseline above does not depend on this extra
            out["cpu_baseline"]["all_cores"] = {"error": str(e)}
        if args.check:
            _check(ca, eng, words, W, L, m, orc)
    eng.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " x%d" % os.cpu_count()
    except OSError:
        pass
    return "unknown"


def _check(ca, eng, words, W, L, m, orc):
    from tests.parity import assert_same_pipeline
    asc = ca.unpack_ascii(words, W, L, m)
    seqs = [asc[i * L:(i + 1) * L].tobytes() for i in range(m)]
    gpu = ca.search_pipeline(seqs)
    assert_same_pipeline(gpu, orc.pipeline(seqs))


if __name__ == "__main__":
    main()
