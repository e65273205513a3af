#!/usr/bin/env python3
"""bench.py — reads/sec through DR search + recruit on synthetic reads (BASELINE.json's metric).

One "step" = one pass of the whole hot path over one resident batch of synthetic reads:
pass 1 (seed scan + extend + QC, crass searchFile/searchCore) -> DR merge
(createNonRedundantSet, on the device; with N>1 ranks preceded by ONE RCCL all-gather of
every rank's distinct candidate DR strings) -> pass 2 (multi-pattern recruit,
findSingletons).  Packed reads are already resident in HBM when the timed region starts;
the timed region ends with the ordered candidate / recruit records, tokens, groups and
pattern list in host memory.

Workloads (BASELINE.json `configs`, selected with --config):
  --config 2  configs[2]: 100 M x 150 bp sharded over the ranks, STRONG scaling — the configuration the metric and
              its targets are quoted on, and the DEFAULT AT EVERY --gpus N (N = 1 is the base of the scaling curve)
  --config 1  configs[1]: 10 M x 150 bp, 50 seeded DRs, one GPU
  --config 3  configs[3]: 1 M x 10 kbp long reads, arrays of 20-60 repeats in 5 % of them
  --config 4  configs[4]: 200 M x 150 bp, 500 seeded DRs, 4 GC classes
  --total-reads T  strong scaling over T reads (rank r holds reads [r*T/N, (r+1)*T/N))
  --reads R        weak scaling, R reads per GPU (the round-1 mode)

Launching:
    python bench.py                                  # N=1
    python bench.py --gpus N                         # --launcher group (default): ONE process, N contexts, the engine itself
                                                     # issues the RCCL all-gather (crass_hip_group_*, ncclCommInitAll)
    python bench.py --gpus N --launcher torch        # spawns N rank processes (one per GPU), RCCL through torch.distributed
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N            # as the ranks of an external launcher: always the torch form
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    1: dict(name="BASELINE.json configs[1]", read_len=150, n_dr=50, gc_classes=0, total=10_000_000, arrays=(0, 0), cpm=10000,
            cpu_sample=4_000_000),
    2: dict(name="BASELINE.json configs[2]", read_len=150, n_dr=50, gc_classes=0, total=100_000_000, arrays=(0, 0), cpm=10000,
            cpu_sample=4_000_000),
    3: dict(name="BASELINE.json configs[3]", read_len=10000, n_dr=50, gc_classes=0, total=1_000_000, arrays=(20, 60), cpm=50000,
            cpu_sample=40_000),
    4: dict(name="BASELINE.json configs[4]", read_len=150, n_dr=500, gc_classes=4, total=200_000_000, arrays=(0, 0), cpm=10000,
            cpu_sample=4_000_000),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="BASELINE.json configs[i]; 0 = configs[2] (100 M reads, strong scaling) at every N")
    ap.add_argument("--total-reads", type=int, default=0, help="strong scaling: reads of the whole job, sharded over the ranks")
    ap.add_argument("--reads", type=int, default=0, help="weak scaling: reads per GPU")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--n-dr", type=int, default=0)
    ap.add_argument("--gc-classes", type=int, default=-1)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-strong-base", action="store_true",
                    help="N>1 strong scaling: skip the untimed one-GPU run of the whole job on rank 0 (speedup_vs_1gpu)")
    ap.add_argument("--alternate", action="store_true", help="alternate between two resident batches (speculation bounds "
                    "are then learnt from a different batch than the one being processed)")
    ap.add_argument("--single-shots", type=int, default=3, help="N=1: fresh contexts timed for single_shot_ms (0 = skip)")
    ap.add_argument("--check", action="store_true", help="also verify the GPU result against the oracle on the CPU sample")
    ap.add_argument("--e2e-reads", type=int, default=-1, help="N=1: reads of the end-to-end command-line run reported as e2e_reads_per_s "
                                                              "(FASTA in tmpfs -> crass-hip -> output files; default for 150 bp configs: 50 M where the host has the memory, else 5 M; 0 = skip)")
    ap.add_argument("--launcher", default="group", choices=["group", "torch"],
                    help="N>1 from a plain interpreter: group = one process drives N contexts through crass_hip_group_* (C++ RCCL); "
                         "torch = one process per GPU over torch.distributed.  Under an external launcher (WORLD_SIZE set) always torch")
    ap.add_argument("--local-copies", action="store_true", help="testing only (group launcher): every context on cuda:0, device copies "
                    "instead of the collective")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for single-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--params", default="", help="crass options other than the defaults, as on crass's command line: d=20,D=40 / w=7 / s=20,S=60 / n=3 / k=5 "
                                                  "(what a user of -d/-D/-s/-S/-w/-n gets: the fast filter and the position hints are built for the defaults)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` from a plain interpreter: start N rank processes (one per GPU) BEFORE anything in this
    process touches the GPU or imports torch, relay rank 0's JSON line, exit non-zero if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CRASS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            rc = rc or p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 1
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    sys.exit(1 if rc else 0)


def source_hash():
    h = hashlib.sha256()
    for f in ("kernels.hip", "dmerge.hip"):
        with open(os.path.join(ROOT, "crass_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    group_mode = args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.launcher == "group"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not group_mode:
        spawn_ranks(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "CRASS_HOST_THREADS" not in os.environ:
        # one process per GPU: the job's host view (tokens, groups, patterns) is built by rank 0 alone (set_host_view below), so
        # rank 0 takes the host-pool threads the node's CPU quota leaves and the other ranks take none
        w, r = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
        os.environ["CRASS_HOST_THREADS"] = "1" if r > 0 else str(max(2, min(16, int(_cpu_quota() * 0.5) - (w - 1))))
    import numpy as np
    import torch
    import crass_amd as ca
    ca.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != args.gpus and not group_mode:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # ---- options (default: crass's own) ----
    pkw = {}
    for kv in [x for x in args.params.split(",") if x]:
        k, v = kv.split("=")
        pkw[{"d": "lowDRsize", "D": "highDRsize", "s": "lowSpacerSize", "S": "highSpacerSize", "w": "searchWindowLength", "n": "minNumRepeats", "k": "kmer_clust_size"}[k]] = int(v)
    prm = ca.default_params(**pkw) if pkw else None
    _orc_params = None
    if pkw:
        from tests import orc as _orc0
        _orc_params = _orc0.Params(prm.lowDRsize, prm.highDRsize, prm.lowSpacerSize, prm.highSpacerSize, prm.searchWindowLength, prm.minNumRepeats, prm.kmer_clust_size)
    # ---- workload ----
    cfg_id = args.config or 2
    cfg = dict(CONFIGS[cfg_id])
    L = args.read_len or cfg["read_len"]
    n_dr = args.n_dr or cfg["n_dr"]
    gc = cfg["gc_classes"] if args.gc_classes < 0 else args.gc_classes
    if args.reads:                                          # weak scaling: every rank holds its own shard of one stream
        scaling, n, first, total = "weak", args.reads, rank * args.reads, world * args.reads
    else:                                                   # strong scaling: the job's reads are split over the ranks
        total = args.total_reads or cfg["total"]
        first, end = total * rank // world, total * (rank + 1) // world
        n = end - first                                     # (group launcher: world == 1, this process holds the whole job)
        # one job split over the ranks: configs[2] (and any --total-reads) is a strong-scaling workload at every N
        # incl. 1; the other configs are single-GPU workloads (per-GPU work fixed: "weak")
        scaling = "strong" if world > 1 or args.total_reads or cfg_id == 2 else "weak"
    custom = bool(args.reads or args.total_reads or args.read_len or args.n_dr or args.gc_classes >= 0 or args.params)
    W = (L + 15) // 16
    spec = ca.synth_spec(read_len=L, n_dr=n_dr, gc_classes=gc, crispr_per_million=cfg["cpm"],
                         array_min_repeats=cfg["arrays"][0], array_max_repeats=cfg["arrays"][1])
    # ---- end to end through the command line FIRST, while this process has not touched the GPU and has no engine, no pool threads
    #      and no pinned buffers of its own: what a user of `crass-hip` gets.  (Run behind the timed steps, beside this process's
    #      resident engine, the same child took 2.0 s instead of 1.6-1.7: profiles/NOTES_r06.md.) ----
    e2e_early = None
    if world == 1 and not group_mode and rank == 0 and torch.cuda.device_count() > 0:      # (counting devices does not initialise one)
        e2e_n = args.e2e_reads if args.e2e_reads >= 0 else (_e2e_default_reads(L) if L <= 300 and not custom else 0)
        if e2e_n > 0:
            e2e_early = _e2e_cli(ca, spec, L, min(e2e_n, total))
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

    t0 = time.time()
    words = ca.synth_packed(spec, first, n)
    t_gen = time.time() - t0

    if group_mode:
        if args.reads:
            print("bench.py: --launcher group shards ONE job (strong scaling); use --total-reads", file=sys.stderr)
            sys.exit(2)
        devs = [0] * n_gpus if args.local_copies else list(range(n_gpus))
        try:
            grp = ca.SearchGroup(devs, local_copies=args.local_copies)
        except ca.CrassError as ex:
            # The group needs RCCL inside THIS process (ncclCommInitAll, one thread per device, peer access between the
            # devices).  If it cannot be created the same job runs as one process per GPU over torch.distributed — in FRESH
            # child processes (this one has initialised the GPU: it is never replaced by an exec); rank 0's line says so.
            print("bench.py: group launcher unavailable (%s); running one process per GPU (torch.distributed) instead" % ex, file=sys.stderr)
            del words
            os.environ["CRASS_BENCH_FALLBACK"] = ("group launcher failed: %s" % ex)[:400]
            spawn_ranks(args)                              # (exits with the children's status)
        eng = _GroupRunner(grp, n_gpus)
        eng.g.load_packed_uniform(words, n, L, read_index_base=first)   # sharded by contiguous read ranges, H2D once
        scaling = "strong"
    else:
        eng = ca.SearchEngine(prm, local_rank)
        eng.load_packed_uniform(words, n, L, read_index_base=first)     # H2D once; resident for every step
    engs = [eng]
    if args.alternate:                                      # a second resident batch (the next reads of the same stream)
        eng_b = ca.SearchEngine(prm, local_rank)
        words_b = ca.synth_packed(spec, total + first, n)
        eng_b.load_packed_uniform(words_b, n, L, read_index_base=total + first)
        del words_b

    xg = None
    if world > 1 and args.dist_backend == "nccl":
        from crass_amd.distributed import GatheredExchange
        xg = GatheredExchange(eng, dist, coll_dev)          # one RCCL all-gather of fixed-size device buffers per step
        if rank > 0:
            # tokens, groups and the pattern list are identical on every rank: the job's host view is rank 0's; the other
            # ranks keep their own candidates' tokens only (as the ranks of a crass_hip_group do)
            eng.set_host_view(1)

    def step(e=eng):
        if isinstance(e, _GroupRunner):
            e.g.step()                                      # pass 1 on every shard -> ONE ncclAllGather -> merge -> pass 2, in C++
            return
        e.seed_scan(fetch=False)
        if xg is not None:
            while not xg.step():                            # capacity raised (first steps only): repeat the seed scan
                e.seed_scan(fetch=False)
        elif world > 1:
            # only the DISTINCT candidate DR strings travel (rank order == read order, so every rank
            # replays the same global token order and builds the same pattern set locally)
            # (gloo dry runs: the same exchange through host arrays)
            from crass_amd.distributed import allgather_distinct
            chars, lens, _ = e.distinct()
            g_chars, g_lens, my_off = allgather_distinct(chars, lens, dist, coll_dev)
            e.merge_distinct(g_chars, g_lens, my_off, fetch=False)
        else:
            e.merge(fetch=False)
        e.recruit(fetch=False)

    def sync():
        if dist is not None:
            dist.barrier()
        if group_mode and not args.local_copies:
            for d in range(n_gpus):
                torch.cuda.synchronize(d)
        else:
            torch.cuda.synchronize()

    # first call of a fresh context (no speculation bounds, buffers not yet allocated): reported, never part of `value`
    sync()
    t0 = time.perf_counter()
    step()
    sync()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(0, args.warmup - 1)):
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
    if args.alternate:
        eng_b.set_stage_timing(1)
        eng_b.set_timing_focus(names[dom][1])
        for _ in range(2):
            step(eng_b)
    for _ in range(2):                                   # (settle into the timed configuration)
        step()
    kern = {names[dom][0]: [], "ms_merge_host": [], "ms_sink_host": []}
    import gc
    gc.collect()
    gc.disable()                                         # (no collector pauses inside the timed region)
    step_marks = [0.0] * (args.steps + 1)                # a step ends with its records in host memory: wall-clock marks between steps
    sync()
    t0 = time.perf_counter()
    step_marks[0] = t0
    for it in range(args.steps):
        e = eng_b if (args.alternate and it & 1) else eng
        step(e)
        c = e.counters()
        for k in kern:
            kern[k].append(c[k])
        step_marks[it + 1] = time.perf_counter()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    per_step = np.diff(np.asarray(step_marks)) * 1e3
    eng.set_timing_focus(7)
    tot_p1, tot_p2 = None, None
    c = eng.counters()
    rccl_ranks = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([c["n_pass1_found"], c["n_pass2_found"], 1], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot_p1, tot_p2 = int(tot[0].item()), int(tot[1].item())
        if args.dist_backend == "nccl":
            rccl_ranks = int(tot[2].item())              # ranks that took part in an RCCL all-reduce
    if group_mode:
        tot_p1, tot_p2 = eng.totals()
        rccl_ranks = eng.g.rccl_ranks
        n = eng.reads_rank0                                  # the roofline below is rank 0's kernel on rank 0's shard
    ms_per_step = dt * 1e3 / args.steps
    value = total * args.steps / dt if not args.reads else world * n * args.steps / dt

    # ---- roofline of the dominant kernel (HIP events on the engine's stream, see engine.cpp) ----
    avg = {k: float(np.mean(v)) for k, v in kern.items()}
    bytes_per_read_per_pass = (L + 3) // 4               # ceil(L/4) B per read per pass
    dom_ms = avg[names[dom][0]]                          # HIP events over the timed region
    n_surv = c["n_filter_survivors"]
    if dom == "survivor":
        alg_bytes = n_surv * bytes_per_read_per_pass
    else:
        alg_bytes = n * bytes_per_read_per_pass
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # per-kernel view (the two streaming kernels touch every read; the survivor kernel only the reads that
    # survive the filter, so for short reads its algorithmic bytes are tiny and it is issue bound by nature)
    per_kernel = {}
    for name, units in (("seed_scan_filter", n), ("survivor", n_surv), ("recruit_scan", n)):
        ms = dom_ms if name == dom else scout[names[name][0]]
        b = units * bytes_per_read_per_pass
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        per_kernel[name] = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes": int(b), "achieved_GBps": round(gbs, 1),
                            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5),
                            "measured_in": "timed steps" if name == dom else "scouting steps (untimed)"}
    # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/ — only when
    # they were taken on THIS kernel source (hash recorded by profiles/summarize_pmc.py) and this workload
    traffic, traffic_src, valu_issue = None, None, None
    try:
        pm = _pmc_file(n, L)
        # (the seed-scan stage is k_filter_fast_impl for short uniform reads, k_hint_positions for long ones, else k_filter_general)
        keys = {"seed_scan_filter": ("k_filter_fast", "k_hint_positions", "k_filter_general"), "survivor": ("k_survivor",),
                "recruit_scan": ("k_anchor_filter",)}[dom]
        if pm and pm.get("source_hash") == source_hash() and pm.get("reads") == n and pm.get("read_len") == L:
            for k, v in sorted(pm["per_launch"].items(), key=lambda kv: -kv[1].get("hbm_bytes", 0)):
                if traffic is None and k.startswith(keys):
                    traffic, traffic_src = v["hbm_bytes"], "profiles/%s (committed PMC passes, same kernel source hash)" % pm["_file"]
                    if "sq" in v and v["sq"].get("SQ_INSTS_VALU"):
                        # what actually bounds this kernel: a wave64 VALU instruction occupies its SIMD for 4 cycles, the chip
                        # has 256 CUs x 4 SIMDs and holds at most 2.4 GHz (MI355X_MICROARCH.md)
                        insts = v["sq"]["SQ_INSTS_VALU"]
                        bound_ms = insts * 4.0 / (256 * 4) / 2.4e9 * 1e3
                        valu_issue = {"insts_valu_per_launch": int(insts), "cycles_per_wave64_inst": 4, "simds": 1024, "clock_ghz": 2.4,
                                      "bound_ms": round(bound_ms, 4), "frac": round(bound_ms / dom_ms, 4) if dom_ms > 0 else None,
                                      "note": "the kernel is VALU-issue bound, not HBM bound: frac = VALU issue time at peak clock / measured launch time"}
    except (OSError, KeyError, ValueError, TypeError):
        pass
    path_bytes = 2 * bytes_per_read_per_pass             # both passes read every base once at 2 bits
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "valu_issue": valu_issue,
                "path_frac": round(value / n_gpus * path_bytes / (HBM_PEAK_GBS * 1e9), 5),
                "path_frac_note": "whole path per GPU: reads/s/GPU x %d B / 8 TB/s (SURVEY 8d)" % path_bytes,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(dom_ms, 4),
                "per_kernel": per_kernel, "host_ms": {"merge": round(avg["ms_merge_host"], 3), "sink": round(avg["ms_sink_host"], 3)},
                "merge_device_ms": stages["ms_merge_device"], "device_merge": int(c.get("used_device_merge", 0)),
                "stages_ms_scouting_steps": stages}

    workload = "%d synthetic %d bp reads%s, %d seeded DRs%s (%s%s); pass1 + merge + pass2" % (
        total, L, "" if n_gpus == 1 else " sharded over %d GPUs" % n_gpus, n_dr,
        ", arrays of %d-%d repeats in %.0f %% of the reads" % (cfg["arrays"] + (cfg["cpm"] / 1e4,)) if cfg["arrays"][1]
        else ", %.0f %% CRISPR reads" % (cfg["cpm"] / 1e4),
        cfg["name"], ", modified by command-line options" if custom else "")
    out = {
        "metric": "reads/sec through DR search+recruit, 150bp synthetic, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": cfg_id, "total_reads": total, "reads_per_gpu": n, "read_len": L,
                   "n_dr": n_dr, "parallelism": "read-shards x%d" % n_gpus,
                   "launcher": ("group: one process, %d contexts, RCCL all-gather issued by the engine (crass_hip_group_*)" % n_gpus) if group_mode
                               else ("torch.distributed: one process per GPU" if world > 1 else "single context"),
                   "pass1_found": int(c["n_pass1_found"]) if tot_p1 is None else tot_p1,
                   "pass2_found": int(c["n_pass2_found"]) if tot_p2 is None else tot_p2,
                   "patterns": int(c["n_patterns"]), "ac_states": int(c["ac_states"]),
                   "filter_survivors": int(n_surv),
                   "fast_filter": int(c["used_fast_filter"]), "lds_automaton": int(c["used_lds_automaton"]),
                   "merge_fallbacks": int(c.get("n_merge_fallbacks", 0)),      # device merges redone on the host: must stay 0
                   "synth_gen_s": round(t_gen, 2), **({"params": args.params} if args.params else {})},
        "first_call_ms": round(first_call_ms, 3),
        # this rank's wall clock between step ends inside the timed region (value / ms_per_step are the contract's total / K)
        "step_ms": {"median": round(float(np.median(per_step)), 4), "min": round(float(per_step.min()), 4),
                    "p90": round(float(np.percentile(per_step, 90)), 4), "max": round(float(per_step.max()), 4)},
        "timed_batches": "two resident batches, alternating" if args.alternate else
                         "one resident batch repeated (speculation bounds learnt from the previous, identical step; "
                         "first_call_ms = the same step on a fresh context; --alternate switches batches)",
        "roofline": roofline,
    }
    if rccl_ranks is not None:
        out["rccl_ranks"] = rccl_ranks
    if os.environ.get("CRASS_BENCH_FALLBACK"):
        out["launcher_fallback"] = os.environ["CRASS_BENCH_FALLBACK"]

    # ---- the single-shot truth: a crass run scans each read set ONCE.  A FRESH context (no learnt bounds; its pools are
    #      sized from the read count when the reads are loaded), reads resident, ONE step, wall clock.  Outside the timed
    #      region. ----
    if world == 1 and not group_mode and args.single_shots > 0 and not args.alternate:
        shots = []
        for _ in range(args.single_shots):
            e1 = ca.SearchEngine(prm, local_rank)
            e1.set_stage_timing(0)
            e1.load_packed_uniform(words, n, L, read_index_base=first)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(e1)
            torch.cuda.synchronize()
            shots.append((time.perf_counter() - t1) * 1e3)
            c1 = e1.counters()
            assert (c1["n_pass1_found"], c1["n_pass2_found"]) == (c["n_pass1_found"], c["n_pass2_found"]), "single shot differs"
            e1.close()
        # ... and the same with the DEVICE kept busy in front of the fresh context's step (steps of the resident context): what is
        # left of the difference is the fresh context's; the rest is the device's clocks, which take ~30 ms of work to come up after
        # an idle period — a load (host packing, H2D copies) is one (tools/idle_effect.py: a steady-state step behind a 50 ms pause
        # costs 1.12 x a back-to-back one, the same as a fresh context's)
        busy = []
        for _ in range(args.single_shots):
            e1 = ca.SearchEngine(prm, local_rank)
            e1.set_stage_timing(0)
            e1.load_packed_uniform(words, n, L, read_index_base=first)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(e1)
            torch.cuda.synchronize()
            busy.append((time.perf_counter() - t1) * 1e3)
            e1.close()
        out["single_shot_ms"] = round(float(np.median(shots)), 3)
        out["single_shot_value"] = round(total / (out["single_shot_ms"] * 1e-3), 1)
        out["single_shot"] = {"ms_all": [round(x, 3) for x in shots], "contexts": len(shots),
                              "vs_steady_state": round(out["single_shot_ms"] / ms_per_step, 3),
                              "busy_device_ms_all": [round(x, 3) for x in busy], "busy_device_ms": round(float(np.median(busy)), 3),
                              "busy_device_vs_steady_state": round(float(np.median(busy)) / ms_per_step, 3),
                              "note": "one step on a fresh context (crass_hip_create + crass_hip_load_reads outside, as for "
                                      "`value`: reads resident in HBM); no learnt speculation bounds, nothing warmed.  busy_device: the "
                                      "same with ten steps of the resident context queued right in front of it — the device's clocks are "
                                      "up; the difference to single_shot_ms is the idle device's, not the fresh context's"}

    # ---- what the adapter pays on top of a step: the ABI's wide arrays (crass_candidates / crass_merge_view / crass_recruits)
    #      are widened from the compact blobs ON REQUEST, outside the timed region ----
    if world == 1 and not group_mode:
        fetch = []
        for _ in range(3):
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nf = eng.fetch_abi()
            fetch.append((time.perf_counter() - t1) * 1e3)
        out["abi_fetch_ms"] = round(float(np.median(fetch)), 3)
        out["abi_fetch"] = {"ms_all": [round(x, 3) for x in fetch], "candidates": nf[0], "tokens": nf[1], "recruits": nf[2],
                            "note": "crass_hip_get_candidates + crass_hip_get_merge + crass_hip_get_recruits after a step (median of 3): "
                                    "the per-record arrays of the ABI, widened from the step's compact blobs; not part of `value`"}
        if e2e_early is not None:                            # (measured at the start of this run, see above)
            out["e2e"] = e2e_early
            if "reads_per_s" in out["e2e"]:
                out["e2e_reads_per_s"] = out["e2e"]["reads_per_s"]

    # ---- strong scaling: the same job on ONE GPU (rank 0, untimed part of the run; the other ranks wait) ----
    if (world > 1 or group_mode) and scaling == "strong" and not args.no_strong_base and args.dist_backend == "nccl":
        base = None
        if rank == 0:
            try:
                w_all = words if group_mode else ca.synth_packed(spec, 0, total)
                e1 = ca.SearchEngine(prm, local_rank)
                e1.load_packed_uniform(w_all, total, L, read_index_base=0)
                del w_all
                for _ in range(3):
                    e1.seed_scan(fetch=False); e1.merge(fetch=False); e1.recruit(fetch=False)
                torch.cuda.synchronize()
                k1 = max(3, min(args.steps, 10))
                t0 = time.perf_counter()
                for _ in range(k1):
                    e1.seed_scan(fetch=False); e1.merge(fetch=False); e1.recruit(fetch=False)
                torch.cuda.synchronize()
                d1 = (time.perf_counter() - t0) / k1
                c1 = e1.counters()
                base = {"value": round(total / d1, 1), "ms_per_step": round(d1 * 1e3, 3), "steps": k1,
                        "pass1_found": int(c1["n_pass1_found"]), "pass2_found": int(c1["n_pass2_found"]),
                        "note": "the whole job on rank 0's GPU alone, same process, measured after the timed region"}
                e1.close()
            except Exception as ex:                     # reported extra; never fails the run
                base = {"error": str(ex)}
            out["one_gpu_same_job"] = base
            if base and "value" in base:
                out["speedup_vs_1gpu"] = round(value / base["value"], 3)
        if dist is not None:
            dist.barrier()

    # ---- CPU baseline: the oracle (single core, same algorithm class as the reference) on a
    #      bounded prefix of the SAME stream; rank 0 at N=1 only ----
    cpu_sample = cfg["cpu_sample"] if args.cpu_sample < 0 else args.cpu_sample
    if rank == 0 and world == 1 and not group_mode and cpu_sample > 0:
        from tests import orc
        m = min(cpu_sample, n)
        asc = ca.unpack_ascii(words, W, L, m)
        off = np.arange(0, (m + 1) * L, L, dtype=np.uint64)
        r = orc.pipeline_time_keep(asc, off, _orc_params)
        cpu_s = r["t_pass1"] + r["t_merge"] + r["t_pass2"]
        # ---- parity gate (a number only counts with it): the HIP path over the SAME prefix, on a fresh context,
        #      against the oracle run that was just timed — every record, token, group and pattern (tests/parity.py).  A mismatch
        #      is printed in the line and the run exits non-zero. ----
        try:
            from tests.parity import assert_same_pipeline
            e2 = ca.SearchEngine(prm, local_rank)
            e2.load_packed_uniform(words[:m * W], m, L)
            cand = e2.seed_scan(); mg = e2.merge(); rec = e2.recruit(); mg = e2.merge_view()
            gpu_res = ca.engine.PipelineResult(cand, mg, rec, cand.max_read_len)
            e2.close()
            try:
                assert_same_pipeline(gpu_res, r["result"])
                out["parity_checked"] = {"reads": int(m), "equal": True, "pass1_records": int(gpu_res.n_pass1), "pass2_records": int(gpu_res.n_pass2),
                                         "tokens": int(gpu_res.n_tokens), "groups": int(gpu_res.n_groups), "patterns": int(gpu_res.n_patterns),
                                         "what": "HIP path == oracle (the cpu_baseline run itself) on the sample's reads: every record field, token, group, pattern in order"}
            except AssertionError as ex:
                out["parity_checked"] = {"reads": int(m), "equal": False, "first_difference": str(ex)[:400]}
        except Exception as ex:                              # (the check could not run: said so, not passed off as equal)
            out["parity_checked"] = {"reads": int(m), "equal": None, "error": str(ex)[:400]}
        del r["result"]
        out["cpu_baseline"] = {"value": round(m / cpu_s, 1), "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads of the same synthetic stream; pass1 %.2fs merge %.2fs pass2 %.2fs"
                                         % (m, r["t_pass1"], r["t_merge"], r["t_pass2"]),
                               "cpu": _cpu_model()}
        # calibration of the port against the COMPILED reference's two hot functions (oracle/_ref travels to the GPU
        # box as a built .so): PatternMatcher::bmpSearch over searchCore's windows, acism_scan over the reads
        try:
            mc = min(m, 200_000 if L <= 1000 else 4_000)
            res = orc.pipeline((asc[:mc * L], off[:mc + 1]), params=_orc_params)
            cal = orc.calibrate(asc[:mc * L], mc, L, res.patterns)
            if cal is None:
                cal = json.load(open(os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")))
                cal["measured_on"] = cal.get("measured_on", "?") + " (committed file; oracle/_ref not present in this run)"
            else:
                cal["measured_on"] = _cpu_model() + " (this run)"
                cal["reference_equivalent_value"] = round(m / (r["t_pass1"] * cal["bmp_ratio"] + r["t_merge"] + r["t_pass2"] * cal["ac_ratio"]), 1)
                cal["note"] = ("ratio = compiled reference time / port time on the same inputs (>1: the port is faster); "
                               "reference_equivalent_value scales pass 1 by bmp_ratio and pass 2 by ac_ratio")
            out["cpu_baseline"]["calibration"] = cal
        except Exception as ex:
            out["cpu_baseline"]["calibration"] = {"error": str(ex)}
        # the same restatement on every host core (SURVEY §8d "node's host cores" figure): the sample is sharded
        # over T threads, each runs its own pass 1 + merge + pass 2 (ctypes releases the GIL during the C call)
        try:
            import threading
            T = max(1, min(os.cpu_count() or 1, 64))
            per = m // T
            if T > 1 and per >= (10000 if L <= 1000 else 200):
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
            _check(ca, eng, words, W, L, min(m, 200_000 if L <= 1000 else 2000), orc)
    eng.close()
    if args.alternate:
        eng_b.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and out.get("parity_checked", {}).get("equal") is False:
        print("bench.py: the HIP path's results differ from the oracle's on the sample: %s" % out["parity_checked"].get("first_difference"), file=sys.stderr)
        sys.exit(4)


class _GroupRunner:
    """bench.py's view of a crass_hip_group: stage timing and counters are rank 0's (the other ranks record no events)"""

    def __init__(self, grp, n_gpus):
        self.g, self.n_gpus = grp, n_gpus
        for r in range(1, n_gpus):
            grp.rank_set_stage_timing(r, 0)

    @property
    def reads_rank0(self):
        return int(self.g.rank_counters(0)["n_reads"])

    def set_stage_timing(self, level):
        self.g.rank_set_stage_timing(0, level)

    def set_timing_focus(self, k):
        self.g.rank_set_timing_focus(0, k)

    def counters(self):
        return self.g.rank_counters(0)

    def totals(self):
        cs = [self.g.rank_counters(r) for r in range(self.n_gpus)]
        return sum(int(c["n_pass1_found"]) for c in cs), sum(int(c["n_pass2_found"]) for c in cs)

    def close(self):
        self.g.close()


def _e2e_dir(need_bytes):
    """where the end-to-end input goes: tmpfs when it has the room (an 8 GB input should be read from memory, not from a disk of
    unknown speed), else the default temporary directory"""
    import shutil
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need_bytes + (4 << 30):
            return "/dev/shm"
    except OSError:
        pass
    return None


def _e2e_default_reads(L):
    """the metric's own configuration is 100 M reads; the end-to-end leg takes 50 M of them (an 8 GB FASTA) when the host has the
    memory (input in tmpfs + the reader's working set) and falls back to 5 M otherwise"""
    try:
        avail = 0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) << 10
        big = 50_000_000
        if avail > 96 << 30 and _e2e_dir(big * (11 + L)) == "/dev/shm":
            return big
    except (OSError, ValueError):
        pass
    return 5_000_000


def _e2e_cli(ca, spec, L, n, modes=("auto",)):
    """End to end through the command line: a FASTA file of n reads of the SAME synthetic stream (written in blocks of 5 M reads,
    into tmpfs when it has the room) -> `crass-hip -g -o DIR` (read + parse + pack + H2D + pass 1 + merge + pass 2 + hand-off +
    consensus + spacer graphs + .crispr / Group_*.fa written), wall clock of the whole process incl. HIP start-up, best of 2; peak
    RSS of the child.  modes: "auto" (the command line's own choice of reader: the indexed one for a plain-text input), or
    CRASS_INGEST values ("whole", "stream", "index") for the comparison tools/e2e_big.py prints."""
    import numpy as np
    import re
    import shutil
    import tempfile
    cli = os.path.join(ROOT, "crass_amd", "crass-hip")
    if not os.path.exists(cli):
        return {"error": "crass_amd/crass-hip not built"}
    td = tempfile.mkdtemp(prefix="crass_e2e_", dir=_e2e_dir(n * (11 + L)))
    try:
        fa = os.path.join(td, "e2e.fa")
        t_gen = time.perf_counter()
        with open(fa, "wb") as f:
            for first in range(0, n, 5_000_000):
                m = min(5_000_000, n - first)
                w = ca.synth_packed(spec, first, m)
                asc = ca.unpack_ascii(w, (L + 15) // 16, L, m).reshape(m, L)
                rec = np.empty((m, 10 + L + 1), np.uint8)
                ids = np.char.zfill(np.arange(first, first + m).astype("S8"), 8)
                rec[:, 0] = ord(">"); rec[:, 1:9] = np.frombuffer(ids.tobytes(), np.uint8).reshape(m, 8); rec[:, 9] = 10
                rec[:, 10:10 + L] = asc; rec[:, 10 + L] = 10
                f.write(rec.tobytes())
                del rec, asc, w
        t_gen = time.perf_counter() - t_gen
        res = {}
        for mode in modes:
            walls, found, rss, stages = [], None, 0.0, []
            # glibc.malloc.hugetlb=1: malloc's arenas in 2 MB pages (ReadHolders, graphs, every temporary string: 1.31-1.38 s instead of
            # 1.55-1.7 for 50 M reads).  `crass-hip` restarts itself once with this tunable when it safely can (nothing preloaded, no
            # device file open: crass_hip_cli.cpp); a GPU box preloads a guard library into every process, so the command line stays
            # as it was started there and the tunable is given to it here, as INTEGRATION.md tells a user to
            env = dict(os.environ, CRASS_TIMING="1")
            if "glibc.malloc.hugetlb" not in env.get("GLIBC_TUNABLES", ""):
                env["GLIBC_TUNABLES"] = (env["GLIBC_TUNABLES"] + ":" if env.get("GLIBC_TUNABLES") else "") + "glibc.malloc.hugetlb=1"
            env.pop("CRASS_INGEST", None)
            if mode != "auto":
                env["CRASS_INGEST"] = mode
            for _ in range(2):
                od = os.path.join(td, "out")
                shutil.rmtree(od, ignore_errors=True)
                os.makedirs(od)
                log = os.path.join(td, "stdout.txt")
                t0 = time.perf_counter()
                e0 = time.time()
                # (peak resident set: the command line's own VmHWM, printed with its stage times — the child's ru_maxrss would start
                # from this process's resident set at fork time)
                with open(log, "wb") as lf:
                    p = subprocess.run([cli, "-g", "-o", od, fa], stdout=lf, stderr=subprocess.STDOUT, env=env)
                walls.append(time.perf_counter() - t0)
                e1 = time.time()
                if p.returncode != 0:
                    return {"error": "crass-hip exited %d (%s)" % (p.returncode, mode)}
                stages = []
                for line in open(log, "rb").read().decode().replace("\r", "\n").splitlines():
                    if "Found" in line and "reads" in line:
                        found = line.strip()
                    if "cli: GLIBC_TUNABLES=" in line:
                        stages.append(line.strip()[15:])
                        continue
                    m3 = re.search(r"cli: (main entered|_exit called) at epoch ([0-9.]+)", line)
                    if m3:                              # what the process spends outside main (loader / address-space + KFD tear-down)
                        stages.append("cli: spawn -> main %.3f s" % (float(m3.group(2)) - e0) if m3.group(1) == "main entered"
                                      else "cli: _exit -> reaped by the parent %.3f s" % (e1 - float(m3.group(2))))
                        continue
                    m2 = re.search(r"peak RSS (\d+) MB", line)
                    if m2:
                        rss = max(rss, float(m2.group(1)))
                    if "[crass_timing]" in line and ("fastx" in line or "searchAndRecruit:" in line or "hand-off:" in line or "inflate:" in line or "cli:" in line or "outputs:" in line or
                                                     ("consensus: " in line and "consensus:   " not in line) or "(adapter)" in line):
                        stages.append(line.strip()[15:])
            res[mode] = {"wall_s": round(min(walls), 3), "reads_per_s": round(n / min(walls), 1), "walls_s": [round(x, 3) for x in walls],
                         "peak_rss_mb": round(rss, 1), "found": found, "stages": stages}
        first = res[modes[0]]
        out = {"reads": n, "wall_s": first["wall_s"], "reads_per_s": first["reads_per_s"], "walls_s": first["walls_s"],
               "peak_rss_mb": first["peak_rss_mb"], "fasta_mb": round(n * (11 + L) / 1e6, 1), "found": first["found"],
               "input_dir": os.path.dirname(td), "input_written_s": round(t_gen, 1), "reader": modes[0],
               "note": "`GLIBC_TUNABLES=glibc.malloc.hugetlb=1 crass-hip -g -o DIR file.fa`: process start + HIP init + read/parse/pack + H2D + pass 1 + merge + pass 2 + "
                       "hand-off + consensus + spacer graphs + output files; PCIe- and parse-inclusive, never part of `value`"}
        if len(modes) > 1:
            out["by_reader"] = res
        else:
            out["stages"] = first["stages"]
        return out
    except Exception as ex:                          # a reported extra: never fails the run
        return {"error": str(ex)[:300]}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def _pmc_file(n, L):
    """the committed PMC summary (profiles/r*_pmc_traffic*.json) taken on this workload, newest round first"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except (OSError, ValueError):
            continue
        if pm.get("reads") == n and pm.get("read_len") == L:
            pm["_file"] = os.path.basename(f)
            return pm
    return None


def _cpu_quota():
    """CPUs the cgroup quota gives this process (cpu.max), else the visible CPUs"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(per) > 0:
            return int(q) / int(per)
    except (OSError, ValueError):
        pass
    return float(os.cpu_count() or 1)


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
