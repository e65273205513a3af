#!/usr/bin/env python3
"""What ONE rank does per step when BASELINE.json configs[2] (100 M reads) is sharded over W GPUs — measured on one GPU.
All W shards live on this GPU (W contexts); the other ranks' send buffers are computed once and stay put, and rank 0's
step is timed with the gathered buffer assembled by a device copy instead of the RCCL all-gather (whose latency over
xGMI is therefore NOT included: ~1 MB per rank, latency bound).  Prints the per-stage times of rank 0 and the speedup
over the same job on one GPU that these times would give.   python tools/scaling_projection.py [total_reads] [W ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crass_amd as ca
from crass_amd.distributed import _DevView
ca.load()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
worlds = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
L = 150
spec = ca.synth_spec(read_len=L)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)

def one_gpu(total=total):
    eng = ca.SearchEngine(device=0)
    eng.load_packed_uniform(ca.synth_packed(spec, 0, total), total, L)
    for _ in range(3):
        eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
    ms = 1e3 * (time.perf_counter() - t0) / K
    eng.close()
    return ms

def rccl_allgather_us(nbytes):
    """latency of ONE ncclAllGather call (torch.distributed, backend nccl = RCCL) on a one-rank communicator, nbytes per rank, in
    a loop on the current stream: the collective's own launch + kernel floor on this box.  What it cannot show is the xGMI hops
    of an 8-rank ring (7 steps of ~1 MB / 8 over links of ~50 GB/s each way: ~20 us of wire time + per-step latencies)."""
    import torch.distributed as dist
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        for _ in range(20): dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K): dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / K
        if not os.environ.get("KEEP_PG"):
            dist.destroy_process_group()                 # (its proxy threads must not sit beside the engines' helper threads below)
        return us
    except Exception as ex:                              # (reported, never fatal: the projection stands without it)
        print("rccl one-rank all-gather not measured: %s" % ex, flush=True)
        return None

ag_us = None if os.environ.get("NO_RCCL_MEASURE") else rccl_allgather_us(1 << 20)
print("one-rank ncclAllGather of 1 MB: %s us per call" % ("%.1f" % ag_us if ag_us else "n/a"), flush=True)
base_ms = one_gpu()
# the weak-scaling baselines (the shard as a job of its own) run HERE, before any rank step: tools/tl_sp.sh traces the LAST step of
# the process, which must be the last world's rank step (round 5 ran them behind it and the committed "N = 8" timeline was the
# 12.5 M-read baseline's)
small_ms_of = {W: one_gpu(total // W) for W in worlds}
out = {"total_reads": total, "one_gpu_ms_per_step": round(base_ms, 3), "rccl_allgather_1rank_1MB_us": ag_us and round(ag_us, 1), "projection": []}
print("one GPU, %d reads: %.3f ms/step" % (total, base_ms), flush=True)
for W in worlds:
    engs, sends = [], []
    cap = 16384
    while True:
        for e in engs: e.close()
        engs, sends = [], []
        for r in range(W):
            first, end = total * r // W, total * (r + 1) // W
            e = ca.SearchEngine(device=0)
            e.load_packed_uniform(ca.synth_packed(spec, first, end - first), end - first, L, read_index_base=first)
            ptr, nbytes = e.exchange_setup(W, r, cap)
            engs.append(e); sends.append(torch.as_tensor(_DevView(ptr, (nbytes,), "|u1"), device=dev))
        for e in engs: e.seed_scan(fetch=False)
        torch.cuda.synchronize()
        recv = torch.cat(sends).contiguous()
        need = engs[0].merge_gathered(recv.data_ptr(), fetch=False)
        if need is None: break
        while cap < 2 * need: cap *= 2
    nb = sends[0].numel()
    ev = torch.cuda.Event()
    sync_p1 = bool(os.environ.get("SP_SYNC_P1"))       # A/B: the round-5 order (the seed scan waits for pass 1, then the exchange is queued)
    ext = torch.cuda.ExternalStream(int(engs[0].stream_handle()), device=dev)
    torch.cuda.synchronize()
    engs[0].set_stage_timing(0)                        # (no event records between the kernels: the library's default since round 6)
    if not sync_p1: engs[0].exchange_set_deferred(True)
    def step():
        t = [time.perf_counter()]
        engs[0].seed_scan(fetch=False); t.append(time.perf_counter())
        if sync_p1:
            recv[:nb].copy_(sends[0])                      # stands in for the all-gather (rank 0's slot refreshed)
            ev.record(torch.cuda.current_stream(dev)); engs[0].stream_wait_event(ev.cuda_event)
        else:
            # the seed scan has returned with pass 1 still queued: the stand-in for the collective goes on the engine's stream, as
            # the group's ncclAllGather does (crass_hip_exchange_set_deferred)
            with torch.cuda.stream(ext):
                recv[:nb].copy_(sends[0])
        assert engs[0].merge_gathered(recv.data_ptr(), fetch=False) is None; t.append(time.perf_counter())
        engs[0].recruit(fetch=False); t.append(time.perf_counter())
        return [1e3 * (b - a) for a, b in zip(t, t[1:])]
    def cpu_stat():
        try:
            return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
        except Exception:
            return {}
    for _ in range(4): step()
    cs0 = cpu_stat()
    K = 30
    acc = np.zeros(3)
    t0 = time.perf_counter()
    for _ in range(K): acc += step()
    ms = 1e3 * (time.perf_counter() - t0) / K
    cs1 = cpu_stat()
    print("cgroup cpu.stat over the %d timed steps: %s" % (K, {k: cs1[k] - cs0.get(k, 0) for k in cs1 if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}), flush=True)
    c = engs[0].counters()
    m = engs[0].merge_view()
    # the device-copy stand-in costs ~5 us; the measured one-rank collective replaces it in the projected figure.  WEAK scaling,
    # labelled as such: W ranks of total / W reads each (the whole job) against ONE GPU with total / W reads (a W times smaller
    # job): efficiency = that small job's step / a rank's step — what the global DR set (exchange, de-duplication, the merge over
    # every rank's strings, a W times larger pattern set in pass 2) costs a rank on top of its own shard
    ms_rccl = ms + ((ag_us - 5.0) / 1e3 if ag_us else 0.0)
    small_ms = small_ms_of[W]                            # the same shard as a job of its own on one GPU (measured up front)
    row = {"world": W, "reads_per_rank": total // W, "rank0_ms_per_step": round(ms, 3), "rank0_ms_with_measured_allgather": round(ms_rccl, 3),
           "projected_speedup_with_measured_allgather": round(base_ms / ms_rccl, 2),
           "weak_scaling": {"reads_per_rank": total // W, "one_gpu_ms_on_that_many_reads": round(small_ms, 3), "efficiency": round(small_ms / ms_rccl, 3)},
           "seed_scan_ms": round(acc[0] / K, 3),
           "exchange_unpack_merge_ms": round(acc[1] / K, 3), "recruit_ms": round(acc[2] / K, 3), "tokens_global": m.n_tokens,
           "patterns": m.n_patterns, "cap_rows": cap, "projected_speedup": round(base_ms / ms, 2), "projected_efficiency": round(base_ms / ms / W, 3),
           "merge_fallbacks": c["n_merge_fallbacks"]}
    out["projection"].append(row)
    print(row, flush=True)
    for e in engs: e.close()
    del recv, sends
print(json.dumps(out))
