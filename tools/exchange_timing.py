#!/usr/bin/env python3
"""Per-step cost of the N>1 exchange path, measured with a process group of one rank (RCCL)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import crass_amd as ca
from crass_amd.distributed import allgather_distinct_device, GatheredExchange
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
n, L = 10_000_000, 150
spec = ca.synth_spec(read_len=L)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, L)
xg = GatheredExchange(eng, dist, dev)
def step(exchange):
    t = [time.perf_counter()]
    eng.seed_scan(fetch=False); t.append(time.perf_counter())
    if exchange == 2:
        while not xg.step():
            eng.seed_scan(fetch=False)
        t.append(time.perf_counter())
    elif exchange:
        g = allgather_distinct_device(eng, dist, dev); t.append(time.perf_counter())
        eng.merge_distinct_device(g[0].data_ptr(), g[1].data_ptr(), g[0].shape[1], g[0].shape[0], g[2], fetch=False)
    else:
        t.append(time.perf_counter())
        eng.merge(fetch=False)
    t.append(time.perf_counter())
    eng.recruit(fetch=False); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
for ex in (0, 1, 2, 0, 2):
    for _ in range(3): step(ex)
    torch.cuda.synchronize()
    acc = np.zeros(4); K = 20
    t0 = time.perf_counter()
    for _ in range(K): acc += step(ex)
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - t0) / K
    print("exchange=%d  %.3f ms/step  seed_scan %.3f  allgather %.3f  merge %.3f  recruit %.3f" % ((ex, tot) + tuple(acc / K)), flush=True)
dist.destroy_process_group()
