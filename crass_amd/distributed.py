"""Multi-GPU exchange step: one rank per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The path shards by contiguous read ranges; the ONLY data-path exchange is the all-gather of
the pass-1 candidates' representative DR strings between the seed pass and the merge
(SURVEY §8e).  Rank order == global read order, so every rank replays the same
addReadHolder token order and builds the identical pattern list locally — no broadcast of
tables.  Payload: <= a few 10^4 candidates x (stride + 2) bytes, i.e. latency-bound.
"""
import numpy as np


def allgather_candidates(chars, lens, dist, device=None):
    """chars: uint8 [n_local, stride], lens: uint16 [n_local]  ->  (uint8 [n_global, stride],
    uint16 [n_global]) concatenated in rank order.  Two collectives: counts, then fixed-size
    padded slots (all_gather needs equal shapes)."""
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    stride = int(chars.shape[1]) if chars.ndim == 2 and chars.shape[0] else None
    meta = torch.tensor([int(chars.shape[0]), stride or 0], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0].item()) for m in metas]
    strides = {int(m[1].item()) for m in metas if int(m[0].item()) > 0}
    if not strides:
        return np.zeros((0, 16), np.uint8), np.zeros(0, np.uint16)
    assert len(strides) == 1, "ranks disagree on the DR slot stride"
    stride = strides.pop()
    cap = max(counts)
    slot = stride + 2                                   # DR bytes + uint16 length
    buf = np.zeros((cap, slot), dtype=np.uint8)
    n = int(chars.shape[0])
    if n:
        buf[:n, :stride] = chars
        buf[:n, stride:] = lens.astype("<u2").view(np.uint8).reshape(n, 2)
    send = torch.from_numpy(buf).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    parts_c, parts_l = [], []
    for r in range(world):
        a = recv[r][:counts[r]].cpu().numpy()
        parts_c.append(a[:, :stride])
        parts_l.append(np.ascontiguousarray(a[:, stride:]).view("<u2").reshape(-1).astype(np.uint16))
    return np.ascontiguousarray(np.concatenate(parts_c, axis=0)), np.concatenate(parts_l)


def allgather_distinct(chars, lens, dist, device=None):
    """Compact exchange: every rank contributes only its DISTINCT candidate strings (first-occurrence
    order).  Returns (global chars, global lens, offset of this rank's list in the concatenation)."""
    g_chars, g_lens = allgather_candidates(chars, lens, dist, device)
    counts = gather_counts([int(chars.shape[0])], dist, device)
    my_off = sum(c[0] for c in counts[:dist.get_rank()])
    return g_chars, g_lens, my_off


def gather_counts(values, dist, device=None):
    """small helper: all-gather a list of python ints -> [world][len(values)]"""
    import torch
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor(list(values), dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


class _DevView:
    """a device buffer of the engine as a __cuda_array_interface__ object (torch.as_tensor makes a view of it)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def allgather_distinct_device(eng, dist, device):
    """The exchange with everything left on the device (backend "nccl" = RCCL over xGMI): every rank contributes
    its distinct candidate strings straight from the engine's device buffers; the rank-ordered concatenation
    comes back as two device tensors (uint8 [n_global, stride] and uint8 [n_global, 2] = the uint16 lengths).
    Returns (g_chars, g_lens, my_offset) or None when this rank's list is not on the device (the caller then
    uses allgather_distinct with host arrays — every rank must take the same branch, see agree_all)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dd = eng.distinct_device()
    ok = torch.tensor([1 if dd is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None
    p_chars, p_len, n, stride = dd
    meta = torch.tensor([n, stride], dtype=torch.int64, device=device)
    metas = torch.empty((world, 2), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(metas, meta)
    metas = metas.cpu()
    counts = [int(metas[r, 0]) for r in range(world)]
    strides = {int(metas[r, 1]) for r in range(world)}
    assert len(strides) == 1, "ranks disagree on the DR slot stride"
    cap = max(max(counts), 1)
    slot = stride + 2                                   # DR bytes + the uint16 length as two bytes: ONE collective
    send = torch.zeros((cap, slot), dtype=torch.uint8, device=device)
    if n:
        send[:n, :stride] = torch.as_tensor(_DevView(p_chars, (n, stride), "|u1"), device=device)
        send[:n, stride:] = torch.as_tensor(_DevView(p_len, (n, 2), "|u1"), device=device)
    recv = torch.empty((world, cap, slot), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    cat = torch.cat([recv[r, :counts[r]] for r in range(world)])
    g_chars = cat[:, :stride].contiguous()
    g_lens = cat[:, stride:].contiguous()               # [n_global, 2] bytes == uint16 little endian
    torch.cuda.current_stream(device).synchronize()         # the engine reads them on its own stream
    return g_chars, g_lens, sum(counts[:rank])


class GatheredExchange:
    """The exchange as ONE collective per step (RCCL all-gather of fixed-size device buffers).  The engine's seed
    scan leaves the rank's distinct list in its send buffer; step() all-gathers it and merges on the device.
    The row capacity adapts: when some rank's list does not fit (every rank learns that from the gathered
    headers) the buffers are enlarged and the caller repeats the seed scan."""

    def __init__(self, eng, dist, device, cap_rows=0, deferred=True):
        self.eng, self.dist, self.device = eng, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self._ev = None
        # deferred: eng.seed_scan() returns with pass 1 still queued, the collective is issued with the ENGINE's stream as torch's
        # current stream (the process group orders itself against the current stream on both sides), merge_gathered queues its
        # kernels behind it — no host wait and no idle device between pass 1 and the exchange (crass_hip_exchange_set_deferred)
        self.deferred = bool(deferred)
        self._ext = None
        if not cap_rows:
            # the same capacity on every rank, from the LARGEST shard (crass_hip_exchange_rows_for: the engine's own first-call
            # bound for a shard's distinct DR strings), so that the first step does not overflow and repeat pass 1
            import torch
            n = torch.tensor([int(eng.counters()["n_reads"])], dtype=torch.int64, device=device)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            cap_rows = int(eng.lib.crass_hip_exchange_rows_for(int(n.item())))
        self._setup(cap_rows)
        # bring the communicator up now (RCCL initialises lazily on the first collective): not inside a timed step
        import torch
        self.dist.all_gather_into_tensor(self.recv, self.send)
        torch.cuda.current_stream(self.device).synchronize()

    def _setup(self, cap_rows):
        import torch
        self.cap_rows = int(cap_rows)
        ptr, nbytes = self.eng.exchange_setup(self.world, self.rank, self.cap_rows)
        self.send = torch.as_tensor(_DevView(ptr, (nbytes,), "|u1"), device=self.device)
        self.recv = torch.empty((self.world * nbytes,), dtype=torch.uint8, device=self.device)
        if self.deferred:
            if self._ext is None:
                self._ext = torch.cuda.ExternalStream(int(self.eng.stream_handle()), device=self.device)
            torch.cuda.current_stream(self.device).synchronize()      # (recv's allocation belongs to the stream that was current)
            self.eng.exchange_set_deferred(True)

    def step(self):
        """after eng.seed_scan(): True when merged, False when the capacity was raised (repeat the seed scan)"""
        import torch
        if self.deferred:
            with torch.cuda.stream(self._ext):
                self.dist.all_gather_into_tensor(self.recv, self.send)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)
            # the engine reads recv on its own stream: ordered behind the collective by an event, no host wait
            if self._ev is None:
                self._ev = torch.cuda.Event()
            self._ev.record(torch.cuda.current_stream(self.device))
            self.eng.stream_wait_event(self._ev.cuda_event)
        need = self.eng.merge_gathered(self.recv.data_ptr(), fetch=False)
        if need is None:
            return True
        cap = self.cap_rows
        while cap < need * 2:
            cap *= 2
        self._setup(cap)
        return False
