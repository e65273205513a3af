"""Multi-rank path end to end on ONE GPU: two processes (torch.distributed, gloo for the dry run;
the same code path uses RCCL with one rank per GPU) each hold a contiguous shard of the reads,
all-gather the candidate DR strings after pass 1, merge locally and recruit their shard.  The
concatenated result must equal the oracle's single-process result on the whole read set."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
import crass_amd as ca
from crass_amd.distributed import allgather_candidates
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
n_total, L = %(n)d, 150
spec = ca.synth_spec(read_len=L, crispr_per_million=30000)
lo = n_total * rank // world; hi = n_total * (rank + 1) // world
words = ca.synth_packed(spec, lo, hi - lo)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(words, hi - lo, L, read_index_base=lo)
cand = eng.seed_scan()
if os.environ["EXCHANGE"] == "candidates":
    # one DR string per candidate travels
    chars, lens = eng.candidate_dr_view()
    g_chars, g_lens = allgather_candidates(chars, lens, dist)
    m = eng.merge(g_chars, g_lens)
    counts = [None] * world
    dist.all_gather_object(counts, int(cand.n))
    off = sum(counts[:rank])                 # this rank's slice of the global candidate token list
else:
    # compact form: only the distinct strings travel (what bench.py uses)
    from crass_amd.distributed import allgather_distinct
    chars, lens, cmap = eng.distinct()
    assert len(cmap) == cand.n and [cand.dr(k) for k in range(0, cand.n, 37)] == \
        [chars[cmap[k], :lens[cmap[k]]].tobytes() for k in range(0, cand.n, 37)]
    g_chars, g_lens, my_off = allgather_distinct(chars, lens, dist)
    m = eng.merge_distinct(g_chars, g_lens, my_off)
    off = 0                                  # cand_token covers this rank's candidates only
rec = eng.recruit()
m = eng.merge_view()
out = dict(read=cand.read_idx.tolist(), low=cand.low_lexi.tolist(), replen=cand.repeat_len.tolist(),
           ss=[cand.ss(k) for k in range(cand.n)], tok=m.cand_token[off:off + cand.n].tolist(),
           r_read=rec.read_idx.tolist(), r_low=rec.low_lexi.tolist(), r_ss=[[int(a), int(b)] for a, b in zip(rec.start, rec.end)],
           r_tok=rec.token.tolist(), tokens=[t.decode() for t in m.tokens], groups=m.groups, patterns=sorted(p.decode() for p in m.patterns))
json.dump(out, open(%(out)r %% rank, "w"))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("exchange", ["candidates", "distinct"])
def test_two_ranks_equal_single_process_oracle(tmp_path, exchange):
    import crass_amd as ca
    n = 120000
    outpat = str(tmp_path / "rank%d.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, n=n, out=outpat))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EXCHANGE=exchange)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    r = [json.load(open(outpat % k)) for k in range(2)]
    spec = ca.synth_spec(read_len=150, crispr_per_million=30000)
    words = ca.synth_packed(spec, 0, n)
    asc = ca.unpack_ascii(words, 10, 150, n)
    off = np.arange(0, (n + 1) * 150, 150, dtype=np.uint64)
    ref = orc.pipeline((asc, off))
    # identical tables on both ranks, equal to the oracle's
    assert r[0]["tokens"] == r[1]["tokens"] == [t.decode() for t in ref.tokens]
    assert r[0]["groups"] == r[1]["groups"] == ref.groups
    assert r[0]["patterns"] == r[1]["patterns"] == sorted(p.decode() for p in ref.patterns)
    # pass-1 records: rank order == read order
    n1 = ref.n_pass1
    assert r[0]["read"] + r[1]["read"] == ref.rec_read[:n1].tolist()
    assert r[0]["low"] + r[1]["low"] == ref.rec_lowlexi[:n1].tolist()
    assert r[0]["replen"] + r[1]["replen"] == ref.rec_replen[:n1].tolist()
    assert r[0]["ss"] + r[1]["ss"] == [ref.ss(k) for k in range(n1)]
    assert r[0]["tok"] + r[1]["tok"] == ref.rec_token[:n1].tolist()
    # pass-2 records
    n2 = ref.n_pass2
    assert r[0]["r_read"] + r[1]["r_read"] == ref.rec_read[n1:n1 + n2].tolist()
    assert r[0]["r_low"] + r[1]["r_low"] == ref.rec_lowlexi[n1:n1 + n2].tolist()
    assert r[0]["r_ss"] + r[1]["r_ss"] == [ref.ss(k) for k in range(n1, n1 + n2)]
    assert r[0]["r_tok"] + r[1]["r_tok"] == ref.rec_token[n1:n1 + n2].tolist()
    assert n1 > 1000 and n2 > 1000


def test_bench_two_rank_dry_run(tmp_path):
    """bench.py's N>1 code path (barrier, max-over-ranks timing, all-gather) with gloo on one GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "500000", "--dist-backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["pass1_found"] > 0 and d["config"]["pass2_found"] > 0
    # same totals as one process over both shards
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--reads", "1000000",
                          "--cpu-sample", "0"], capture_output=True, timeout=600)
    d1 = json.loads([l for l in one.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert (d["config"]["pass1_found"], d["config"]["pass2_found"], d["config"]["patterns"]) == \
        (d1["config"]["pass1_found"], d1["config"]["pass2_found"], d1["config"]["patterns"])


DEVICE_EXCHANGE = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
torch.cuda.init()                      # torch's HIP runtime first (it does not come up once another one holds the GPU)
from tests import orc
import crass_amd as ca
ca.load()
n, L = 120000, 150
spec = ca.synth_spec(read_len=L, crispr_per_million=30000)
engs, lists = [], []
for r in range(2):
    lo, hi = n * r // 2, n * (r + 1) // 2
    e = ca.SearchEngine(device=0)
    e.load_packed_uniform(ca.synth_packed(spec, lo, hi - lo), hi - lo, L, read_index_base=lo)
    e.seed_scan()
    engs.append(e)
    lists.append(e.distinct())
g_chars = np.concatenate([l[0] for l in lists])
g_lens = np.concatenate([l[1] for l in lists])
t_chars = torch.from_numpy(g_chars).cuda()
t_lens = torch.from_numpy(g_lens.view(np.int16)).cuda()
torch.cuda.synchronize()
offs = [0, len(lists[0][1])]
views, recs = [], []
for r in range(2):
    assert engs[r].distinct_device() is not None
    m = engs[r].merge_distinct_device(t_chars.data_ptr(), t_lens.data_ptr(), g_chars.shape[1], g_chars.shape[0], offs[r])
    assert engs[r].counters()["used_device_merge"] == 1
    recs.append(engs[r].recruit())
    views.append(engs[r].merge_view())
# host-side exchange on fresh state for comparison
os.environ["CRASS_HOST_MERGE"] = "1"
try:
    hv = []
    for r in range(2):
        engs[r].reload_env()               # (the switches are read once per context)
        engs[r].seed_scan()
        engs[r].merge_distinct(g_chars, g_lens, offs[r])
        assert engs[r].counters()["used_device_merge"] == 0
        hr = engs[r].recruit()
        hv.append((engs[r].merge_view(), hr))
finally:
    os.environ.pop("CRASS_HOST_MERGE", None)
    for e in engs:
        e.reload_env()
for r in range(2):
    assert views[r].tokens == hv[r][0].tokens and views[r].groups == hv[r][0].groups and views[r].patterns == hv[r][0].patterns
    assert views[r].cand_token.tolist() == hv[r][0].cand_token.tolist()
    assert recs[r].read_idx.tolist() == hv[r][1].read_idx.tolist() and recs[r].token.tolist() == hv[r][1].token.tolist()
    assert recs[r].start.tolist() == hv[r][1].start.tolist() and recs[r].end.tolist() == hv[r][1].end.tolist()
asc = ca.unpack_ascii(ca.synth_packed(spec, 0, n), 10, L, n)
ref = orc.pipeline((asc, np.arange(0, (n + 1) * L, L, dtype=np.uint64)))
assert views[0].tokens == ref.tokens and views[0].groups == ref.groups
n1, n2 = ref.n_pass1, ref.n_pass2
assert views[0].cand_token.tolist() + views[1].cand_token.tolist() == ref.rec_token[:n1].tolist()
assert recs[0].read_idx.tolist() + recs[1].read_idx.tolist() == ref.rec_read[n1:n1 + n2].tolist()
assert recs[0].token.tolist() + recs[1].token.tolist() == ref.rec_token[n1:n1 + n2].tolist()
# one-collective form: the engines' send buffers, concatenated as the all-gather delivers them
from crass_amd.distributed import _DevView
def gathered(cap):
    xs = [engs[r].exchange_setup(2, r, cap) for r in range(2)]
    for r in range(2):
        engs[r].seed_scan()
    bufs = [torch.as_tensor(_DevView(p_, (nb_,), "|u1"), device="cuda") for p_, nb_ in xs]
    recv = torch.cat(bufs).contiguous()
    torch.cuda.synchronize()
    return recv
recv = gathered(4096)
for r in range(2):
    v2 = engs[r].merge_gathered(recv.data_ptr())
    assert not isinstance(v2, int) and engs[r].counters()["used_device_merge"] == 1
    r2 = engs[r].recruit()
    v2 = engs[r].merge_view()
    assert v2.tokens == views[r].tokens and v2.groups == views[r].groups and v2.patterns == views[r].patterns
    assert v2.cand_token.tolist() == views[r].cand_token.tolist()
    assert r2.read_idx.tolist() == recs[r].read_idx.tolist() and r2.token.tolist() == recs[r].token.tolist()
recv = gathered(64)                                  # far too few rows: every rank must report the same need
need = [engs[r].merge_gathered(recv.data_ptr()) for r in range(2)]
assert isinstance(need[0], int) and need[0] == need[1] == max(len(lists[0][1]), len(lists[1][1]))
for e in engs:
    e.close()
print("OK")
"""


def test_device_resident_exchange_two_shards_one_process():
    """the device-resident form of the exchange (crass_hip_merge_distinct_device): two contexts hold the two
    shards, the concatenation of their distinct lists lives in a device tensor, both merge from it on the
    device; tables must equal the host-side exchange's and the oracle's"""
    r = subprocess.run([sys.executable, "-c", DEVICE_EXCHANGE % dict(root=ROOT)], capture_output=True, timeout=600)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr.decode()[-3000:]


def test_bench_single_rank_rccl_exchange():
    """bench.py's N>1 step (RCCL all-gather between device buffers + device-resident merge) with a process
    group of ONE rank: the collective calls, tensor views of the engine's buffers and pointer hand-over are the
    code the 8-GPU run executes"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import crass_amd as ca
from crass_amd.distributed import allgather_distinct_device, GatheredExchange
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, L = 300000, 150
spec = ca.synth_spec(read_len=L)
eng = ca.SearchEngine(device=0)
eng.load_packed_uniform(ca.synth_packed(spec, 0, n), n, L)
eng.seed_scan(fetch=False)
g = allgather_distinct_device(eng, dist, torch.device("cuda", 0))
assert g is not None
g_chars, g_lens, my_off = g
eng.merge_distinct_device(g_chars.data_ptr(), g_lens.data_ptr(), g_chars.shape[1], g_chars.shape[0], my_off, fetch=False)
eng.recruit(fetch=False)
c1 = eng.counters(); v1 = eng.merge_view()
eng.seed_scan(fetch=False); eng.merge(fetch=False); eng.recruit(fetch=False)
c2 = eng.counters(); v2 = eng.merge_view()
assert c1["used_device_merge"] == 1 and c2["used_device_merge"] == 1
assert (c1["n_pass1_found"], c1["n_pass2_found"], c1["n_patterns"]) == (c2["n_pass1_found"], c2["n_pass2_found"], c2["n_patterns"])
assert v1.tokens == v2.tokens and v1.groups == v2.groups and v1.patterns == v2.patterns
assert c1["n_pass2_found"] > 0
# bench.py's form: one collective per step; the capacity starts too small on purpose
xg = GatheredExchange(eng, dist, torch.device("cuda", 0), cap_rows=128)
tries = 0
eng.seed_scan(fetch=False)
while not xg.step():
    tries += 1
    eng.seed_scan(fetch=False)
assert tries == 1 and xg.cap_rows >= len(v1.tokens)
eng.recruit(fetch=False)
c3 = eng.counters(); v3 = eng.merge_view()
assert c3["used_device_merge"] == 1 and v3.tokens == v1.tokens and v3.patterns == v1.patterns
assert (c3["n_pass1_found"], c3["n_pass2_found"]) == (c1["n_pass1_found"], c1["n_pass2_found"])
dist.destroy_process_group()
print("OK")
""" % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr.decode()[-3000:]


EXCHANGE_WITH_N = r"""
import os, sys, random
sys.path.insert(0, %(root)r)
import numpy as np
import torch
torch.cuda.init()
from tests import orc
import crass_amd as ca
from crass_amd.distributed import _DevView
ca.load()
n, L = 90000, 150
rng = random.Random(3)
spec = ca.synth_spec(read_len=L, crispr_per_million=40000)
asc = ca.unpack_ascii(ca.synth_packed(spec, 0, n), 10, L, n)
seqs = [bytearray(asc[i * L:(i + 1) * L].tobytes()) for i in range(n)]
ref0 = orc.pipeline([bytes(s) for s in seqs])
for k in range(ref0.n_pass1):                      # an N at the same offset of every repeat of a read: DR variants with an N
    if k %% 4:
        continue
    r, ss, d = int(ref0.rec_read[k]), ref0.ss(k), rng.randrange(int(ref0.rec_replen[k]))
    for q in range(0, len(ss), 2):
        p = ss[q] + d
        p = p if ref0.rec_lowlexi[k] else L - 1 - p
        if 0 <= p < L:
            seqs[r][p] = ord("N")
for i in rng.sample(range(n), 2000):               # and N's anywhere
    seqs[i][rng.randrange(L)] = ord("N")
seqs = [bytes(s) for s in seqs]
ref = orc.pipeline(seqs)
assert sum(1 for t in ref.tokens if b"N" in t) >= 20
engs, packs = [], []
for r in range(3):
    lo, hi = n * r // 3, n * (r + 1) // 3
    e = ca.SearchEngine(device=0)
    pk = ca.PackedReads(seqs[lo:hi])
    e.load_reads(pk, None, read_index_base=lo)
    engs.append(e); packs.append(pk)
xs = [engs[r].exchange_setup(3, r, 8192) for r in range(3)]
for e in engs:
    e.seed_scan()
recv = torch.cat([torch.as_tensor(_DevView(p_, (nb_,), "|u1"), device="cuda") for p_, nb_ in xs]).contiguous()
torch.cuda.synchronize()
views, recs = [], []
for r in range(3):
    v = engs[r].merge_gathered(recv.data_ptr())
    assert not isinstance(v, int) and engs[r].counters()["used_device_merge"] == 1
    recs.append(engs[r].recruit())
    assert engs[r].counters()["used_device_merge"] == 1
    views.append(engs[r].merge_view())
for r in range(3):
    assert views[r].tokens == ref.tokens and views[r].groups == ref.groups
    assert sorted(views[r].patterns) == sorted(ref.patterns)
n1, n2 = ref.n_pass1, ref.n_pass2
assert sum((v.cand_token.tolist() for v in views), []) == ref.rec_token[:n1].tolist()
assert sum((q.read_idx.tolist() for q in recs), []) == ref.rec_read[n1:n1 + n2].tolist()
assert sum((q.token.tolist() for q in recs), []) == ref.rec_token[n1:n1 + n2].tolist()
assert sum((q.low_lexi.tolist() for q in recs), []) == ref.rec_lowlexi[n1:n1 + n2].tolist()
for e in engs:
    e.close()
print("OK")
"""


def test_exchange_three_shards_with_n_variants():
    """three shards, reads with N (also inside the repeats, so that DR variants with an N cross the exchange):
    one-collective form + device merge over the gathered lists against the oracle over all reads"""
    r = subprocess.run([sys.executable, "-c", EXCHANGE_WITH_N % dict(root=ROOT)], capture_output=True, timeout=600)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr.decode()[-3000:]


def _bench_line(r):
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_bench_spawns_its_own_ranks_strong_scaling():
    """`python bench.py --gpus 2 --launcher torch` from a plain interpreter (no torchrun): the parent starts the rank processes
    before it touches the GPU; --total-reads = strong scaling (each rank holds total/N reads); totals equal the one-process run
    (the default launcher, one process driving a group of contexts, is covered by tests/test_gpu_group.py)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher", "torch", "--dist-backend", "gloo", "--share-gpu",
                        "--total-reads", "1000000", "--steps", "2", "--warmup", "1"], capture_output=True, timeout=900)
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_reads"] == 1000000
    assert d["config"]["reads_per_gpu"] == 500000 and d["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--total-reads", "1000000", "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "100000"], capture_output=True, timeout=900)
    d1 = _bench_line(one)
    assert d1["scaling"] == "strong" and "roofline" in d1 and "cpu_baseline" in d1
    assert 0 < d1["roofline"]["path_frac"] < 1 and d1["roofline"]["frac"] > 0
    assert (d["config"]["pass1_found"], d["config"]["pass2_found"], d["config"]["patterns"]) == \
        (d1["config"]["pass1_found"], d1["config"]["pass2_found"], d1["config"]["patterns"])
    cal = d1["cpu_baseline"]["calibration"]
    assert "error" not in cal and cal["bmp_ratio"] > 0 and cal["ac_ratio"] > 0


def test_bench_group_failure_falls_back_to_fresh_rank_processes():
    """`python bench.py --gpus 2` (group launcher) whose group cannot be created — here an injected
    CRASS_ERR_RCCL — prints RCCL's text and runs the torch launcher in FRESH child processes; the line carries the reason and
    the same totals as the one-process run"""
    env = dict(os.environ, CRASS_GROUP_INJECT_RCCL_FAIL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu",
                        "--total-reads", "1000000", "--steps", "2", "--warmup", "1"], capture_output=True, timeout=900, env=env)
    d = _bench_line(r)
    assert b"group launcher unavailable" in r.stderr and b"injected failure" in r.stderr
    assert d["n_gpus"] == 2 and d["config"]["launcher"].startswith("torch.distributed")
    assert "injected failure" in d["launcher_fallback"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--total-reads", "1000000", "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "0", "--single-shots", "0"], capture_output=True, timeout=900)
    d1 = _bench_line(one)
    assert (d["config"]["pass1_found"], d["config"]["pass2_found"], d["config"]["patterns"]) == \
        (d1["config"]["pass1_found"], d1["config"]["pass2_found"], d1["config"]["patterns"])


def test_bench_rccl_strong_scaling_one_rank_per_gpu():
    """the RCCL form of the same launch on however many GPUs this box has (1 on the test pool: a one-rank communicator,
    `--gpus 1` under an external launcher environment)"""
    import torch
    ng = torch.cuda.device_count()
    if ng >= 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-reads", "2000000", "--steps", "3",
                            "--warmup", "2"], capture_output=True, timeout=900)
        d = _bench_line(r)
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rccl_ranks"] == 2
        assert d["one_gpu_same_job"]["pass1_found"] == d["config"]["pass1_found"]
        assert d["one_gpu_same_job"]["pass2_found"] == d["config"]["pass2_found"]
    else:
        pytest.skip("one GPU: the N>1 RCCL launch needs a multi-GPU node (covered by the gloo run and the 1-rank RCCL exchange test)")


@pytest.mark.parametrize("cfg,extra", [(3, ["--total-reads", "20000"]), (4, ["--total-reads", "2000000"])])
def test_bench_other_configs_small(cfg, extra):
    """--config 3 (10 kbp reads, arrays of 20-60 repeats) and --config 4 (500 DRs, 4 GC classes) at reduced size, with
    the oracle check on a prefix"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg), "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "2000" if cfg == 3 else "100000", "--check"] + extra, capture_output=True, timeout=900)
    d = _bench_line(r)
    assert d["config"]["baseline_config"] == cfg and d["config"]["pass1_found"] > 0 and d["value"] > 0
    assert "roofline" in d and "cpu_baseline" in d
