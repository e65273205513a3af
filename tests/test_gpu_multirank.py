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
