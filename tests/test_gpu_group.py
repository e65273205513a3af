"""Several GPUs from ONE process (crass_hip_group_*, include/crass_hip.h): N contexts, contiguous read shards,
one all-gather of the distinct candidate DR strings issued by the engine, one host view.  On the one-GPU test box the
contexts share the device and the collective is the local-copy stand-in; the RCCL path itself runs with one rank.  The
group's hand-off must equal the single-context result and the oracle's on the whole read set, record by record."""
import os
import random

import numpy as np
import pytest

from tests import orc, fastx
from tests.parity import assert_same_pipeline
from tests.test_gpu_parity import synth_reads, DATA, KNOWN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    crass_amd.load()
    return crass_amd


@pytest.mark.parametrize("n_ctx,fused", [(2, True), (3, False), (4, True)])
def test_group_on_one_device_equals_oracle_and_single_context(ca, n_ctx, fused):
    seqs = synth_reads(ca, 120000, read_len=150, crispr_per_million=30000)
    ref = orc.pipeline(seqs)
    grp = ca.search_pipeline_group(seqs, [0] * n_ctx, local_copies=True, fused=fused)
    assert_same_pipeline(grp, ref)
    one = ca.search_pipeline(seqs)
    np.testing.assert_array_equal(grp.rec_read, one.rec_read)
    np.testing.assert_array_equal(grp.rec_token, one.rec_token)
    assert grp.tokens == one.tokens and grp.groups == one.groups and grp.patterns == one.patterns
    # recruits carry their token's string although only rank 0 builds the host view
    for k in range(0, grp.rec.n, 97):
        assert grp.rec.dr(k) == grp.tokens[int(grp.rec.token[k]) - 2]
    assert len(grp.counters) == n_ctx and all(c["used_device_merge"] == 1 for c in grp.counters)
    assert sum(c["n_reads"] for c in grp.counters) == len(seqs)


def test_group_of_one_rank_runs_the_rccl_collective(ca):
    """the RCCL path proper (ncclCommInitAll + ncclGroupStart / ncclAllGather / ncclGroupEnd) with the one rank a
    one-GPU box allows"""
    seqs = synth_reads(ca, 60000, read_len=150, crispr_per_million=30000)
    packed = ca.PackedReads(seqs)
    with ca.SearchGroup([0]) as g:
        assert g.rccl_ranks == 1
        g.load_reads(packed)
        for _ in range(3):                      # (a reused group: learnt bounds, queued merge)
            g.step()
        res = g.result()
    assert_same_pipeline(res, orc.pipeline(seqs))


def test_duplicate_device_needs_the_stand_in(ca):
    with pytest.raises(ca.CrassError) as e:
        ca.SearchGroup([0, 0])
    assert e.value.status == 1 and "listed twice" in str(e.value)


def test_group_ragged_reads_exceptions_and_duplicate_headers_across_shards(ca):
    rng = random.Random(11)
    base = synth_reads(ca, 30000, read_len=150, crispr_per_million=50000)
    seqs, hdrs = [], []
    for i, s in enumerate(base):
        s = bytearray(s[:rng.randint(40, 150)]) if rng.random() < 0.5 else bytearray(s)
        if rng.random() < 0.03:
            s[rng.randrange(len(s))] = ord("N")
        seqs.append(bytes(s))
        # a later read re-uses an earlier header — from anywhere in the file, so also from another shard
        hdrs.append(b"r%d" % (i if rng.random() > 0.08 else rng.randrange(0, i + 1)))
    ref = orc.pipeline(seqs, hdrs)
    for n_ctx in (2, 3):
        grp = ca.search_pipeline_group(seqs, [0] * n_ctx, headers=hdrs, local_copies=True)
        assert_same_pipeline(grp, ref)
    # the cross-shard suppression is really exercised: without headers more reads are recruited
    assert orc.pipeline(seqs).n_pass2 > ref.n_pass2


@pytest.mark.parametrize("fname", sorted(KNOWN))
def test_group_reference_inputs(ca, fname):
    recs = fastx.read_fastx(os.path.join(DATA, fname))
    seqs = [r[2] for r in recs]
    hdrs = [r[0] for r in recs]
    grp = ca.search_pipeline_group(seqs, [0, 0, 0], headers=hdrs, local_copies=True)
    assert_same_pipeline(grp, orc.pipeline(seqs, hdrs))
    got = (len(recs), grp.n_pass1, len(set(grp.rec_token[:grp.n_pass1].tolist())), grp.n_groups, grp.n_patterns, grp.n_pass1 + grp.n_pass2)
    assert got == KNOWN[fname]


def test_group_exchange_overflow_grows_the_buffers(ca):
    """more distinct DR strings than the exchange's row capacity: every rank learns it from the gathered headers, the
    buffers grow and pass 1 is repeated inside crass_hip_group_merge"""
    seqs = synth_reads(ca, 60000, read_len=150, crispr_per_million=100000, n_dr=300)
    os.environ["CRASS_GROUP_CAP_ROWS"] = "64"
    try:
        grp = ca.search_pipeline_group(seqs, [0, 0], local_copies=True, fused=False)
    finally:
        os.environ.pop("CRASS_GROUP_CAP_ROWS", None)
    assert grp.n_tokens > 128
    assert_same_pipeline(grp, orc.pipeline(seqs))


_HYGIENE = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import crass_amd as ca
from tests.test_gpu_parity import synth_reads
ca.load()
seqs = synth_reads(ca, 60000, read_len=150, crispr_per_million=100000, n_dr=300)
one = ca.search_pipeline(seqs)
os.environ["CRASS_GROUP_CAP_ROWS"] = "64"
grp = ca.search_pipeline_group(seqs, [0, 0], local_copies=True, fused=False)
os.environ.pop("CRASS_GROUP_CAP_ROWS")
big = ca.search_pipeline_group(seqs, [0, 0, 0], local_copies=True, fused=True)
for g in (grp, big):
    assert g.tokens == one.tokens and g.groups == one.groups and g.patterns == one.patterns
    np.testing.assert_array_equal(g.rec_read, one.rec_read)
    np.testing.assert_array_equal(g.rec_token, one.rec_token)
print("hygiene ok", grp.n_tokens)
"""


@pytest.mark.parametrize("env", [{"CRASS_POISON": "1"}, {"CRASS_GUARD_PAGES": "1"}, {"CRASS_GUARD_PAGES": "1", "CRASS_POISON": "1"}])
def test_overflowed_exchange_reads_no_unwritten_rows(env):
    """csrc/devmem.h's debugging allocators, in a child process (the switches are read once): fresh buffers filled with
    0xA5 instead of zero and / or every buffer between unmapped pages.  An exchange whose lists did not fit used to leave
    the sum of the headers as the device-side row count: the de-duplication queued behind it walked rows nobody had
    written — harmless zeros in a young process, a wild length and a memory fault once hipMalloc recycled a block."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _HYGIENE], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "hygiene ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_group_gathered_bound_overflow(ca):
    """the bound for the GLOBAL distinct list (merge queued behind the exchange's de-duplication) too small: the merge is
    launched again with the exact count, same result"""
    seqs = synth_reads(ca, 60000, read_len=150, n_dr=40, crispr_per_million=60000)
    os.environ["CRASS_TEST_BOUNDS"] = "0,0,0,16"
    try:
        grp = ca.search_pipeline_group(seqs, [0, 0], local_copies=True)
    finally:
        os.environ.pop("CRASS_TEST_BOUNDS", None)
    assert_same_pipeline(grp, orc.pipeline(seqs))
    assert all(c["n_bound_overflows"][3] >= 1 for c in grp.counters)


def test_group_deferred_pass1_whose_survivor_bound_overflows_is_repeated_by_every_rank(ca):
    """the ranks' seed scans return with pass 1 still queued (crass_hip_exchange_set_deferred), the exchange and the merge are
    queued behind it; a survivor bound that turns out too small is seen by the kernel that fills the send buffer, which marks the
    header: every rank's merge reports an exchange that did not fit and the step is repeated (synchronously on the marked ranks)"""
    seqs = synth_reads(ca, 90000, read_len=150, n_dr=40, crispr_per_million=60000)
    ref = orc.pipeline(seqs)
    os.environ["CRASS_TEST_BOUNDS"] = "64,0,0,0"
    try:
        grp = ca.search_pipeline_group(seqs, [0, 0, 0], local_copies=True)
    finally:
        os.environ.pop("CRASS_TEST_BOUNDS", None)
    assert_same_pipeline(grp, ref)
    assert all(c["n_bound_overflows"][0] >= 1 for c in grp.counters)
    # the A/B switch: the round-5 order (every seed scan waits for its pass 1 before the exchange is queued)
    os.environ["CRASS_GROUP_SYNC_P1"] = "1"
    try:
        sync = ca.search_pipeline_group(seqs, [0, 0, 0], local_copies=True)
    finally:
        os.environ.pop("CRASS_GROUP_SYNC_P1", None)
    assert_same_pipeline(sync, ref)


def test_deferred_seed_scan_is_settled_by_whoever_reads_pass1(ca):
    """one context with an exchange set up (world of one) in deferred mode: crass_hip_seed_scan returns with pass 1 queued; the
    getters settle it; merge_gathered — on the context's own send buffer, which IS the gathered buffer of a world of one —
    finishes it after queueing its kernels"""
    seqs = synth_reads(ca, 50000, read_len=150, n_dr=30, crispr_per_million=50000)
    ref = orc.pipeline(seqs)
    eng = ca.SearchEngine(device=0)
    eng.load_reads(ca.PackedReads(seqs))
    ptr, nbytes = eng.exchange_setup(1, 0, 16384)
    eng.exchange_set_deferred(True)
    for it in range(3):
        eng.seed_scan(fetch=False)
        if it == 1:
            cand = eng.candidates()                 # settles the pending scan
            assert cand.n == ref.n_pass1
        assert eng.merge_gathered(ptr, fetch=False) is None
        cand = eng.candidates(); mg = eng.merge_view(); rec = eng.recruit(); mg = eng.merge_view()
        assert_same_pipeline(ca.engine.PipelineResult(cand, mg, rec, cand.max_read_len), ref)
    eng.close()
    # a bound that is too small: the send buffer carries the mark, merge_gathered reports it like a list that did not fit
    os.environ["CRASS_TEST_BOUNDS"] = "64,0,0,0"
    try:
        eng = ca.SearchEngine(device=0)
        eng.load_reads(ca.PackedReads(seqs))
        ptr, nbytes = eng.exchange_setup(1, 0, 16384)
        eng.exchange_set_deferred(True)
        eng.seed_scan(fetch=False)
        need = eng.merge_gathered(ptr, fetch=False)
        assert need is not None and need <= 16384
        eng.seed_scan(fetch=False)                  # (synchronous this time)
        assert eng.merge_gathered(ptr, fetch=False) is None
        cand = eng.candidates(); mg = eng.merge_view(); rec = eng.recruit(); mg = eng.merge_view()
        assert_same_pipeline(ca.engine.PipelineResult(cand, mg, rec, cand.max_read_len), ref)
        assert eng.counters()["n_bound_overflows"][0] >= 1
        eng.close()
    finally:
        os.environ.pop("CRASS_TEST_BOUNDS", None)


def test_group_with_an_empty_shard_and_with_no_candidates(ca):
    seqs = synth_reads(ca, 3, read_len=150, crispr_per_million=1000000)
    grp = ca.search_pipeline_group(seqs, [0] * 4, local_copies=True)          # 4 ranks, 3 reads
    assert_same_pipeline(grp, orc.pipeline(seqs))
    rnd = synth_reads(ca, 2000, read_len=150, crispr_per_million=0)
    grp = ca.search_pipeline_group(rnd, [0, 0], local_copies=True)
    assert grp.n_pass1 == 0 and grp.n_pass2 == 0 and grp.n_patterns == 0


def test_bench_group_launcher_on_one_gpu():
    """`bench.py --gpus N` from a plain interpreter drives the group API (one process, N contexts); here with the
    contexts sharing cuda:0 and the local-copy stand-in for the collective"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--local-copies", "--total-reads", "2000000", "--steps", "5",
                        "--warmup", "2", "--cpu-sample", "0"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["launcher"].startswith("group")
    assert d["config"]["total_reads"] == 2000000 and d["config"]["reads_per_gpu"] == 1000000
    one = d["one_gpu_same_job"]
    assert (one["pass1_found"], one["pass2_found"]) == (d["config"]["pass1_found"], d["config"]["pass2_found"])
    assert d["rccl_ranks"] == 0 and d["config"]["merge_fallbacks"] == 0
