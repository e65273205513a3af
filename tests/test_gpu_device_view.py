"""crass_merge_view assembled on the device (k_dmx_*, dmerge.hip) against the view the host rebuilds from the same merge's
per-token results (CRASS_NO_DEVICE_VIEW=1, merge.cpp) and against the oracle: tokens, token offsets, groups, the pattern list
with its order (WorkHorse.cpp:690-697 per group; remove_redundant's order inside a group), pattern groups, candidates' tokens."""
import os

import numpy as np
import pytest

from tests import orc
from tests.parity import assert_same_pipeline
from tests.test_gpu_parity import synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    import crass_amd
    crass_amd.load()
    return crass_amd


def run(ca, seqs, **env):
    # (a single context builds the view on the host by default — it is hidden behind its own pass 2; CRASS_DEVICE_VIEW=1 makes it
    # take the path a rank of a multi-rank job takes)
    env = dict({"CRASS_DEVICE_VIEW": "1"}, **env)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return ca.search_pipeline(seqs)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def assert_same_view(a, b):
    assert a.tokens == b.tokens
    assert a.groups == b.groups
    assert a.patterns == b.patterns
    assert list(a.pat_group) == list(b.pat_group)
    np.testing.assert_array_equal(a.rec_token, b.rec_token)
    np.testing.assert_array_equal(a.rec_read, b.rec_read)
    assert a.n_pass2 == b.n_pass2


@pytest.mark.parametrize("L,n_dr,n,cpm", [(150, 50, 200000, 30000), (101, 3, 60000, 50000), (250, 300, 120000, 40000), (150, 1, 50000, 200000)])
def test_device_view_equals_host_view_and_oracle(ca, L, n_dr, n, cpm):
    seqs = synth_reads(ca, n, read_len=L, n_dr=n_dr, crispr_per_million=cpm)
    dev = run(ca, seqs)
    host = run(ca, seqs, CRASS_NO_DEVICE_VIEW=1)
    assert dev.counters["used_device_merge"] == 1 and host.counters["used_device_merge"] == 1
    assert dev.counters["used_device_view"] == 1 and dev.counters["n_view_fallbacks"] == 0
    assert host.counters["used_device_view"] == 0
    assert_same_view(dev, host)
    assert_same_pipeline(dev, orc.pipeline(seqs))


def test_large_view(ca):
    """>= 8192 tokens: several 1024-token tiles, groups of > 100 members, the blob beyond its first-call pinned size"""
    seqs = synth_reads(ca, 1_500_000, read_len=150, n_dr=50, crispr_per_million=250000)
    dev = run(ca, seqs)
    host = run(ca, seqs, CRASS_NO_DEVICE_VIEW=1)
    assert len(dev.tokens) >= 8192 and dev.counters["used_device_view"] == 1
    assert_same_view(dev, host)


def test_groups_on_both_sides_of_the_sorted_range(ca):
    """k_dmx_sort ranks the groups of 65 .. 4 096 members (two sorts in LDS), k_dmx_rank the smaller and the larger ones (counting):
    a set with one group beyond 4 096 members beside small ones"""
    seqs = synth_reads(ca, 600_000, read_len=150, n_dr=1, crispr_per_million=400000)
    seqs += synth_reads(ca, 200000, read_len=150, n_dr=40, crispr_per_million=60000)
    dev = run(ca, seqs, CRASS_VIEW_SORT_MAX=512)        # (the range's upper end lowered: a group of 4 097 members takes millions of reads)
    host = run(ca, seqs, CRASS_NO_DEVICE_VIEW=1)
    sizes = sorted(len(g) for g in host.groups)
    assert sizes[-1] > 512 and sizes[0] <= 64, sizes
    assert dev.counters["used_device_view"] == 1
    assert_same_view(dev, host)


def test_group_beyond_the_rank_cap_is_built_by_the_host(ca):
    seqs = synth_reads(ca, 60000, read_len=150, n_dr=2, crispr_per_million=300000)
    dev = run(ca, seqs, CRASS_VIEW_GROUP_CAP=8)
    ref = run(ca, seqs, CRASS_NO_DEVICE_VIEW=1)
    assert dev.counters["used_device_merge"] == 1
    assert dev.counters["used_device_view"] == 0 and dev.counters["n_view_fallbacks"] >= 1
    assert max(len(g) for g in ref.groups) > 8
    assert_same_view(dev, ref)


def test_repeated_steps_on_one_context(ca):
    """steady state: the merge (and its view export) is queued by the seed scan ahead of the counts; every step's view is the same"""
    n, L = 300000, 150
    spec = ca.synth_spec(read_len=L, crispr_per_million=30000)
    words = ca.synth_packed(spec, 0, n)
    views = []
    os.environ["CRASS_DEVICE_VIEW"] = "1"
    try:
        eng_ctx = ca.SearchEngine()
    finally:
        os.environ.pop("CRASS_DEVICE_VIEW", None)
    with eng_ctx as eng:
        eng.load_packed_uniform(words, n, L)
        for _ in range(4):
            eng.seed_scan(fetch=False)
            eng.merge(fetch=False)
            eng.recruit(fetch=False)
            m = eng.merge_view()
            c = eng.counters()
            assert c["used_device_view"] == 1 and c["n_bound_overflows"] == [0, 0, 0, 0]
            views.append((m.tokens, m.groups, m.patterns, list(m.pat_group), m.cand_token.tolist(), eng.recruits().token.tolist()))
    for v in views[1:]:
        assert v == views[0]
