"""Shared comparison helpers: HIP engine result vs oracle result (bit-exact)."""
import numpy as np


def assert_same_pipeline(gpu, orc, check_pass2=True):
    assert orc.error == 0
    assert gpu.n_pass1 == orc.n_pass1, (gpu.n_pass1, orc.n_pass1)
    n1 = orc.n_pass1
    np.testing.assert_array_equal(gpu.rec_read[:n1], orc.rec_read[:n1])
    np.testing.assert_array_equal(gpu.rec_lowlexi[:n1], orc.rec_lowlexi[:n1])
    np.testing.assert_array_equal(gpu.rec_replen[:n1], orc.rec_replen[:n1])
    np.testing.assert_array_equal(gpu.rec_nss[:n1], orc.rec_nss[:n1])
    for k in range(n1):
        assert gpu.ss(k) == orc.ss(k), (k, gpu.ss(k), orc.ss(k))
    np.testing.assert_array_equal(gpu.rec_token[:n1], orc.rec_token[:n1])
    # tokens discovered by pass 1, groups, patterns
    assert gpu.n_groups == orc.n_groups
    assert gpu.groups == orc.groups
    assert gpu.n_patterns == orc.n_patterns
    # layout: per group (ascending GID) the survivors, then their reverse complements (WorkHorse.cpp:690-697); inside a group
    # length ascending, token order among equal lengths (std::sort / std::partition leave that order unspecified; product
    # and oracle both take the stable outcome — DESIGN.md 6)
    assert list(gpu.pat_group) == list(orc.pat_group)
    assert list(gpu.patterns) == list(orc.patterns)
    if check_pass2:
        assert gpu.n_pass2 == orc.n_pass2, (gpu.n_pass2, orc.n_pass2)
        n = n1 + orc.n_pass2
        np.testing.assert_array_equal(gpu.rec_read[n1:n], orc.rec_read[n1:n])
        np.testing.assert_array_equal(gpu.rec_lowlexi[n1:n], orc.rec_lowlexi[n1:n])
        np.testing.assert_array_equal(gpu.rec_nss[n1:n], orc.rec_nss[n1:n])
        for k in range(n1, n):
            assert gpu.ss(k) == orc.ss(k), (k, gpu.ss(k), orc.ss(k))
        np.testing.assert_array_equal(gpu.rec_token[n1:n], orc.rec_token[n1:n])
        assert gpu.n_tokens == orc.n_tokens
        assert gpu.tokens == orc.tokens
    else:
        assert gpu.tokens[:orc.n_tokens] == orc.tokens
