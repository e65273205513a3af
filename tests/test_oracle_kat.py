"""Oracle vs the reference's own known-answer tests (src/test/test_libcrispr.cpp,
7 TEST_CASEs / 139 assertions; fixture made by tests/golden/make_kat_libcrispr.py)."""
import ctypes as C
import json
import os

import pytest

from tests import orc

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_libcrispr.json")))


def test_fixture_has_all_139_assertions():
    assert KAT["n_assertions"] == 139
    assert len(KAT["cases"]) == 20


@pytest.mark.parametrize("case", KAT["cases"], ids=lambda c: c["func"] + ":" + c["section"][:40])
def test_kat(case):
    L = orc.lib()
    seq = case["read"].encode()
    cap = 64
    ss = (C.c_uint32 * cap)(*case["start_stops_in"])
    nss = C.c_int(len(case["start_stops_in"]))
    # ReadHolder::startStopsAdd clamps stops to L-1 (ReadHolder.cpp:292-295)
    for k in range(1, nss.value, 2):
        assert ss[k] <= len(seq) - 1
    if case["func"] == "scanRight":
        rc = L.orc_scan_right(seq, len(seq), ss, C.byref(nss), cap, case["pattern"].encode(),
                              len(case["pattern"]), case["min_spacer"], case["scan_range"])
        assert rc == 0
    else:
        rl = L.orc_extend_pre_repeat(seq, len(seq), ss, nss.value, case["window"], case["min_spacer"])
        assert rl == case["repeat_length"]
    got = list(ss[:nss.value])
    if "size" in case:
        assert len(got) == case["size"]
    assert got[:len(case["start_stops_out"])] == case["start_stops_out"]
