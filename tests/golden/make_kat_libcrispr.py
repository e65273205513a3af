#!/usr/bin/env python3
"""Extract the known-answer vectors held by the reference's own unit tests
(src/test/test_libcrispr.cpp: 7 TEST_CASEs / 139 assertions on scanRight and
extendPreRepeat) into a small JSON fixture: inputs and expected outputs only.

Run in the build container (needs /root/reference):
    python tests/golden/make_kat_libcrispr.py > tests/golden/kat_libcrispr.json
"""
import json
import re
import sys

SRC = "/root/reference/src/test/test_libcrispr.cpp"


def main():
    text = open(SRC).read()
    # strip // comments
    text = re.sub(r"//[^\n]*", "", text)
    cases = []
    n_assert = 0
    # split into TEST_CASE blocks
    tc_iter = list(re.finditer(r'TEST_CASE\("([^"]+)"', text))
    for ti, tc in enumerate(tc_iter):
        body = text[tc.end(): tc_iter[ti + 1].start() if ti + 1 < len(tc_iter) else len(text)]
        reads = {}
        for m in re.finditer(r'ReadHolder\s+(\w+)\(\s*"([ACGTN]+)"\s*,\s*"([^"]*)"\s*\)', body):
            reads[m.group(1)] = (m.group(2), m.group(3))
        sec_iter = list(re.finditer(r'SECTION\("([^"]+)"\)', body))
        for si, sec in enumerate(sec_iter):
            sbody = body[sec.end(): sec_iter[si + 1].start() if si + 1 < len(sec_iter) else len(body)]
            adds = re.findall(r'(\w+)\.startStopsAdd\(\s*(\d+)\s*,\s*(\d+)\s*\)', sbody)
            var = adds[0][0]
            case = {"test_case": tc.group(1), "section": sec.group(1),
                    "read": reads[var][0], "header": reads[var][1],
                    "start_stops_in": [int(x) for a in adds for x in a[1:]]}
            m = re.search(r'scanRight\(\s*\w+\s*,\s*pattern\s*,\s*(\d+)\s*,\s*(\d+)\s*\)', sbody)
            if m:
                pat = re.search(r'pattern\s*=\s*"([ACGT]+)"', sbody).group(1)
                case.update(func="scanRight", pattern=pat, min_spacer=int(m.group(1)), scan_range=int(m.group(2)))
            else:
                m = re.search(r'extendPreRepeat\(\s*\w+\s*,\s*(\d+)\s*,\s*(\d+)\s*\)', sbody)
                case.update(func="extendPreRepeat", window=int(m.group(1)), min_spacer=int(m.group(2)))
                rl = re.search(r'REQUIRE\(repeat_length == (\d+)\)', sbody)
                case["repeat_length"] = int(rl.group(1))
                n_assert += 1
            sz = re.search(r'REQUIRE\(reppos\.size\(\) == (\d+)\)', sbody)
            if sz:
                case["size"] = int(sz.group(1))
                n_assert += 1
            exp = {}
            for idx, val in re.findall(r'REQUIRE\(reppos\[(\d+)\] == (\d+)\)', sbody):
                exp[int(idx)] = int(val)
                n_assert += 1
            case["start_stops_out"] = [exp[i] for i in range(len(exp))]
            if "size" in case:
                assert len(exp) == case["size"], (case["section"], exp)
            cases.append(case)
    json.dump({"source": "src/test/test_libcrispr.cpp", "n_assertions": n_assert, "cases": cases},
              sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
