#!/usr/bin/env python3
"""VGPR / SGPR / spill / scratch / LDS of the kernels in libcrass_hip.so whose name contains any of the arguments
(llvm-readelf notes of the embedded gfx950 code objects).   python tools/kernel_regs.py k_survivor k_hint"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crass_amd import vgpr_guard as g
pats = sys.argv[1:] or [""]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crass_amd", "libcrass_hip.so")
with tempfile.TemporaryDirectory() as td:
    for co in g._code_objects(lib, td):
        notes = subprocess.run([g.LLVM_BIN + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "agpr_count":
                cur = {}
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "wavefront_size" and any(p in cur.get("name", "") for p in pats):
                print("%-60s vgpr %s sgpr %s spill %s scratch %s lds %s" % (cur.get("name", "")[:60], cur.get("vgpr_count"), cur.get("sgpr_count"),
                      cur.get("vgpr_spill_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size")))
