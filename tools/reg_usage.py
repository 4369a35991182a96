#!/usr/bin/env python3
"""Registers / LDS / scratch per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/reg_usage.py timeband_amd/csrc/lstm.hip [filter]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", d + "/x.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
t = r.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip()
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(anonymous namespace\)::", "", dn)
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if flt not in dn:
        continue

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-52s vgpr %3s agpr %3s spill %3s scratch %4s lds %6s occ %s" % (
        dn[:52], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
