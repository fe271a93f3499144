#!/usr/bin/env python3
"""Kernel resource usage (VGPR/SGPR/scratch/LDS/occupancy) of one .hip file, from the compiler's own remarks:
    tools/kres.py sdrpp_radiosonde_amd/csrc/demod_kernel.hip"""
import os
import re
import subprocess
import sys

f = os.path.abspath(sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fno-slp-vectorize", "-c", os.path.basename(f), "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=os.path.dirname(f), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], text=True).strip()
        except Exception:
            pass
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    g = r.get
    print("%-58s VGPR %3s AGPR %2s SGPR %3s scratch %3s occ %s LDS %6s" % (
        r["name"][:58], g("VGPRs", "?"), g("AGPRs", "?"), g("SGPRs", "?"), g("ScratchSize [bytes/lane]", "?"),
        g("Occupancy [waves/SIMD]", "?"), g("LDS Size [bytes/block]", "?")))
