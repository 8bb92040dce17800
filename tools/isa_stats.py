#!/usr/bin/env python
"""Resource table (VGPR / AGPR / SGPR / LDS / scratch / code bytes) of every kernel in a hipcc -save-temps .s file.
    hipcc --offload-arch=gfx950 -O3 -c x.hip -save-temps && python tools/isa_stats.py x-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], text=True).strip()
    except Exception:
        pass
    if flt and flt not in name:
        continue
    print("vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill %s/%s  %s" % (
        g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"),
        g("vgpr_spill_count"), g("sgpr_spill_count"), name[:150]))
