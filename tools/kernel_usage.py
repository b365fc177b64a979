"""Summarises hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile) as one line per kernel:
   hipcc ... -Rpass-analysis=kernel-resource-usage -c pt_render.hip -o /dev/null 2> usage.txt; python tools/kernel_usage.py usage.txt"""
import re
import sys

txt = open(sys.argv[1]).read()
keys = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("SGPRs", "SGPR"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "LDS")]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", b.split()[0])[:44]
    vals = []
    for k, label in keys:
        m = re.search(k + r": (\d+)", b)
        vals.append("%s %4s" % (label, m.group(1) if m else "?"))
    print("%-46s %s" % (name, "  ".join(vals)))
