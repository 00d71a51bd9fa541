#!/usr/bin/env python3
"""Static instruction mix of the alignment kernels from the device assembly:
   (cd graphtyper_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S gtx_api.hip -o /tmp/gtx_api.s); python tools/static_counts.py /tmp/gtx_api.s"""
import re
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/gtx_api.s").read()
for name in sys.argv[2:] or ["gtx_align_hinted_kernel", "gtx_align_express4q_kernel", "gtx_align_kernelE", "gtx_score_kernel"]:
    m = re.search(r"^(_ZN3gtx\d+%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm" % name, txt, re.S | re.M)
    if not m:
        print(name, "not found")
        continue
    lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((".", ";"))]
    lines = [l for l in lines if not l.endswith(":")]
    count = lambda *p: sum(1 for l in lines if l.startswith(p))
    print("%-30s static: VALU %5d  SALU %5d  LDS %4d  VMEM %4d  total %5d" % (name, count("v_"), count("s_"), count("ds_"), count("global_", "buffer_", "flat_", "scratch_"), len(lines)))
