#!/usr/bin/env python3
"""profiles/<tag>_lk_resources.json: registers, scratch, occupancy and LDS of every tracker kernel, from the compiler's
resource remarks (hipcc -Rpass-analysis=kernel-resource-usage on the sources as the library builds them).
Usage: tools/lk_resources.py <tag>"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}
for src, base in (("k_lk_fast.hip", "k_lk_fast"), ("k_lk_multi.hip", "k_lk_multi")):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    txt = subprocess.run(cmd, cwd=os.path.join(ROOT, "iceberg_tracking_code_amd", "csrc"), capture_output=True, text=True).stderr
    cur = None
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            # k_lk_fast88: the 88-VGPR entry point; k_lk_fast_sums: the same kernel under a "lk_sums" variant
            mm = re.search(base + r"(88|_sums)?ILi(\d+)ELi(\d+)E(?:Li(\d+)E)?Lb([01])E", m.group(1))
            cur = None
            if mm:
                kind, w, h, f, fb = mm.groups()
                cur = "%s%s<%s,%s%s,%s>" % (base, "_sums" if kind == "_sums" else "", w, h, "," + f if f else "", "true" if fb == "1" else "false")
                out[cur] = {}
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch_bytes", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr_spills", r"SGPRs Spill: (\d+)"),
                         ("vgpr_spills", r"VGPRs Spill: (\d+)"), ("lds_bytes", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                out[cur][key] = int(m.group(1))
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_lk_resources.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
