#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR / scratch / LDS / occupancy) from hipcc remarks."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or [f for f in os.listdir(os.path.join(root, "vampire_amd/csrc")) if f.endswith(".hip")]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                          "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
                          os.path.join(root, "vampire_amd/csrc", f)], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
                if key == "lds":
                    nm = re.sub(r"\(.*", "", cur["name"]).replace("void vamp::", "")
                    print(f"{f:22s} {nm[:60]:60s} vgpr={cur.get('vgpr'):4d} scratch={cur.get('scratch'):5d} occ={cur.get('occ')} lds={cur['lds']}")
