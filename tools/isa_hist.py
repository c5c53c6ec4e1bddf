#!/usr/bin/env python3
"""Instruction histogram of one kernel from a hipcc -save-temps .s file.

    hipcc ... -save-temps -c x.hip ; python tools/isa_hist.py x-hip-amdgcn-amd-amdhsa-gfx950.s <substring of mangled name>
"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
want = sys.argv[2]
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, flags=re.M)]
for i, (pos, name) in enumerate(labels):
    if want not in name:
        continue
    end = text.find("s_endpgm", pos)
    body = text[pos:end]
    ops = re.findall(r"^\s+([a-z][a-z_0-9]+)\s", body, flags=re.M)
    c = Counter(ops)
    groups = Counter()
    for k, v in c.items():
        g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else
             "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        groups[g] += v
    print(name, "instructions:", len(ops), dict(groups))
    for k, v in c.most_common(40):
        print(f"  {k:30s}{v}")
