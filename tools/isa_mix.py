#!/usr/bin/env python3
"""Instruction mix per kernel of a hipcc -S device assembly file (development aid).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/conv.s codeps_amd/csrc/conv.hip
    python tools/isa_mix.py /tmp/conv.s [substring ...]"""
import re
import subprocess
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):\s", l + " "))]
names = subprocess.run(["c++filt"] + [n for _, n in starts], capture_output=True, text=True).stdout.strip().split("\n")
want = sys.argv[2:]
for k, ((i, n), d) in enumerate(zip(starts, names)):
    j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    if want and not any(w in d for w in want):
        continue
    body = lines[i:j]
    c = Counter()
    for l in body:
        m = re.match(r"\s+([a-z_0-9]+)\s", l)
        if m and not m.group(1).startswith(("amdhsa", "p2align", "section", "globl", "type", "size", "set")):
            c[m.group(1)] += 1
    info = {}
    for l in body:
        m = re.match(r"\s+\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|accum_offset)\s+(\d+)", l)
        if m:
            info[m.group(1)] = int(m.group(2))
    print(d.replace("(anonymous namespace)::", "")[:110], len(body), info)
    print("   ", dict(c.most_common(26)))
