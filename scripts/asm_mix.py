#!/usr/bin/env python3
"""instruction mix of the hot kernels from `hipcc -S --cuda-device-only` output: scripts/asm_mix.py file.s [name substrings...]"""
import collections
import re
import sys

t = open(sys.argv[1]).read()
want = sys.argv[2:] or ["k_scanILi10", "k_bfilterILi10ELi3", "k_filterILi10", "k_finishILi10ELi1ELb0", "k_dpallILi20"]
for m in re.finditer(r"^(_ZN2vf\S*?): +; @", t, re.M):
    name = m.group(1)
    if not any(k in name for k in want):
        continue
    end = t.find(".Lfunc_end", m.end())
    body = t[m.end():end]
    ops = collections.Counter(re.findall(r"^\s+([vs]_[a-z0-9_]+|global_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+)", body, re.M))
    print(name[:48], sum(ops.values()), ops.most_common(16))
