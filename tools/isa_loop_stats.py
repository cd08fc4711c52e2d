#!/usr/bin/env python3
"""Developer tool: instruction mix of the largest basic block (the 2x-unrolled z-loop body) of kernels in a gfx950 .s file.
Usage: tools/isa_loop_stats.py file.s substring [substring...]   (.s from hipcc -S --cuda-device-only)"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = re.compile(r'^(_ZN3psa\S+):\s*; @', re.M)
ms = list(pat.finditer(s))
for k, m in enumerate(ms):
    name = m.group(1)
    if not any(sub in name for sub in sys.argv[2:]):
        continue
    body = s[m.end(): ms[k + 1].start() if k + 1 < len(ms) else len(s)].split('.Lfunc_end')[0]
    blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
    best = None
    for j in range(1, len(blocks), 2):
        ins = [l.strip() for l in blocks[j + 1].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
        if best is None or len(ins) > len(best[1]):
            best = (blocks[j], ins)
    ins = best[1]
    c = Counter(l.split()[0] for l in ins)
    dp = sum(v for kk, v in c.items() if kk.endswith('_f64'))
    valu = sum(v for kk, v in c.items() if kk.startswith('v_'))
    print(name[:80], best[0], 'instrs', len(ins), 'valu', valu, 'f64', dp, 'dpp', sum(1 for l in ins if 'quad_perm' in l))
    print('    non-f64:', {kk: v for kk, v in c.items() if not kk.endswith('_f64')})
