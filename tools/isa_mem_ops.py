#!/usr/bin/env python3
"""Developer tool: memory-instruction forms per kernel in a device assembly file (tools/asm_f64.sh output).
Usage: tools/isa_mem_ops.py /tmp/f32.s 'pk_kernel<4, 1, true, 256>' [...]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
wanted = sys.argv[2:]
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN3psa\w+):\s+; @", text, re.M)]
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.splitlines()
for k, ((pos, _), dem) in enumerate(zip(starts, names)):
    if wanted and not any(w in dem for w in wanted):
        continue
    end = text.find(".end_amdhsa_kernel", pos)
    end = text.find("s_endpgm", pos) if end < 0 else end
    nxt = starts[k + 1][0] if k + 1 < len(starts) else len(text)
    body = text[pos:min(nxt, len(text))]
    ops = {}
    for m in re.finditer(r"^\s+((?:global|buffer|scratch|flat|ds)_\w+)\b(.*)$", body, re.M):
        key = m.group(1) + (" nt" if re.search(r"\bnt\b", m.group(2)) else "") + (" saddr" if re.search(r", s\[\d+:\d+\]", m.group(2)) else "")
        ops[key] = ops.get(key, 0) + 1
    print(dem)
    for key in sorted(ops):
        print(f"    {key:40s} {ops[key]}")
