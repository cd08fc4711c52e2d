#!/usr/bin/env python3
"""Developer tool: compile one kernel TU for gfx950 with -Rpass-analysis=kernel-resource-usage and print a compact
table (kernel, VGPRs, AGPRs, spilled VGPRs, scratch bytes, waves/SIMD).  Usage: tools/kernel_resources.py psa_rk4_f64.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "psa-simulation-ode-rk-mvp-dispersion_amd", "csrc")
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{CSRC}",
       *(["-mllvm", "-amdgpu-sched-strategy=" + os.environ.get("SCHED", "max-ilp")] if os.environ.get("SCHED", "max-ilp") != "default" else []), "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(CSRC, src), "-o", "/tmp/_kr.o"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)(?: \[-Rpass)", line) or re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(psa::SweepArgs<\w+>\)", "", n).replace("void psa::", "")
    print(f"{n:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill', r.get('VGPR Spill','?')):>3} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')}")
