#!/usr/bin/env python3
"""Developer tool: byte alignment of the hot z-loops (back edges spanning > 2 000 B) of every sweep kernel in a built object.
A stream of 8-byte VALU encodings (VOP3 / VOP3P: every FP64 and packed-FP32 instruction) that starts 4 bytes off an 8-byte
boundary was measured 15 % slower with one wave per SIMD (profiles/r03_loop_alignment.log).
Usage: tools/loop_alignment.py psa-.../csrc/psa_rk4_f64.o [substring ...]"""
import re
import subprocess
import sys
import tempfile

obj, subs = sys.argv[1], sys.argv[2:]
co = tempfile.NamedTemporaryFile(suffix=".co").name
LLVM = "/opt/rocm/lib/llvm/bin/"


def unbundle(src):
    return subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={src}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True).returncode == 0


if not unbundle(obj):       # a host object: the device code object sits in its .hip_fatbin section
    fat = tempfile.NamedTemporaryFile(suffix=".fatbin").name
    subprocess.run([LLVM + "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, "/dev/null"], check=True)
    assert unbundle(fat), "no gfx950 code object found"
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines()
name, bad, total = None, 0, 0
for l in dis:
    m = re.match(r"^[0-9a-f]+ <(_ZN3psa\S+)>:", l)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("void psa::", "").split("(")[0]
        continue
    if name is None or "s_cbranch_scc" not in l or (subs and not any(s in name for s in subs)):
        continue
    mm = re.search(r"s_cbranch_scc[01] (\d+)\s+// ([0-9A-Fa-f]+):", l)
    if not mm or int(mm.group(1)) <= 32767:
        continue
    pc = int(mm.group(2), 16)
    tgt = pc + 4 + (int(mm.group(1)) - 65536) * 4
    if pc - tgt > 2000:
        total += 1
        bad += tgt % 8 != 0
        if tgt % 8 or subs:
            print(f"{name:64s} loop of {pc - tgt:5d} B at {tgt:#x}: mod 8 = {tgt % 8}, mod 64 = {tgt % 64}")
print(f"{obj}: {total} hot loops, {bad} not 8-byte aligned")
