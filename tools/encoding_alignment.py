#!/usr/bin/env python3
"""Developer tool: for every hot z-loop (back edge spanning > 2 000 B) of the sweep kernels in a built object, how many
8-byte VALU encodings start 4 bytes off an 8-byte boundary.  With one wave per SIMD such an instruction takes 5 cycles
instead of 4 (tools/issue_probe.hip, profiles/r03_issue_probe.log): every 4-byte encoding (v_fmac_f64_e32 ...) flips the
parity of what follows it.
Usage: tools/encoding_alignment.py psa-.../csrc/psa_rk4_f64.o [substring ...] [--summary]"""
import re
import subprocess
import sys
import tempfile

obj, subs = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
co = tempfile.NamedTemporaryFile(suffix=".co").name
LLVM = "/opt/rocm/lib/llvm/bin/"


def unbundle(src):
    return subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={src}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True).returncode == 0


if not unbundle(obj):
    fat = tempfile.NamedTemporaryFile(suffix=".fatbin").name
    subprocess.run([LLVM + "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, "/dev/null"], check=True)
    assert unbundle(fat), "no gfx950 code object found"
dis = subprocess.run([LLVM + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines()

kernels, name = {}, None
for l in dis:
    m = re.match(r"^[0-9a-f]+ <(_ZN3psa\S+)>:", l)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("void psa::", "").split("(")[0]
        kernels[name] = []
        continue
    m = re.match(r"^\s+(\S+)\s.*// ([0-9A-Fa-f]+): ((?:[0-9A-Fa-f]{8} ?)+)", l)
    if name and m:
        kernels[name].append((int(m.group(2), 16), m.group(1), 4 * len(m.group(3).split()), l))

tot_long = tot_off = loops = 0
for name, ins in kernels.items():
    if subs and not any(s in name for s in subs):
        continue
    for pc, op, size, l in ins:
        mm = re.search(r"s_cbranch_scc[01] (\d+)", l)
        if not mm or int(mm.group(1)) <= 32767:
            continue
        tgt = pc + 4 + (int(mm.group(1)) - 65536) * 4
        if pc - tgt <= 2000:
            continue
        body = [(a, o, s) for a, o, s, _ in ins if tgt <= a <= pc]
        valu = [(a, o, s) for a, o, s in body if o.startswith("v_")]
        long_ = [x for x in valu if x[2] >= 8]
        off = [x for x in long_ if x[0] % 8]
        cyc = 4 * len(valu) + len(off)
        tot_long += len(long_)
        tot_off += len(off)
        loops += 1
        if "--summary" not in sys.argv:
            print(f"{name:66s} loop at {tgt:#07x}: {len(valu):4d} VALU, {len(long_):4d} of 8 B, {len(off):4d} off by 4 "
              f"-> {cyc / len(valu):.3f} cycles per instruction with one wave per SIMD")
print(f"{obj}: {loops} hot loops, {tot_long} 8-byte VALU instructions in them, {tot_off} off by 4")
