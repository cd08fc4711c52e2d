"""csrc/align_encodings.py (a build step of the two sweep units) on a hand-written function: after it every
8-byte VALU encoding starts on an 8-byte boundary, nothing but the encoding of 4-byte instructions (and `s_nop`) changed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "psa-simulation-ode-rk-mvp-dispersion_amd", "csrc", "align_encodings.py")
LLVM = "/opt/rocm/lib/llvm/bin/"

ASM = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.globl\tprobe
\t.p2align\t8
\t.type\tprobe,@function
probe:
\tv_fmac_f64_e32 v[0:1], v[2:3], v[4:5]
\tv_fma_f64 v[6:7], v[0:1], v[2:3], v[4:5]
\tv_mul_f64 v[8:9], v[0:1], v[2:3]
\tv_fmac_f64_e32 v[0:1], v[2:3], v[4:5]
\tv_fmac_f64_e32 v[6:7], v[2:3], v[4:5]
\tv_add_f64 v[8:9], v[0:1], v[2:3]
\ts_add_i32 s0, s0, 1
\tv_mul_f64 v[10:11], v[0:1], v[2:3]
\tv_mul_f64 v[12:13], v[0:1], v[2:3]
\tv_mul_f64 v[14:15], v[0:1], v[2:3]
\ts_cmp_lg_u32 s0, 5
\ts_cbranch_scc1 .LBB0_1
\t.p2align\t3
.LBB0_1:
\tv_cndmask_b32_e32 v16, v17, v18, vcc
\tv_mov_b32_e32 v19, v20
\tv_cmp_gt_f64_e32 vcc, v[0:1], v[2:3]
\tv_pk_fma_f32 v[20:21], v[0:1], v[2:3], v[4:5]
\tv_mov_b32_e32 v22, 0x12345678
\tv_fma_f64 v[6:7], v[0:1], v[2:3], v[4:5]
\ts_endpgm
.Lfunc_end0:
\t.size\tprobe, .Lfunc_end0-probe
"""


def _disassemble(path_s, tmp):
    obj = os.path.join(tmp, os.path.basename(path_s) + ".o")
    subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", path_s, "-o", obj],
                   check=True, capture_output=True)
    out = subprocess.run([LLVM + "llvm-objdump", "-d", obj], check=True, capture_output=True, text=True).stdout
    ins = []
    for l in out.splitlines():
        m = re.match(r"^\s+(\S+)\s.*// ([0-9A-Fa-f]{12}): ((?:[0-9A-Fa-f]{8}\b ?)+)", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1), 4 * len(m.group(3).split())))
    return ins


@pytest.mark.skipif(not os.path.exists(LLVM + "clang"), reason="ROCm LLVM tools not present")
def test_every_eight_byte_valu_encoding_ends_up_aligned(tmp_path):
    src, dst = str(tmp_path / "in.s"), str(tmp_path / "out.s")
    with open(src, "w") as f:
        f.write(ASM)
    r = subprocess.run([sys.executable, SCRIPT, src, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    before, after = _disassemble(src, str(tmp_path)), _disassemble(dst, str(tmp_path))
    off_before = [i for i in before if i[1].startswith("v_") and i[2] == 8 and i[0] % 8]
    off_after = [i for i in after if i[1].startswith("v_") and i[2] == 8 and i[0] % 8]
    assert len(off_before) >= 5 and not off_after, (off_before, off_after)
    # same operations in the same order: only the _e32/_e64 suffix of re-encoded instructions differs, plus s_nop padding

    def ops(ins):
        return [re.sub(r"_e(32|64)$", "", o) for _, o, _ in ins if o != "s_nop"]
    assert ops(before) == ops(after)
    # the literal-carrying v_mov (4-byte opcode + 32-bit literal) has no 8-byte VOP3 form: it must be left as it is
    text = open(dst).read()
    assert "v_mov_b32_e32 v22, 0x12345678" in text and "v_fmac_f64_e64" in text
