#!/usr/bin/env python3
"""Build step for the sweep kernels: keep every 8-byte VALU encoding on an 8-byte boundary.

Why (measured, tools/issue_probe.hip -> profiles/r03_issue_probe.log): with ONE wave resident on a SIMD -- the shape of the
headline sweep (65 536 points = 1 024 waves) and of every sweep smaller than the chip -- an 8-byte VALU instruction
(VOP3: v_fma_f64 / v_mul_f64 / v_add_f64, VOP3P: v_pk_fma_f32, DPP moves) that starts 4 bytes off an 8-byte boundary
takes 5 cycles instead of 4; a 4-byte one (v_fmac_f64_e32) takes 4 wherever it lies.  The compiler shrinks every
accumulate-form FMA to the 4-byte encoding, so in its output each such instruction flips the alignment of everything
behind it: 27 % of the 8-byte instructions of the float64 z-loop, 54 % of the two-lane one, sat off by 4.

What: reads the device assembly `hipcc -S --cuda-device-only` wrote, and in every run of 4-byte instructions of odd
length that precedes an 8-byte VALU instruction re-encodes ONE 4-byte VALU instruction in its 8-byte form
(`_e32` -> `_e64`: same operation, same operands, same result; 4 cycles either way).  Where a run has nothing to
re-encode (scalar instructions only) and at least MIN_RUN 8-byte instructions follow, it appends an `s_nop 0`.  Basic
blocks already start on 8-byte boundaries (`-mllvm -align-all-blocks=3`).  Instruction sizes are not guessed: the input is
assembled once and the sizes are read back from the disassembly.

Usage: align_encodings.py in.s out.s [--mcpu gfx950] [--quiet]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin").rstrip("/") + "/"
TMP = tempfile.TemporaryDirectory(prefix="psa_align_")      # removed when the script exits
ENC = re.compile(r"// [0-9A-Fa-f]{12}: ((?:[0-9A-Fa-f]{8}\b ?)+)")      # the encoding column of llvm-objdump -d
MIN_RUN = 3          # an s_nop is only worth its issue slot in front of this many 8-byte instructions


def instruction_lines(lines):
    """line numbers that hold an instruction of a text section"""
    found, in_text = [], False
    for n, l in enumerate(lines):
        t = l.strip()
        if t.startswith(".text") or re.match(r"\.section\s+\.text", t):
            in_text = True
        elif re.match(r"\.(section|amdgpu_metadata|data|bss|rodata)\b", t):
            in_text = False
        if not in_text or not t or t[0] in ".;#" or t.startswith("//") or t.split()[0].endswith(":"):
            continue
        found.append(n)
    return found


SYMBOL = re.compile(r"^([A-Za-z_$][\w$.]*):")


def assemble_and_size(lines, mcpu):
    """sizes in bytes of the instruction lines of `lines`, in order.  The input is assembled without its alignment
    directives (so that the disassembly holds exactly the instructions written) and matched function by function: the
    object orders its sections differently from the text."""
    body = [l for l in lines if not re.match(r"\s*\.p2align", l)]
    src, obj = os.path.join(TMP.name, "sized.s"), os.path.join(TMP.name, "sized.o")
    with open(src, "w") as f:
        f.write("\n".join(body) + "\n")
    r = subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={mcpu}", "-c", src, "-o", obj],
                       capture_output=True, text=True)
    if r.returncode:
        raise SystemExit("align_encodings: assembling the input failed\n" + r.stderr[-2000:])
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", obj], capture_output=True, text=True).stdout
    per_symbol, cur = {}, None
    for l in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            cur = per_symbol.setdefault(m.group(1), [])
            continue
        m = ENC.search(l)
        if m and cur is not None:
            cur.append(4 * len(m.group(1).split()))
    sizes, cur, taken = [], None, {}
    instr = set(instruction_lines(body))
    for n, l in enumerate(body):
        m = SYMBOL.match(l)
        if m and not m.group(1).startswith(".L") and m.group(1) in per_symbol:
            cur = m.group(1)
            taken[cur] = 0
        if n in instr:
            if cur is None or taken[cur] >= len(per_symbol[cur]):
                raise SystemExit(f"align_encodings: instruction outside a disassembled function at line {n + 1}: {l.strip()}")
            sizes.append(per_symbol[cur][taken[cur]])
            taken[cur] += 1
    for name, k in taken.items():
        # the compiler ends .text with a `.fill` of s_code_end words: they disassemble behind the last function
        if k != len(per_symbol[name]) and set(per_symbol[name][k:]) != {4}:
            raise SystemExit(f"align_encodings: {name}: {k} instructions written, {len(per_symbol[name])} disassembled")
    return sizes


def promotable_forms(texts, mcpu):
    """which `_e32` instruction texts assemble in their `_e64` form to exactly 8 bytes"""
    ok = {}
    texts = sorted(texts)
    cand = [re.sub(r"_e32\b", "_e64", t, count=1) for t in texts]
    # one file per attempt round: drop the lines the assembler rejects
    alive = list(range(len(texts)))
    while alive:
        src, obj = os.path.join(TMP.name, "forms.s"), os.path.join(TMP.name, "forms.o")
        with open(src, "w") as f:
            f.write(f'.amdgcn_target "amdgcn-amd-amdhsa--{mcpu}"\n.text\n')
            for i in alive:
                f.write(cand[i] + "\n")
        r = subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={mcpu}", "-c", src, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode == 0:
            dis = subprocess.run([LLVM + "llvm-objdump", "-d", obj], capture_output=True, text=True).stdout
            sz = [4 * len(m.group(1).split()) for m in ENC.finditer(dis)]
            assert len(sz) == len(alive)
            for i, s in zip(alive, sz):
                ok[texts[i]] = cand[i] if s == 8 else None
            break
        badl = sorted({int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error", r.stderr)})
        if not badl:
            raise SystemExit("align_encodings: cannot parse assembler errors\n" + r.stderr[-2000:])
        drop = {alive[b - 3] for b in badl if 0 <= b - 3 < len(alive)}
        for i in drop:
            ok[texts[i]] = None
        alive = [i for i in alive if i not in drop]
    return ok


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mcpu = "gfx950"
    if "--mcpu" in sys.argv:
        mcpu = sys.argv[sys.argv.index("--mcpu") + 1]
        args.remove(mcpu)
    src, dst = args
    lines = open(src).read().split("\n")
    sizes = assemble_and_size(lines, mcpu)

    idx_of = {n: k for k, n in enumerate(instruction_lines(lines))}     # line number -> instruction ordinal
    e32 = {lines[n].strip() for n in idx_of if re.match(r"v_\w+_e32\b", lines[n].strip()) and sizes[idx_of[n]] == 4}
    forms = promotable_forms(e32, mcpu)

    out = list(lines)
    insert_after = {}            # line number -> text to add behind it
    stats = dict(valu8=0, off_before=0, promoted=0, promoted_off=0, nops=0, left=0)
    off = 0                      # byte offset mod 8 inside the current aligned region
    run = []                     # 4-byte instructions since the last 8-byte one / alignment point: (line, offset, promotable)

    # the 8-byte VALU instructions that follow position n without a 4-byte instruction or label in between
    def valu8_run_from(n):
        c = 0
        for m in range(n, len(lines)):
            if m not in idx_of:
                t = lines[m].strip()
                if t.startswith(".p2align") or (t and t.split()[0].endswith(":") and not t.startswith(";")):
                    break
                continue
            if sizes[idx_of[m]] == 8 and lines[m].strip().startswith("v_"):
                c += 1
            else:
                break
        return c

    for n, l in enumerate(lines):
        t = l.strip()
        if re.match(r"\.p2align\s+([3-9]|1\d)\b", t):
            off, run = 0, []
            continue
        if n not in idx_of:
            continue
        size = sizes[idx_of[n]]
        if size == 4:
            run.append((n, off, forms.get(t) is not None and t in forms))
            off = (off + 4) % 8
            continue
        is_valu = t.startswith("v_")
        if is_valu:
            stats["valu8"] += 1
        if is_valu and off == 4:
            stats["off_before"] += 1
            follow = valu8_run_from(n)
            even = [r for r in run if r[2] and r[1] == 0]
            odd = [r for r in run if r[2] and r[1] == 4]
            if even:
                ln = even[-1][0]
                out[ln] = out[ln].replace(lines[ln].strip(), forms[lines[ln].strip()])
                stats["promoted"] += 1
                off = 0
            elif odd and follow >= 2:
                ln = odd[-1][0]
                out[ln] = out[ln].replace(lines[ln].strip(), forms[lines[ln].strip()])
                stats["promoted_off"] += 1
                off = 0
            elif follow >= MIN_RUN:
                insert_after[n - 1] = insert_after.get(n - 1, "") + "\ts_nop 0\n"
                stats["nops"] += 1
                off = 0
            else:
                stats["left"] += 1
        off = (off + size) % 8
        run = []

    # PC-relative address sequences (s_getpc_b64 + s_add_u32 / s_addc_u32 with @rel32 literals whose +4 / +12 assume the three
    # instructions are adjacent) must come through untouched: only VALU encodings change and an s_nop goes directly in front
    # of an 8-byte VALU instruction, so they do by construction -- checked all the same.
    def pc_sequences(text_lines):
        ins = [l.strip() for l in text_lines if l.strip() and l.strip()[0] not in ".;" and not l.split()[0].endswith(":")]
        return [tuple(ins[k:k + 3]) for k, l in enumerate(ins) if l.startswith("s_getpc_b64")]
    final = []
    for n, l in enumerate(out):
        final.append(l)
        if n in insert_after:
            final.extend(insert_after[n].rstrip("\n").split("\n"))
    if pc_sequences(lines) != pc_sequences(final):
        raise SystemExit("align_encodings: a PC-relative address sequence would change")

    with open(dst, "w") as f:
        for n, l in enumerate(out):
            f.write(l + ("\n" if n + 1 < len(out) else ""))
            if n in insert_after:
                f.write(insert_after[n])
    if "--quiet" not in sys.argv:
        print(f"align_encodings {src}: {stats['valu8']} 8-byte VALU instructions, {stats['off_before']} started 4 bytes off; "
              f"re-encoded {stats['promoted']} (+{stats['promoted_off']} at an odd slot), {stats['nops']} s_nop, {stats['left']} left")


if __name__ == "__main__":
    main()
