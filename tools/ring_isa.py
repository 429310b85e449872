#!/usr/bin/env python3
"""ISA side of the ring reproducer (tools/kbench/ring_repro.hip).

    python tools/ring_isa.py [--out DIR]

1. compiles ring_repro.hip for gfx950 to assembly (device side only, the flags of lightmotif_amd/build.py);
2. writes the main loop of the faulty (look-ahead MP - 1) and of the shipped (MP - 2) kernel to
   DIR/ring_isa_faulty.txt / ring_isa_cured.txt, marking every VMEM load whose destination register is the ADDRESS
   register of a DS read issued before it with no `s_waitcnt lgkmcnt` proving that read complete
   (the `isa_audit` rule: tools/isa_audit.py applies it to every kernel of the shipped library);
3. builds three code objects next to the tool (tools/kbench/):
      ring_asis.hsaco      the compiler's output, assembled unchanged (control: the module path itself is innocent)
      ring_renamed.hsaco   the faulty kernel with ONLY register names changed: every LDS address is formed in a
                           register of its own (v64 ... v79) instead of in place in the ring register, so that no load
                           lands on the address of a queued DS read.  Same instructions, same order, same wait counts.
      ring_nop.hsaco       the faulty kernel with `s_nop 0` x N between the DS reads and the load behind them
                           (N from --nops, default 4): same registers, the issue distance changed
   `ring_repro --hsaco FILE` runs the three kernels out of such a module.
"""
from __future__ import annotations

import argparse
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KB = ROOT / "tools" / "kbench"
LLVM = Path("/opt/rocm/lib/llvm/bin")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-fast-math",
         f"-I{ROOT / 'include'}", f"-I{ROOT / 'lightmotif_amd' / 'csrc'}"]

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text: str) -> set[int]:
    out: set[int] = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(line: str) -> tuple[str, list[str]]:
    code = line.split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return "", []
    parts = code.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def hazards(lines: list[str]) -> list[tuple[int, int, int]]:
    """(index of the load, index of the DS instruction, register) for every VMEM / FLAT / SMEM-free load whose vdst
    overlaps an ADDRESS or DATA source register of a DS instruction still counted by lgkmcnt.

    DS instructions complete in order, so `s_waitcnt lgkmcnt(n)` retires all but the newest n of them.  Labels do not
    clear the list (a loop's back edge carries it), a waitcnt with lgkmcnt(0) does."""
    pending: list[tuple[int, set[int]]] = []   # (line index, source registers)
    found = []
    for i, line in enumerate(lines):
        op, ops = split_operands(line)
        if not op:
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        if op.startswith("ds_"):
            # reads: ops[0] = vdst, the rest sources; writes: all sources
            srcs = ops[1:] if re.match(r"ds_(read|load|bpermute|permute|swizzle|consume|append|.*_rtn)", op) else ops
            regs: set[int] = set()
            for s in srcs:
                regs |= vregs(s)
            pending.append((i, regs))
            continue
        if re.match(r"(global|flat|buffer|scratch)_load", op) or re.match(r"(global|flat|buffer)_atomic.*", op) and "sc0" in line:
            dst = vregs(ops[0])
            for j, regs in pending:
                hit = dst & regs
                if hit:
                    found.append((i, j, min(hit)))
    return found


def function_range(lines: list[str], name: str) -> tuple[int, int]:
    a = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return a, b


def main_loop(lines: list[str], a: int, b: int) -> tuple[int, int]:
    """The innermost loop: the last label that a backward branch inside the function targets."""
    best = None
    for i in range(a, b):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
        if m:
            t = next((j for j in range(a, i) if lines[j].startswith(m.group(1) + ":")), None)
            if t is not None and (best is None or i - t > best[1] - best[0]):
                best = (t, i)
    assert best, "no loop found"
    return best


def rename_addresses(lines: list[str], a: int, b: int, pool_first: int = 64, pool_n: int = 16) -> int:
    """In lines[a:b]: every `v_mad_u32_u24 vD, vS, <row bytes>, base` that forms an LDS row address gets a destination
    of its own from the pool; the DS reads that use vD up to its next definition follow.  Returns the renames made."""
    alias: dict[int, int] = {}
    nxt = 0
    made = 0
    for i in range(a, b):
        op, ops = split_operands(lines[i])
        if not op:
            continue
        if op == "v_mad_u32_u24" and len(ops) == 4 and ops[2].isdigit():
            d = int(ops[0][1:])
            # sources first (the multiplicand may itself be an aliased... it never is: addresses feed DS reads only)
            new = pool_first + nxt % pool_n
            nxt += 1
            lines[i] = re.sub(r"(v_mad_u32_u24\s+)v%d\b" % d, r"\g<1>v%d" % new, lines[i], count=1)
            alias[d] = new
            made += 1
            continue
        if op.startswith("ds_read"):
            addr = ops[1].split()[0]
            r = int(addr[1:])
            if r in alias:
                head, tail = lines[i].split(",", 1)
                lines[i] = head + "," + re.sub(r"\bv%d\b" % r, "v%d" % alias[r], tail, count=1)
            continue
        # any other write of an aliased register ends the alias (the ring register is reloaded)
        if ops and re.fullmatch(r"v\d+", ops[0]) and int(ops[0][1:]) in alias and not op.startswith(("ds_write", "global_store")):
            del alias[int(ops[0][1:])]
    return made


def insert_nops(lines: list[str], a: int, b: int, n: int) -> int:
    """`s_nop 0` x n in front of every load flagged by hazards() inside lines[a:b]."""
    hz = sorted({i for i, _, _ in hazards(lines[a:b])}, reverse=True)
    for i in hz:
        for _ in range(n):
            lines.insert(a + i, "\ts_nop 0")
    return len(hz)


def set_vgprs(lines: list[str], name: str, n: int) -> None:
    a, _ = function_range(lines, name)
    for i in range(a, len(lines)):
        if ".amdhsa_next_free_vgpr" in lines[i]:
            lines[i] = f"\t\t.amdhsa_next_free_vgpr {n}"
        if ".amdhsa_accum_offset" in lines[i]:
            lines[i] = f"\t\t.amdhsa_accum_offset {n}"
            break
    k = next(i for i, l in enumerate(lines) if re.match(r"\s+\.name:\s+" + re.escape(name) + r"\s*$", l))
    for i in range(k, k + 12):
        if ".vgpr_count:" in lines[i]:
            lines[i] = re.sub(r"\d+", str(n), lines[i])
            break


def assemble(asm: Path, out: Path) -> None:
    obj = out.with_suffix(".o")
    subprocess.run([str(LLVM / "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(asm), "-o", str(obj)], check=True)
    subprocess.run([str(LLVM / "ld.lld"), "-shared", str(obj), "-o", str(out)], check=True)
    obj.unlink()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "profiles"))
    ap.add_argument("--nops", type=int, default=4)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--wide", type=int, default=1)
    ap.add_argument("--loop-only", action="store_true")
    a = ap.parse_args()
    out = Path(a.out)
    mp = (a.m + 1) // 2 * 2
    asm = KB / "ring_repro.s"
    subprocess.run(["hipcc", *FLAGS, "-DLM_RING_LOOKAHEAD_RAW", f"-DRING_M={a.m}", f"-DRING_WIDE={a.wide}", "--cuda-device-only", "-S", str(KB / "ring_repro.hip"), "-o", str(asm)], check=True)
    text = asm.read_text().split("\n")
    # the shipped kernel (part 2 of the reproducer); --loop-only: the stripped kernel of part 1
    if a.loop_only:
        name = lambda pfe: f"_Z9ring_scanILi{a.m}ELi{pfe}ELi{a.wide}EEvPKhPKjiyyyjPy"
    else:
        name = lambda pfe: f"_ZN2lm19score_c32_prefilterILi{a.m}ELi{pfe}ELi{a.wide}EEEvPKhPKjiyyyyjNS_8FusedOutE"
    report = []
    for tag, pfe in (("faulty", mp - 1), ("cured", mp - 2)):
        fa, fb = function_range(text, name(pfe))
        la, lb = main_loop(text, fa, fb)
        body = text[la:lb + 1]
        hz = hazards(body + body)  # twice: what the back edge carries
        marks = {}
        for i, j, r in hz:
            marks.setdefault(i % len(body), set()).add((j % len(body), r))
        lines_out = [f"; {'ring_scan' if a.loop_only else 'lm::score_c32_prefilter'}<M = {a.m}, look-ahead {pfe} (MP {'- 1' if tag == 'faulty' else '- 2'}), WIDE = {a.wide}>: main loop, hipcc {FLAGS[1]} gfx950",
                     f"; '<== WAR' marks a VMEM load whose destination is the address register of a DS read still in flight (no lgkmcnt wait between them):",
                     f";   the round-5 working theory, INNOCENT (profiles/r06_ring_fault_experiments.txt 2-5); the fault is the 64-bit shift marked below"]
        nfree = int(next(l for l in text[fa:fb + 200] if "next_free_vgpr" in l).split()[-1])
        alloc = (nfree + 7) // 8 * 8
        for i, l in enumerate(body):
            if i in marks:
                js = ", ".join(f"v{r} of line {j + 1}" for j, r in sorted(marks[i]))
                l = f"{l:60s} ; <== WAR: {js}"
            ms = re.match(r"\s+v_(lshlrev_b64|lshrrev_b64|ashrrev_i64)\s+v\[\d+:\d+\], v(\d+),", l)
            if ms:
                n = int(ms.group(2))
                last = n % 8 == 7 and n + 1 >= alloc
                l = f"{l:60s} ; <== 64-bit shift by v{n}; {nfree} VGPRs in use, {alloc} allocated" + (": THE LAST ONE -- the fault (isa_audit)" if last else "")
            lines_out.append(f"{i + 1:4d} {l}")
        whole = hazards(text[fa:fb])
        lines_out.append(f"; whole kernel: {len(whole)} load(s) onto the address of a DS read in flight; main loop: {len(marks)}")
        (out / f"r06_ring_isa_{tag}.txt").write_text("\n".join(lines_out) + "\n")
        report.append((tag, pfe, len(whole), len(marks)))
    for tag, pfe, w, m in report:
        print(f"{tag}: look-ahead {pfe}: {w} hazard loads in the kernel, {m} in the main loop")

    assemble(asm, KB / "ring_asis.hsaco")
    ren = list(text)
    fa, fb = function_range(ren, name(mp - 1))
    made = rename_addresses(ren, fa, fb)
    set_vgprs(ren, name(mp - 1), 80)
    fa, fb = function_range(ren, name(mp - 1))
    left = hazards(ren[fa:fb])
    print(f"renamed: {made} address registers moved to v64..v79, hazards left in the faulty kernel: {len(left)}")
    p = KB / "ring_renamed.s"
    p.write_text("\n".join(ren))
    assemble(p, KB / "ring_renamed.hsaco")
    nop = list(text)
    fa, fb = function_range(nop, name(mp - 1))
    n = insert_nops(nop, fa, fb, a.nops)
    print(f"nop: {a.nops} x s_nop in front of {n} loads")
    p = KB / "ring_nop.s"
    p.write_text("\n".join(nop))
    assemble(p, KB / "ring_nop.hsaco")
    return 0


if __name__ == "__main__":
    sys.exit(main())
