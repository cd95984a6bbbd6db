#!/usr/bin/env python3
"""Static instruction budget of one kernel of csrc/pdp_solve.hip, per phase of the sweep and per basic block.

    tools/isa_budget.py [--kernel MANGLED] [--blocks] [--fast] [--out FILE]

Compiles pdp_solve.hip for gfx950 with -gline-tables-only (same code as the shipped object: line tables do not
change instruction selection or scheduling), walks the kernel's assembly, attributes every instruction to the
phase whose source lines (the `// ---- E1` ... markers inside k_sp_solve_lds) the last `.loc` of the kernel body
named, and classifies it:

    fp    v_{add,sub,mul,fma,fmac,pk_*}_f32               the arithmetic the reference asks for
    fpx   other float VALU: max/min, ldexp, frexp, cvt, rcp, div_*, cmp_*_f32, cndmask, trans
    int   integer / bit / move VALU (index unpack, address arithmetic, selects on integers)
    xl    cross-lane VALU (dpp moves, readlane / writelane, permlane)
    salu  scalar ALU (loop control, exec masks, address bases)
    br    s_cbranch / s_branch
    lds   ds_* (count of wave instructions, not bytes)
    vmem  global / buffer / scratch / flat
    wait  s_waitcnt, s_nop, s_barrier

The per-block listing (--blocks) gives the loop bodies: the table committed under profiles/ multiplies those by
the trip counts of the headline instance (n = 200, m = 840, e = 2 520, 512 threads).
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pdp-solver_amd", "csrc", "pdp_solve.hip")
HEADLINE = "_Z14k_sp_solve_ldsILb0ELb0ELb0ELb1EEv5PView11SolveParams"      # <false, false, false, true>: pass 1 through the dispatch list (round 5)

FP = re.compile(r"^v_(pk_)?(add|sub|subrev|mul|fma|fmac|mac|mad)_(f32|legacy_f32)")
XL = re.compile(r"^v_(readlane|writelane|readfirstlane|permlane|mov_b32_dpp|bpermute)|dpp|row_|quad_perm")
FPX = re.compile(r"^v_(max|min|ldexp|frexp|cvt|rcp|rsq|sqrt|exp|log|div|cmp\w*_f32|cmpx\w*_f32|cndmask|trunc|rndne|floor|ceil|fract|med3_f32|max3_f32|min3_f32|pk_max|pk_min)")


def classify(op, rest):
    if op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "br"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm") or op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        if XL.search(op) or "dpp" in rest or "row_" in rest or "quad_perm" in rest:
            return "xl"
        if FP.match(op):
            return "fp"
        if FPX.match(op):
            return "fpx"
        return "int"
    return "other"


def phase_table(src_lines):
    """source line -> phase name, from the markers inside k_sp_solve_lds"""
    start = next(i for i, l in enumerate(src_lines, 1) if "__global__" in l and "k_sp_solve_lds(" in l)
    end = next(i for i, l in enumerate(src_lines, 1) if i > start and l.startswith("}"))
    marks = [(start, "load")]
    names = [("---- E1", "E1"), ("---- R1", "R1"), ("---- E2", "E2"), ("---- P4", "P4"), ("---- P5:", "P5"), ("---- P5b", "P5b"),
             ("PROF_MARK(5)", "gate"), ("---- P6", "P6"), ("---- P7", "P7"), ("---- P8", "P8"), ("---- leave", "leave")]
    for i in range(start, end + 1):
        l = src_lines[i - 1]
        for key, nm in names:
            if key in l and not any(m[1] == nm for m in marks):
                marks.append((i, nm))
    marks.sort()
    tab = {}
    for k, (ln, nm) in enumerate(marks):
        hi = marks[k + 1][0] if k + 1 < len(marks) else end + 1
        for i in range(ln, hi):
            tab[i] = nm
    return tab, start, end


def compile_asm(fast):
    tmp = tempfile.mkdtemp(prefix="isa_budget_")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Wno-unused-function", "-Wno-unused-result", "-Wno-pass-failed", "-gline-tables-only", "--save-temps=obj", "--cuda-device-only",
           "-c", SRC, "-o", os.path.join(tmp, "pdp_solve.o")]
    if fast:
        cmd.insert(1, "-DPDP_FAST_MATH")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(tmp, "pdp_solve-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=HEADLINE)
    ap.add_argument("--asm", default=None, help="an existing .s with line tables (skips the compile)")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--dump", default=None, help="write the annotated kernel listing here")
    a = ap.parse_args()

    src_lines = open(SRC).read().split("\n")
    ptab, k_start, k_end = phase_table(src_lines)
    asm = a.asm or compile_asm(a.fast)
    lines = open(asm).read().split("\n")
    # the main source file's number in this translation unit
    main_file = None
    for l in lines:
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"pdp_solve\.hip\"", l) or re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*pdp_solve\.hip\"", l)
        if m:
            main_file = int(m.group(1))
            break
    begin = next(i for i, l in enumerate(lines) if l.startswith(a.kernel + ":"))
    phase = "load"
    block = "entry"
    per_phase = collections.defaultdict(collections.Counter)
    per_block = collections.OrderedDict()
    dump = []
    for l in lines[begin + 1:]:
        s = l.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            f, ln = int(m.group(1)), int(m.group(2))
            if f == main_file and ln in ptab:
                phase = ptab[ln]
            continue
        m = re.match(r"(\.LBB[\w]+):", s)
        if m:
            block = m.group(1)
            continue
        if not s or s.startswith((".", ";", "//")):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        if not re.match(r"^[sv]_|^ds_|^global_|^buffer_|^scratch_|^flat_", op):
            continue
        c = classify(op, rest)
        per_phase[phase][c] += 1
        key = (block, phase)
        per_block.setdefault(key, collections.Counter())[c] += 1
        dump.append("%-6s %-5s %-12s %s" % (phase, c, block, s))
    cols = ["fp", "fpx", "int", "xl", "salu", "br", "lds", "vmem", "wait"]
    print("kernel %s%s" % (a.kernel, " (PDP_FAST_MATH)" if a.fast else ""))
    print("static instruction counts per phase (whole kernel text, cold paths included)")
    print("%-6s " % "phase" + " ".join("%6s" % c for c in cols) + "   total")
    order = ["load", "E1", "R1", "E2", "P4", "P5", "P5b", "gate", "P6", "P7", "P8", "leave"]
    tot = collections.Counter()
    for ph in order:
        if ph not in per_phase:
            continue
        c = per_phase[ph]
        tot.update(c)
        print("%-6s " % ph + " ".join("%6d" % c[k] for k in cols) + "  %6d" % sum(c.values()))
    print("%-6s " % "all" + " ".join("%6d" % tot[k] for k in cols) + "  %6d" % sum(tot.values()))
    if a.blocks:
        print("\nper basic block (>= 12 instructions)")
        print("%-14s %-6s " % ("block", "phase") + " ".join("%5s" % c for c in cols) + "  total")
        for (blk, ph), c in per_block.items():
            n = sum(c.values())
            if n >= 12:
                print("%-14s %-6s " % (blk, ph) + " ".join("%5d" % c[k] for k in cols) + "  %5d" % n)
    if a.dump:
        with open(a.dump, "w") as f:
            f.write("\n".join(dump) + "\n")


if __name__ == "__main__":
    sys.exit(main())
