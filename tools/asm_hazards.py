#!/usr/bin/env python3
"""Audit of the hazards hipcc cannot see around inline-asm instructions (gfx950).

hipcc's hazard recogniser inserts the wait states gfx940/gfx950 need between ordinary instructions, but an
inline-asm block is opaque to it: it does not know that the block holds a VALU instruction, so the rules that key on
"the CONSUMER is a VALU instruction" are not applied when the consumer sits inside asm.  Probed with the ROCm 7.2
compiler (tools/asm_hazard_probe.hip, the listing is in profiles/r05/asm_hazard_probe.txt):

  rule                                                        wait states   ordinary consumer   asm consumer
  VALU writes SGPR (v_readlane, v_cmp ...) -> VALU reads it        2          s_nop inserted      NOT inserted
  trans op (v_rcp, v_sqrt, ...) writes VGPR -> VALU reads it       1          s_nop inserted      NOT inserted
  VALU writes VGPR -> v_readlane / v_readfirstlane reads it        1          s_nop inserted      inserted (asm producer too)
  VALU writes VGPR -> DPP instruction reads it                     2          s_nop inserted      (round 6: the asm carries its own s_nop)
  VALU writes EXEC (v_cmpx ...) -> DPP instruction                 5          s_nop inserted      (round 6: the asm carries its own s_nop)

This script walks the compiler's assembly (hipcc -S) kernel by kernel, finds every instruction between ;;#ASMSTART and
;;#ASMEND, and reports each one whose source registers were written inside the hazard window by a producer of the
kinds above.  It scans backwards from each asm instruction through the hazard window; at a label it follows every predecessor
(the fall-through and each branch that targets the label; the branch itself counts as one wait state).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -Iinclude -o advect.s csrc/advect.hip
    python tools/asm_hazards.py advect.s [--kernel REGEX] [--json out.json]

Exit code 1 when a hazard is found.
"""
from __future__ import annotations

import json
import re
import sys

TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy|_clamp)?_(f16|f32|f64|bf16)")
REG = re.compile(r"\b([sv])(\d+)\b|\b([sv])\[(\d+):(\d+)\]")
VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32")


def regs(text: str):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    if re.search(r"\bvcc\b", text):
        out.add(("s", 106))
        out.add(("s", 107))
    return out


def split_operands(ins: str):
    parts = ins.split(None, 1)
    if len(parts) == 1:
        return parts[0], []
    return parts[0], [p.strip() for p in parts[1].split(",")]


def dest_regs(op: str, operands):
    """Registers written by an instruction (first operand; VOP3 compares with an SGPR pair; vcc for e32 compares)."""
    if not operands or op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "ds_write", "global_store", "buffer_store", "flat_store", "s_endpgm", "s_barrier")):
        return set()
    d = regs(operands[0])
    if op.startswith("v_cmp") and op.endswith("_e32"):
        d = {("s", 106), ("s", 107)}
    return d


def wait_states(op: str, operands) -> int:
    if op == "s_nop":
        return int(operands[0], 0) + 1
    return 1


def audit(path: str, kernel_re: str | None = None):
    findings, sites = [], 0
    kern = None
    lines = open(path, encoding="utf-8", errors="replace").read().splitlines()
    body = []          # (lineno, kind, op, operands, in_asm) of the current kernel; kind: "ins" | "label"
    in_asm = False

    def flush():
        nonlocal sites
        if kern is None or (kernel_re and not re.search(kernel_re, kern)):
            return
        branches_to = {}
        for i, (ln, kind, op, operands, asm) in enumerate(body):
            if kind == "ins" and op.startswith(("s_cbranch", "s_branch")) and operands:
                branches_to.setdefault(operands[0], []).append(i)
        for i, (ln, kind, op, operands, asm) in enumerate(body):
            if kind != "ins" or not asm or not op.startswith("v_"):
                continue
            sites += 1
            srcs = set()
            for o in operands[1:]:
                srcs |= regs(o)
            # asm instructions that read and write one register ("+v"): the first operand is a source too when repeated
            need_s = {r for r in srcs if r[0] == "s"}
            need_v = {r for r in srcs if r[0] == "v"}
            seen = set()
            dpp = "_dpp" in op or any(re.search(r"\b(row_|quad_perm|wave_|bank_mask|row_mask)", o) for o in operands)
            limit = 5 if dpp else 2

            def walk(j, ws):
                """Backwards from body[j] with ``ws`` wait states already between it and the asm instruction; follows
                every predecessor at a label (the fall-through and each branch that targets it)."""
                while j >= 0 and ws < limit:
                    if (j, ws) in seen:
                        return
                    seen.add((j, ws))
                    pln, pkind, pop, poperands, pasm = body[j]
                    if pkind == "label":
                        for b in branches_to.get(pop, ()):       # the branch instruction itself is one wait state
                            walk(b - 1, ws + 1)
                        if j > 0 and body[j - 1][1] == "ins" and body[j - 1][2] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                            return                               # no fall-through into this label
                        j -= 1
                        continue
                    d = dest_regs(pop, poperands)
                    if dpp and ws < 2 and pop.startswith("v_") and (d & need_v):
                        findings.append({"kernel": kern, "line": ln, "asm": f"{op} {', '.join(operands)}", "kind": "VALU-written VGPR read by asm DPP (needs 2)",
                                         "producer": f"{pop} {', '.join(poperands)}", "producer_line": pln, "wait_states_seen": ws})
                    if dpp and pop.startswith("v_cmpx"):
                        findings.append({"kernel": kern, "line": ln, "asm": f"{op} {', '.join(operands)}", "kind": "VALU-written EXEC before asm DPP (needs 5)",
                                         "producer": f"{pop} {', '.join(poperands)}", "producer_line": pln, "wait_states_seen": ws})
                    if ws < 2 and pop.startswith("v_") and (d & need_s):
                        findings.append({"kernel": kern, "line": ln, "asm": f"{op} {', '.join(operands)}", "kind": "VALU-written SGPR read by asm VALU (needs 2)",
                                         "producer": f"{pop} {', '.join(poperands)}", "producer_line": pln, "wait_states_seen": ws})
                    if ws < 1 and TRANS.match(pop) and (d & need_v):
                        findings.append({"kernel": kern, "line": ln, "asm": f"{op} {', '.join(operands)}", "kind": "trans result read by asm VALU (needs 1)",
                                         "producer": f"{pop} {', '.join(poperands)}", "producer_line": pln, "wait_states_seen": ws})
                    ws += wait_states(pop, poperands)
                    j -= 1
            walk(i - 1, 0)
            # asm producer -> v_readlane / v_readfirstlane consumer (the compiler handles this; verified here all the same)
            d = regs(operands[0]) if operands else set()
            if i + 1 < len(body):
                nln, nkind, nop, noperands, nasm = body[i + 1]
                if nkind == "ins" and nop in VALU_SGPR_WRITERS and len(noperands) > 1 and (regs(noperands[1]) & d):
                    findings.append({"kernel": kern, "line": nln, "asm": f"{op} {', '.join(operands)}", "kind": "asm VALU result read by readlane next (needs 1)",
                                     "consumer": f"{nop} {', '.join(noperands)}"})

    for n, raw in enumerate(lines, 1):
        s = raw.strip()
        m = re.match(r"^([A-Za-z_.][\w$.]*):", raw)
        if m and not raw.startswith("\t"):
            name = m.group(1)
            if name.startswith(".L"):
                body.append((n, "label", name, [], False))
            elif name.startswith("_Z") or not name.startswith("."):
                flush()
                kern, body, in_asm = name, [], False
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        op, operands = split_operands(s)
        body.append((n, "ins", op, operands, in_asm))
    flush()
    return sites, findings


if __name__ == "__main__":
    argv = sys.argv[1:]
    if not argv:
        sys.exit(__doc__)
    kre = argv[argv.index("--kernel") + 1] if "--kernel" in argv else None
    sites, findings = audit(argv[0], kre)
    hard = findings
    print(f"{sites} asm VALU sites audited; {len(hard)} hazards")
    for f in findings:
        print(json.dumps(f))
    if "--json" in argv:
        json.dump({"sites": sites, "findings": findings}, open(argv[argv.index("--json") + 1], "w"), indent=1)
    sys.exit(1 if hard else 0)
