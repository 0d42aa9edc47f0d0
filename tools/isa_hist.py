#!/usr/bin/env python
"""Per-opcode histogram of the COMMON path through one kernel's main loop, from hipcc's assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o advect.s lagrangiancoherence_amd/csrc/advect.hip
    python tools/isa_hist.py advect.s 'advect_lds_kernelILi1ELi4ELb1E' [--json out.json]

The walk starts at the kernel's outermost hot loop header (--loop LABEL, default: the depth-1 loop with the
most LDS reads, else the longest) and
follows the path a wave takes when no lane needs a rare-case redo: `s_cbranch_execz` is taken (the
exec-masked block is skipped), `s_cbranch_execnz` falls through, `s_branch` is followed, and a uniform
branch (`vccz/vccnz/scc0/scc1`) takes the side given with --take LABEL (default: fall through; every such
decision is listed).  The walk ends at the back edge.  Classes follow the SQ counters: valu (SQ_INSTS_VALU),
salu + branch (SQ_INSTS_SALU + SQ_INSTS_BRANCH), lds, vmem, smem, and misc (s_waitcnt / s_nop / barriers).
"""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_code_end")):
        return "misc"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        if not on and re.match(r"^_Z\w*:", ln) and key in ln:
            on = True
            continue
        if on:
            if ln.startswith(".Lfunc_end"):   # (a kernel may hold several s_endpgm: early exits)
                break
            out.append(ln.rstrip("\n"))
    if not out:
        raise SystemExit(f"kernel matching {key!r} not found in {path}")
    return out


def main():
    args = sys.argv[1:]
    take = set()
    out_json = None
    while "--take" in args:
        i = args.index("--take")
        take.add(args[i + 1])
        del args[i:i + 2]
    loop = None
    if "--loop" in args:
        i = args.index("--loop")
        loop = args[i + 1]
        del args[i:i + 2]
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    path, key = args[0], args[1]
    lines = kernel_lines(path, key)
    label_at = {}
    insts = []   # (index in lines, opcode, operand text)
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(?:\s*;.*)?$", ln)
        if m and not m.group(1).startswith("."):
            insts.append((i, m.group(1), m.group(2)))
    # depth-1 loop headers; the hot loop is the one given, else the longest (header .. next depth-1 header)
    headers = [m.group(1) for ln in lines for m in [re.match(r"^(\.LBB\w+):.*This Loop Header: Depth=1", ln)] if m]
    if not headers:
        raise SystemExit("no depth-1 loop found")
    if loop is None:   # the loop with the most LDS reads (the tile kernels' hot loop), else the longest
        pos = sorted(label_at[h] for h in headers) + [len(insts)]
        ext = {h: (label_at[h], pos[pos.index(label_at[h]) + 1]) for h in headers}
        nds = {h: sum(1 for k in range(*ext[h]) if insts[k][1].startswith("ds_read")) for h in headers}
        loop = max(headers, key=lambda h: (nds[h], ext[h][1] - ext[h][0]))
    header, start = loop, label_at[loop]
    pc = start
    hist = collections.Counter()
    cls = collections.Counter()
    decisions = []
    steps = 0
    closing = False   # a back edge may land on a latch block just above the header: walk it down to the header
    while steps < 100000:
        steps += 1
        if closing and pc == start:
            break
        _, op, arg = insts[pc]
        hist[op] += 1
        cls[classify(op)] += 1
        tgt = arg.strip()
        jump = None
        if op == "s_branch":
            jump = tgt
        elif op.startswith("s_cbranch"):
            if op.endswith("execz"):
                jump = tgt
            elif op.endswith("execnz"):
                jump = None
            else:
                taken = tgt in take or (label_at[tgt] <= start and not closing and tgt not in take and f"!{tgt}" not in take
                                        and pc > start and label_at[tgt] <= start)
                decisions.append(f"{op} {tgt}: {'taken' if taken else 'fall through'}")
                jump = tgt if taken else None
        elif op == "s_endpgm":
            break
        if jump is not None:
            if label_at[jump] <= start:
                if label_at[jump] == start:
                    break
                closing = True
            pc = label_at[jump]
        else:
            pc += 1
    total = sum(cls.values())
    res = {"kernel": key, "loop_header": header, "instructions_on_common_path": total, "classes": dict(cls),
           "uniform_branch_decisions": decisions, "opcodes": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
    print(f"{key}: loop {header}, {total} instructions on the common path")
    print("  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
    for d in decisions:
        print("  uniform:", d)
    for op, n in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"  {n:4d}  {op}")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
