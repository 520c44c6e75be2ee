#!/usr/bin/env python3
"""Instruction histogram of the innermost loop(s) of a kernel from `tools/pack_inspect.py --dump NAME` (llvm-objdump -d with
--symbolize-operands): every backward branch closes a loop; for each loop the instructions between its target label and the branch
are counted by class (VALU packed / VALU other / transcendental / VMEM load / VMEM store / LDS / SALU / waitcnt / branch).
usage: tools/pack_inspect.py --dump fm_jit_<hash> | tools/loop_histogram.py [kernel suffix, default _t]"""
import collections, re, sys


def classify(op):
    if op.startswith("v_pk_"): return "valu_packed"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "valu_transcendental"
    if op.startswith("v_") and "f64" in op: return "valu_f64"
    if op.startswith(("v_cmp", "v_cndmask")): return "valu_compare_select"
    if op.startswith(("v_max3_u32", "v_lshl_add_u32", "v_max_u32")): return "valu_range_key"
    if op.startswith(("v_mov", "v_accvgpr")): return "valu_move"
    if op.startswith("v_"): return "valu_other"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store")): return "vmem_store"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"


def hot_path(insts, label_at, start):
    """The instructions executed from label `start` until control returns to it, under the rules that hold for the generated kernels:
    s_branch taken; s_cbranch_vccz taken and s_cbranch_vccnz not taken (the wave-uniform tests `any lane outside the fast range?` of
    ueval_div_all / sqrt_all / log_all: false for Monte-Carlo data); s_cbranch_scc* not taken (last iteration / loop exit)."""
    i, path, seen = label_at[start] + 1, [], 0
    while seen < 100000:
        seen += 1
        lab, op, args = insts[i]
        if lab:
            if lab == start: break
            i += 1; continue
        path.append((op, args))
        m = re.search(r"\b(L\d+)\b", args or "")
        if op == "s_branch" or op == "s_cbranch_vccz": i = label_at[m.group(1)]; continue
        i += 1
        if i >= len(insts): break
    return path


def main():
    suffix = "_t"
    trace = None
    for a in sys.argv[1:]:
        if a.startswith("--trace="): trace = a.split("=", 1)[1]
        elif not a.startswith("--"): suffix = a
    lines = sys.stdin.read().splitlines()
    kernel, body = None, []
    insts = []                                      # (label or None, opcode, operands)
    for ln in lines:
        m = re.match(r"^(?:[0-9a-f]+ )?<(\S+)>:", ln.strip())
        if m and not re.fullmatch(r"L\d+", m.group(1)):
            kernel = m.group(1)
            continue
        if kernel is None or not kernel.endswith(suffix): continue
        if m:
            insts.append((m.group(1), None, None)); continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", ln)
        if m: insts.append((None, m.group(1), m.group(2)))
    label_at = {lab: i for i, (lab, _, _) in enumerate(insts) if lab}
    loops = []
    for i, (lab, op, args) in enumerate(insts):
        if op and op.startswith("s_cbranch"):
            m = re.search(r"\b(L\d+)\b", args or "")
            if m and m.group(1) in label_at and label_at[m.group(1)] < i: loops.append((label_at[m.group(1)], i))
    if trace:
        path = hot_path(insts, label_at, trace)
        c = collections.Counter(classify(op) for op, _ in path)
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print(f"hot path from {trace} back to {trace}: {len(path)} instructions, {valu} VALU")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(f"    {k:22s} {v:5d}")
        ops = collections.Counter(op for op, _ in path if op.startswith("v_"))
        print("    VALU opcodes: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(40)))
        print("    s_waitcnt: " + "; ".join(args for op, args in path if op == "s_waitcnt"))
        if "--print" in sys.argv:
            for op, args in path: print("        " + op + " " + (args or ""))
        return
    total = collections.Counter(classify(op) for _, op, _ in insts if op)
    print(f"kernel *{suffix}: {sum(total.values())} instructions in all")
    for (a, b) in loops:
        inner = [(x, y) for (x, y) in loops if a <= x and y <= b and (x, y) != (a, b)]
        c = collections.Counter(classify(op) for _, op, _ in insts[a:b + 1] if op)
        n = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print(f"loop {insts[a][0]} … instruction {b}: {n} instructions, {valu} VALU{' (contains %d inner loops)' % len(inner) if inner else ''}")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(f"    {k:22s} {v:5d}")
        ops = collections.Counter(op for _, op, _ in insts[a:b + 1] if op and op.startswith("v_"))
        print("    most frequent VALU opcodes: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(12)))
        waits = [args for _, op, args in insts[a:b + 1] if op == "s_waitcnt"]
        print("    s_waitcnt: " + "; ".join(waits))


if __name__ == "__main__":
    main()
