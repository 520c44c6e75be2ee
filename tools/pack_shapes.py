#!/usr/bin/env python3
"""The rolled-loop kernels of the kernel pack by shape, with the registers they were compiled to.
usage: finmath-lib-cuda-extensions_amd/build/jit_pack_tool --names finmath-lib-cuda-extensions_amd/csrc/kernel_pack.txt > names.txt; tools/pack_inspect.py > regs.txt; tools/pack_shapes.py names.txt regs.txt"""
import re, sys
regs = {}
for ln in open(sys.argv[2]):
    f = ln.split()
    if f and f[0].startswith('fm_jit') and f[0].endswith('_t'): regs[f[0][:-2]] = (int(f[1]), int(f[5]))
for ln in open(sys.argv[1]):
    n, l = ln.split(None, 1)
    if not l.startswith('rolled'): continue
    g = int(re.search(r'globals (\d+)', l).group(1)); li = int(re.search(r'inputs (\d+)', l).group(1))
    carried = len(re.search(r'carried(.*?) final', l).group(1).split())
    out = len(re.search(r' out(.*?) body', l).group(1).split())
    parts = l.split(' body ')[1].split(' peel ')
    peel = len(parts) > 1
    what = 'plain'
    if peel:
        pre = len(re.search(r' pre (.*?) post', ' ' + parts[1]).group(1).split()) if ' pre ' in parts[1] else 0
        m = re.search(r' post (.*?)( reduce|$)', parts[1]); post = len(m.group(1).split()) if m else 0
        what = f"peeled: {pre} operations in front, {post} behind{', reduces its root' if ' reduce ' in parts[1] else ''}"
    r = regs.get(n, (0, 0))
    print(f"{n}  {r[0]:3d} VGPRs {r[1]} waves/SIMD   {g} global, {li} per-iteration inputs, {carried} carried, {out} stored per iteration, period {len(parts[0].split())}; {what}")
