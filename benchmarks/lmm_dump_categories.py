"""FMHIP_PROFILE_DUMP tables of the LMM calibration by kind of launch: valuation chains (one output + fused expectation), four-step and
two-step simulation components, interpreter launches.  usage: lmm_dump_categories.py dump.txt [dump.txt …]"""
import re, sys
from collections import defaultdict


def load(p):
    rows = []
    for ln in open(p):
        m = re.match(r'\s+ops\s+(\d+) in\s+(\d+) out\s+(\d+) red (\d) rows\s+(\d+) n\s+(\d+) (\w+)\s+(\d+) launches\s+([\d.]+) ms\s+([\d.]+) us each\s+(\d+) GB/s', ln)
        if m:
            ops, i, o, red, r, n, tier, l, ms, us, gbs = m.groups()
            rows.append(dict(ops=int(ops), i=int(i), o=int(o), red=int(red), rows=int(r), tier=tier, l=int(l), ms=float(ms), gbs=float(gbs)))
    return rows


def cat(r):
    if r['tier'] == 'interpreter': return 'interpreter'
    if r['o'] <= 1 and r['ops'] >= 9: return 'valuation %s%s%s' % ('>= 40 rows' if r['rows'] >= 40 else '< 40 rows', '' if r['red'] else ' (no expectation)', '' if r['o'] else ', value not stored')
    if r['o'] > 20 and r['ops'] > 0:         # a group of time steps: ≈ 25 operations per component read for four steps, ≈ 13 for two; a state kept inside the group is a second output per component
        steps = 'simulation 4 steps' if r['ops'] / max(1, r['i']) > 20 else 'simulation 2 steps'
        return steps + (' + a kept state' if r['o'] > 1.4 * r['i'] else '') + (' x 8 rows' if r['rows'] == 8 else ' x 1-2 rows')
    return 'other'


for p in sys.argv[1:]:
    c = defaultdict(lambda: [0.0, 0.0, 0])
    for r in load(p):
        k = cat(r); c[k][0] += r['ms']; c[k][1] += r['ms'] * r['gbs']; c[k][2] += r['l']
    tot = sum(v[0] for v in c.values())
    print(f"{p}: {tot:.1f} ms of kernel time, {sum(v[1] for v in c.values()) / tot:.0f} GB/s = {sum(v[1] for v in c.values()) / tot / 8000:.4f} of peak")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1][0]):
        print(f"    {k:42s} {v[0]:8.1f} ms {v[1] / v[0]:7.0f} GB/s {v[2]:5d} launches")
