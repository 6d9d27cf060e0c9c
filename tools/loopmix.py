#!/usr/bin/env python3
"""Instruction mix of the innermost loop around the N-th cluster of an instruction pattern in one kernel of a hipcc -S dump.
   tools/loopmix.py file.s <kernel-name-substring> <pattern e.g. v_mfma_f32_16x16x4> [cluster index]"""
import re, sys, collections
src, kname, pat = sys.argv[1], sys.argv[2], sys.argv[3]
which = int(sys.argv[4]) if len(sys.argv) > 4 else 0
text = open(src).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and kname in l and l.rstrip().endswith(('EEvPKfS2_S2_S2_S2_PfS3_S3_S3_iiiiiiiiii', ':')) or (l.startswith('_Z') and kname in l and ':' in l))
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
lines = text[start:end + 1]
idx = [i for i, l in enumerate(lines) if pat in l]
cl, cur = [], [idx[0]]
for i in idx[1:]:
    if i - cur[-1] > 400: cl.append(cur); cur = [i]
    else: cur.append(i)
cl.append(cur)
print('kernel lines', len(lines), 'clusters', [(c[0], c[-1], len(c)) for c in cl])
tgt = cl[which][0]
labpos = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labpos[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labpos and labpos[t] < i: loops.append((labpos[t], i))
cont = [(a, b) for a, b in loops if a <= tgt <= b]
a, b = min(cont, key=lambda x: x[1] - x[0]) if cont else (max(0, tgt - 200), min(len(lines) - 1, cl[which][-1] + 1500))
cnt = collections.Counter()
for l in lines[a:b + 1]:
    l = l.strip()
    if not l or l[0] in ';.': continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, c in cnt.items():
    g = ('mfma' if op.startswith('v_mfma') else 'v_exp' if op.startswith('v_exp') else 'lds' if op.startswith('ds_') else
         'valu' if op.startswith('v_') else 'waitcnt' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_') else
         'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'other')
    groups[g] += c
print('loop lines', a, b, 'instructions', sum(cnt.values()), dict(groups))
print(cnt.most_common(30))
