#!/usr/bin/env python3
"""Issue-slot budget between consecutive MFMAs of one kernel in a hipcc -S dump (development aid).

usage: mfma_gaps.py file.s <mangled-name-substring> [first_mfma last_mfma]
Every instruction of a lone wave costs one 4-cycle issue slot; quarter-rate transcendentals 4 slots, ds_read_b128 2.  An
fp16 32x32x16 MFMA keeps the matrix pipe busy for 8 slots (itself included), the scaled fp8 32x32x64 one for 21."""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = [x for x in re.finditer(r'^(_Z\w+):', s, re.M) if key in x.group(1)][0]
body = s[m.end():].split('.Lfunc_end')[0]
ins = []
for l in body.split('\n'):
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    ins.append(l)


def cost(op):
    if op in ('v_sin_f32', 'v_cos_f32', 'v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32'):
        return 4
    if op.startswith('ds_read_b128') or op.startswith('ds_read_b64_tr'):
        return 2
    return 1


rows = []
cur = None
for l in ins:
    op = l.split()[0]
    if op.startswith('v_mfma'):
        if cur:
            rows.append(cur)
        cur = {'mfma': 'f8' if 'f8f6f4' in op else 'h', 'n': 0, 'slots': 0, 'ops': {}}
    elif cur is not None:
        cur['n'] += 1
        cur['slots'] += cost(op)
        k = op if op.startswith(('s_waitcnt', 's_barrier', 'ds_', 'buffer', 'global', 'v_sin', 'v_exp', 's_nop')) else None
        if k:
            cur['ops'][k] = cur['ops'].get(k, 0) + 1
if cur:
    rows.append(cur)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(rows)
tot_b = tot_s = over = 0
for i, r in enumerate(rows[lo:hi], lo):
    budget = 20 if r['mfma'] == 'f8' else 7
    tot_b += budget + 1
    tot_s += r['slots'] + 1
    over += max(0, r['slots'] - budget)
    print(f"{i:5d} {r['mfma']:>2} n={r['n']:3d} slots={r['slots']:3d} budget={budget:2d} {'OVER' if r['slots'] > budget else '':4} {r['ops']}")
print(f'mfmas {hi - lo}: matrix slots {tot_b}, issue slots {tot_s}, overflow beyond per-MFMA budgets {over}')
