#!/usr/bin/env python3
"""Instruction-mix summary per kernel of a hipcc -S dump (development aid)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = re.compile(r'^(_Z\w+):', re.M)
names = [(m.group(1), m.start()) for m in pat.finditer(s)]
for i, (name, pos) in enumerate(names):
    end = names[i + 1][1] if i + 1 < len(names) else len(s)
    body = s[pos:end].split('.Lfunc_end')[0]
    ins = []
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
            continue
        ins.append(l.split()[0])
    if len(ins) < 50:
        continue
    c = Counter(ins)
    print(name[:80], 'total', len(ins))
    keys = sys.argv[2:] or ['v_mfma_f32_32x32x16_f16', 'ds_read_b128', 'ds_write_b128', 'v_accvgpr_read_b32',
                            'v_accvgpr_write_b32', 'v_sin_f32', 'v_cvt_f16_f32', 'v_cvt_f32_f16', 'v_pack_b32_f16',
                            'v_fma_mix_f32', 's_waitcnt', 's_barrier', 'global_load_dwordx4', 'scratch_load_dword',
                            'scratch_store_dword', 'v_mov_b32', 's_nop']
    print('   ', {k: c.get(k, 0) for k in keys})
    print('    top', c.most_common(22))
