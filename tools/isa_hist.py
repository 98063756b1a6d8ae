"""Static instruction histogram of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only).
    python tools/isa_hist.py /tmp/march.s 'k_level_marchILi3ELb1' [per_row_divisor]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
starts = [(i, l) for i, l in enumerate(s) if re.match(r'^_Z.*:\s*; @', l) and pat in l]
ends = [i for i, l in enumerate(s) if 's_endpgm' in l]
for i, l in starts:
    e = min(x for x in ends if x > i)
    body = [x.strip() for x in s[i:e] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    c = Counter(x.split()[0] for x in body)
    cls = Counter()
    for k, v in c.items():
        if k.startswith('v_pk_'): cls['valu pk f32'] += v
        elif k.startswith('v_') and 'f64' in k: cls['valu f64'] += v
        elif k.startswith('v_mov') or k.startswith('v_accvgpr'): cls['valu mov'] += v
        elif k.startswith('v_cndmask'): cls['valu cndmask'] += v
        elif k.startswith('v_'): cls['valu other'] += v
        elif k.startswith('s_cbranch') or k == 's_branch': cls['branch'] += v
        elif k.startswith('s_nop'): cls['s_nop'] += v
        elif k.startswith('s_waitcnt'): cls['s_waitcnt'] += v
        elif k.startswith('s_'): cls['salu'] += v
        elif k.startswith('ds_'): cls['lds'] += v
        elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cls['vmem'] += v
        else: cls['other'] += v
    print(l.split(':')[0][:70], 'instructions', len(body))
    print('  ' + '  '.join(f'{k} {v / div:.1f}' for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
    print('  valu other:', sorted(((k, v) for k, v in c.items() if k.startswith('v_') and not k.startswith(('v_pk_', 'v_mov', 'v_cndmask')) and 'f64' not in k), key=lambda kv: -kv[1])[:12])
