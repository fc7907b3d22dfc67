"""Per-loop instruction census of one kernel's ISA (development tool).
usage: isa_loops.py file.s [kernel-name-substring]
Groups the basic blocks by the innermost loop header the assembler comments
name and counts instructions by class; loops are where the dynamic counts come
from."""
import re
import sys
from collections import Counter, defaultdict

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2] if len(sys.argv) > 2 else 'vmc_step_kernel'
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ':' in l)][0]
end = [i for i, l in enumerate(lines) if 's_endpgm' in l and i > start][-1]
for i in range(start, len(lines)):
    if '.Lfunc_end' in lines[i]:
        end = i
        break
cur = 'straight-line'
stats = defaultdict(Counter)
ops = defaultdict(Counter)
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m or '; %bb.' in l:
        rest = m.group(2) if m else l
        h = re.search(r'Header=(BB\d+_\d+)', rest)
        if h:
            cur = 'loop ' + h.group(1)
        elif 'Loop Header' in rest and m:
            cur = 'loop ' + m.group(1)[2:]
        else:
            cur = 'straight-line'
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_'):
        cls = 'valu'
    elif op.startswith('s_cbranch') or op == 's_branch':
        cls = 'branch'
    elif op.startswith('s_waitcnt') or op == 's_nop':
        cls = 'wait'
    elif op.startswith('s_load'):
        cls = 'smem'
    elif op.startswith('s_'):
        cls = 'salu'
    elif op.startswith('ds_'):
        cls = 'lds'
    else:
        cls = 'vmem'
    stats[cur][cls] += 1
    ops[cur][op] += 1
for k, v in stats.items():
    print(f'{k:18s}', ' '.join(f'{c}={n}' for c, n in sorted(v.items())))
    sal = [(o, n) for o, n in ops[k].most_common() if o.startswith('s_')][:10]
    print('     ', ' '.join(f'{o}:{n}' for o, n in sal))
