#!/usr/bin/env python3
"""Static instruction census of a gfx950 ISA dump (tools/isa_one.sh).

usage: isa_count.py file.s [--sections]

Counts instructions by class for every kernel in the file.  With `; SECTION
<name>` marker comments in the stream (the kernels emit them through
`QMC_SECTION("name")` when built with -DQMC_SECTIONS) the counts are also
broken down per section.  Static counts: a loop body counts once -- multiply
by the trip count yourself (the rotation loop is marked as its own section).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_'):
        if op.startswith(('v_mfma', 'v_smfma')):
            return 'mfma'
        return 'valu'
    if op.startswith('s_'):
        if op.startswith(('s_load', 's_buffer_load', 's_store')):
            return 'smem'
        if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')):
            return 'wait'
        if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc',
                          's_swappc')):
            return 'branch'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    path = sys.argv[1]
    kernel = None
    section = 'top'
    counts = collections.OrderedDict()
    valu_ops = collections.defaultdict(collections.Counter)
    for line in open(path):
        s = line.strip()
        m = re.match(r'^([A-Za-z_][\w$.]*):', s)
        if m and not s.startswith(('.L', 'BB')) and '@function' not in s:
            name = m.group(1)
            if name.startswith('_Z') or name.endswith('kernel'):
                kernel = name
                section = 'top'
            continue
        if kernel is None:
            continue
        m = re.search(r';\s*SECTION\s+(\S+)', s)
        if m:
            section = m.group(1)
            continue
        if s.startswith('.end_amdhsa_kernel') or s.startswith('.section'):
            continue
        if s.startswith(('.', ';', '//')) or not s:
            if s.startswith('.Lfunc_end'):
                kernel = None
            continue
        op = s.split()[0]
        if not re.match(r'^[a-z_0-9]+$', op):
            continue
        c = classify(op)
        counts.setdefault(kernel, collections.OrderedDict()) \
              .setdefault(section, collections.Counter())[c] += 1
        if c == 'valu':
            valu_ops[(kernel, section)][op] += 1
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv \
        else None
    for k, secs in counts.items():
        if only and only not in k:
            continue
        tot = collections.Counter()
        print(f'== {k}')
        for sec, c in secs.items():
            tot.update(c)
            print(f'  {sec:24s} ' + ' '.join(f'{n}={c[n]}' for n in
                  ('valu', 'salu', 'lds', 'vmem', 'smem', 'branch', 'wait')
                  if c[n]))
            if '--ops' in sys.argv:
                top = valu_ops[(k, sec)].most_common(12)
                print('      ' + ' '.join(f'{o}:{n}' for o, n in top))
        print(f'  {"TOTAL (static)":24s} ' + ' '.join(f'{n}={tot[n]}' for n in
              ('valu', 'salu', 'lds', 'vmem', 'smem', 'branch', 'wait')
              if tot[n]))


if __name__ == '__main__':
    main()
