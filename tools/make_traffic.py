"""profiles/traffic.json from the PMC summaries of tools/profile.sh
(development tool).  usage: make_traffic.py N=<summary.txt> kernel-substring ...
Per unit (chain-step / walker-step): HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE)
x 1024 / SQ_WAVES x waves-per-unit -- FETCH_SIZE counts half of a coalesced
read on gfx950 (MI355X_MICROARCH.md, HBM section) -- and VALU / SALU
instructions per unit = SQ_INSTS_* / SQ_WAVES (one wavefront per walker)."""
import json
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith('=='):
            cur = None
            continue
        if not line.startswith(' ') and line.strip():
            cur = line.strip()
            out[cur] = {}
        elif cur is not None:
            m = re.match(r'\s+(\S+)\s+([0-9.eE+-]+)', line)
            if m:
                out[cur][m.group(1)] = float(m.group(2))
    return out


res = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* in '
                 'separate passes (tools/profile.sh), profiles/r03_*_pmc_summary.txt',
       'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-count of '
                     'coalesced reads), write bytes = WRITE_SIZE x 1024'}
for arg in sys.argv[1:]:
    parts = arg.split(',')
    key, path, kern, unit = parts[:4]
    # waves launched per unit of work (DMC launches max_num_walkers waves,
    # the ones beyond the population exit at once)
    scale = float(parts[4]) if len(parts) > 4 else 1.0
    for name, c in parse(path).items():
        if kern in name and 'SQ_WAVES' in c:
            w = c['SQ_WAVES'] / scale
            ent = res.setdefault(key, {})
            short = 'vmc_step_kernel' if 'vmc' in kern else 'dmc_evolve_kernel'
            ent[f'{short}_bytes_per_{unit}'] = \
                (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / w
            ent[f'{short}_valu_instr_per_{unit}'] = c['SQ_INSTS_VALU'] / w
            ent[f'{short}_salu_instr_per_{unit}'] = c['SQ_INSTS_SALU'] / w
            ent[f'{short}_lds_instr_per_{unit}'] = c['SQ_INSTS_LDS'] / w
            break
print(json.dumps(res, indent=1))
