"""profiles/traffic.json from the PMC summaries of tools/profile.sh
(development tool).  usage: make_traffic.py key,summary.txt,kernel-substring,unit[,scale] ...
Per unit (chain-step / walker-step): HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE)
x 1024 / SQ_WAVES x waves-per-unit -- FETCH_SIZE counts half of a coalesced
read on gfx950 (MI355X_MICROARCH.md, HBM section) -- and VALU / SALU
instructions per unit = SQ_INSTS_* / SQ_WAVES (one wavefront per walker).

Every entry records WHICH kernels were measured: `kernel_source_sha` is the
`qmc_source_hash()` of the library the counter passes ran (tools/profile.sh
appends it to the summary) and `head` the commit the file was made at.
bench.py compares the hash with the library it loads and reports
`roofline.traffic: null` when they differ."""
import json
import re
import subprocess
import sys


def parse(path):
    out, cur, sha = {}, None, None
    for line in open(path):
        if line.startswith('kernel_source_sha='):
            sha = line.strip().split('=', 1)[1]
            continue
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
    return out, sha


def main(argv):
    try:
        head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'],
                              capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ''
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* in '
                     'separate passes (tools/profile.sh), profiles/'
                     'r04_*_pmc_summary.txt',
           'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-count '
                         'of coalesced reads), write bytes = WRITE_SIZE x 1024',
           'head': head}
    for arg in argv:
        parts = arg.split(',')
        key, path, kern, unit = parts[:4]
        # waves launched per unit of work (DMC launches max_num_walkers waves,
        # the ones beyond the population exit at once)
        scale = float(parts[4]) if len(parts) > 4 else 1.0
        kernels, sha = parse(path)
        for name, c in kernels.items():
            if kern in name and 'SQ_WAVES' in c:
                w = c['SQ_WAVES'] / scale
                ent = res.setdefault(key, {})
                short = 'vmc_step_kernel' if 'vmc' in kern else 'dmc_evolve_kernel'
                ent[f'{short}_bytes_per_{unit}'] = \
                    (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / w
                ent[f'{short}_read_bytes_per_{unit}'] = \
                    2 * c['FETCH_SIZE'] * 1024 / w
                ent[f'{short}_write_bytes_per_{unit}'] = \
                    c['WRITE_SIZE'] * 1024 / w
                ent[f'{short}_valu_instr_per_{unit}'] = c['SQ_INSTS_VALU'] / w
                ent[f'{short}_salu_instr_per_{unit}'] = c['SQ_INSTS_SALU'] / w
                ent[f'{short}_lds_instr_per_{unit}'] = c['SQ_INSTS_LDS'] / w
                # (one hash per kernel family: the passes of a key may come
                # from different summaries)
                ent[f'{short}_kernel_source_sha'] = sha
                break
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(sys.argv[1:])
