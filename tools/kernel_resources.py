"""Register / scratch / occupancy table of the PRODUCTION instantiations of a
translation unit (same flags as csrc/Makefile, device code only; development
tool).  usage: kernel_resources.py [shape ...]   (default: 64_1 64_2)
QMC_EXTRA="-D..." in the environment adds flags (A/B variants)."""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = sys.argv[1:] or ['64_1', '64_2']
os.makedirs(os.path.join(R, 'build_tmp'), exist_ok=True)
for s in shapes:
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950',
           '--cuda-device-only', '-S', '-Rpass-analysis=kernel-resource-usage',
           '-Wno-unused-value', '-fno-slp-vectorize'] + \
        os.environ.get('QMC_EXTRA', '').split() + \
        ['-o', os.path.join(R, 'build_tmp', f'inst_{s}.s'),
         os.path.join(R, 'phd_qmclib_amd', 'csrc', f'inst_{s}.hip')]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: (.*?)( \[-Rpass.*)?$', line)
        if not m:
            continue
        t = m.group(1).strip()
        f = re.match(r'Function Name: (\S+)', t)
        if f:
            cur = {'name': f.group(1)}
            rows.append(cur)
            continue
        f = re.match(r'(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|'
                     r'Occupancy \[waves/SIMD\]): (\d+)', t)
        if f and cur is not None:
            cur[f.group(1).split()[0]] = int(f.group(2))
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows),
                           capture_output=True, text=True).stdout.split('\n')
    for r, n in zip(rows, names):
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        print(f"{n:62s} vgpr={r.get('VGPRs')} sgpr={r.get('TotalSGPRs')} "
              f"scratch={r.get('ScratchSize')} waves={r.get('Occupancy')}")
