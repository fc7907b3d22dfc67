"""Aggregate rocprofv3 CSV output (kernel trace + counter passes) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
# --last N: average only the last N dispatches of every kernel (skips an
# equilibration phase run by the same process)
LAST = int(sys.argv[sys.argv.index('--last') + 1]) if '--last' in sys.argv else 0


def short(name):
    return name.split('(')[0].replace('void ', '')[:60]


dur = defaultdict(list)
for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
if LAST:
    dur = defaultdict(list, {k: v[-LAST:] for k, v in dur.items()})
print('== kernel durations (trace pass) ==')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:62s} calls={len(v):4d} avg={sum(v)/len(v)/1e3:10.1f} us total={sum(v)/1e6:9.3f} ms')

cnt = defaultdict(lambda: defaultdict(float))
ncall = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    rows = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        rows[short(r['Kernel_Name'])][r['Counter_Name']].append(
            (int(r.get('Dispatch_Id', 0) or 0), float(r['Counter_Value'])))
    for k, per in rows.items():
        for c, vals in per.items():
            vals.sort()
            if LAST:
                vals = vals[-LAST:]
            cnt[k][c] += sum(v for _, v in vals)
            ncall[k][c] += len(vals)
print('== counters, average per dispatch ==')
for k in sorted(cnt, key=lambda k: -cnt[k].get('SQ_WAVE_CYCLES', 0)):
    print(k)
    for c in sorted(cnt[k]):
        print(f'    {c:28s} {cnt[k][c]/max(1,ncall[k][c]):16.1f}   (n={ncall[k][c]})')
