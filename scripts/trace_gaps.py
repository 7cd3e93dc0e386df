"""dev tool: per-queue gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (a lane's dependent kernels: how long the
queue sits between the end of one kernel and the start of the next).
    python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] > t_end - 100_000_000]          # the last 100 ms
for key_i, key_n in ((3, 'queue'), (4, 'stream')):
    by = defaultdict(list)
    for r in rows:
        by[r[key_i]].append(r)
    print('---- by', key_n)
    for k, v in sorted(by.items()):
        gaps = sorted(max(0, b[0] - a[1]) for a, b in zip(v, v[1:]))
        if not gaps:
            continue
        busy = sum(r[1] - r[0] for r in v)
        print('%s %-6s kernels %5d  busy %7.2f ms  gap median %6.2f us  p25 %6.2f  p75 %6.2f  p90 %6.2f  sum of gaps < 30 us: %7.2f ms'
              % (key_n, k, len(v), busy / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[len(gaps) // 4] / 1e3, gaps[3 * len(gaps) // 4] / 1e3,
                 gaps[9 * len(gaps) // 10] / 1e3, sum(g for g in gaps if g < 30000) / 1e6))
# gap after a specific small kernel
for pat in ('bn_finalize', 'bn_bwd_finalize', 'bn_apply_kernel', 'conv_slab'):
    g_before, g_after, dur = [], [], []
    by = defaultdict(list)
    for r in rows:
        by[r[3]].append(r)
    for v in by.values():
        for i in range(1, len(v) - 1):
            if pat in v[i][2]:
                g_before.append(max(0, v[i][0] - v[i - 1][1])); g_after.append(max(0, v[i + 1][0] - v[i][1])); dur.append(v[i][1] - v[i][0])
    if dur:
        med = lambda x: sorted(x)[len(x) // 2] / 1e3
        print('%-18s n %4d  duration median %6.2f us  gap before %6.2f us  gap after %6.2f us' % (pat, len(dur), med(dur), med(g_before), med(g_after)))
