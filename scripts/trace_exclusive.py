"""dev tool: where a multi-lane step's wall time goes.  Input: a rocprofv3 --kernel-trace CSV of `bench.py --train-only ...`.
For the last steady-state steps it splits the time axis at every kernel start / end and charges each slice to the kernels running
in it: `exclusive` = slices where the kernel runs alone (shortening it shortens the step one for one), `shared` = its share of
slices it runs beside others (1/n each), `idle` = slices with nothing running.
    python scripts/trace_exclusive.py <kernel_trace.csv> [steps_to_analyse]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# a step ends with its LAST Adam launch (the optimizer runs in IFCBK_OPT_BUCKETS launches per step): the one that ends last
# among those started before the next step's first forward kernel (the stem conv)
adam_all = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
stems = [r[0] for r in rows if 'stem_u8_fwd' in r[2]]
if stems:
    adam = []
    for a, b in zip(stems, stems[1:] + [1 << 62]):
        mine = [i for i in adam_all if a <= rows[i][0] < b]
        if mine:
            adam.append(max(mine, key=lambda i: rows[i][1]))
    adam = sorted(set(adam))
else:
    adam = adam_all
if len(adam) < nsteps + 1:
    raise SystemExit('not enough steps in the trace')
lo, hi = adam[-nsteps - 1] + 1, adam[-1] + 1
sel = rows[lo:hi]
t0, t1 = rows[adam[-nsteps - 1]][1], max(r[1] for r in sel)

def short(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*$', '', n)
    return n.replace('unsigned short', 'bf16')[:44]


ev = []
for s, e, n in sel:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
excl, shared, tot = defaultdict(float), defaultdict(float), defaultdict(float)
active = defaultdict(int)
idle = 0.0
prev = t0
for t, d, n in ev:
    dt = t - prev
    if dt > 0:
        cur = [k for k, v in active.items() if v > 0]
        if not cur:
            idle += dt
        elif len(cur) == 1 and active[cur[0]] == 1:
            excl[cur[0]] += dt
        else:
            w = sum(active[k] for k in cur)
            for k in cur:
                shared[k] += dt * active[k] / w
    active[n] += d
    prev = t
for s, e, n in sel:
    tot[n] += e - s
wall = (t1 - t0) / nsteps / 1e6
print('step wall %.3f ms | idle %.3f ms | exclusive %.3f ms | shared %.3f ms   (per step, %d steps)'
      % (wall, idle / nsteps / 1e6, sum(excl.values()) / nsteps / 1e6, sum(shared.values()) / nsteps / 1e6, nsteps))
agg = defaultdict(lambda: [0.0, 0.0, 0.0])
for n in tot:
    a = agg[short(n)]
    a[0] += excl[n]; a[1] += shared[n]; a[2] += tot[n]
print('%-46s %9s %9s %9s' % ('kernel', 'excl ms', 'shared ms', 'sum ms'))
for k, (a, b, c) in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:40]:
    print('%-46s %9.3f %9.3f %9.3f' % (k, a / nsteps / 1e6, b / nsteps / 1e6, c / nsteps / 1e6))
# idle gaps: which kernel ended last before the gap, which one started after it
gaps = defaultdict(lambda: [0, 0.0])
active2, last_end, prev2 = 0, None, t0
for t, d, n in ev:
    if active2 == 0 and d == 1 and last_end is not None and t > prev2:
        g = gaps[(short(last_end), short(n))]
        g[0] += 1
        g[1] += t - prev2
    active2 += d
    if d == -1:
        last_end = n
    prev2 = t
print('\nidle gaps by (kernel that ended, kernel that started): count per step, ms per step')
for (a, b), (cnt, ns) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-40s -> %-40s %6.1f %8.3f' % (a[:40], b[:40], cnt / nsteps, ns / nsteps / 1e6))
