#!/usr/bin/env python
"""same-box A/B of engine switches: runs `bench.py --train-only --no-cpu-baseline --no-events` once per environment given on the
command line (';'-separated KEY=VAL lists, '-' = defaults), interleaved ROUNDS times, and prints ms/step per configuration.

    python scripts/ab_step.py 2 - IFCBK_WGRAD_GROUP=0 'IFCBK_WGRAD_LANE=1;IFCBK_LANES=3'
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
cfgs = sys.argv[2:]
res = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        env = dict(os.environ)
        if c != '-':
            for kv in c.split(';'):
                k, v = kv.split('=', 1)
                env[k] = v
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--train-only', '--no-cpu-baseline', '--no-events',
                            '--steps', os.environ.get('AB_STEPS', '40'), '--warmup', '10'], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        if p.returncode != 0 or not line:
            print('FAILED', c, p.stderr[-1500:], flush=True)
            res[c].append(float('nan'))
            continue
        ms = json.loads(line[-1])['ms_per_step']
        res[c].append(ms)
        print('round %d  %-60s %.3f ms/step' % (r, c, ms), flush=True)
print('---- summary (ms/step per round)')
for c in cfgs:
    print('%-60s %s' % (c, '  '.join('%.3f' % v for v in res[c])))
