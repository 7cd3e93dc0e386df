"""Parse the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as MI355X_MICROARCH.md §HBM prescribes)
into per-kernel average HBM traffic per launch.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events
  python scripts/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_traffic.json

Units/corrections: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane)
coalesced reads -- every read of these kernels is a 16-byte-per-lane load -- so the read side is doubled; WRITE_SIZE is exact
for 16-byte-per-lane stores.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict



def build_id():
    """which build of libifcbk.so these counters belong to: bench.py attaches them to a run only when the hash matches"""
    import hashlib
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get('IFCBK_LIB') or os.path.join(root, 'ifcb_classifier_amd', 'libifcbk.so')
    sha = hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16] if os.path.exists(lib) else None
    try:
        head = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        head = None
    return dict(lib_sha16=sha, git_head=head)


def load(d, name):
    f = glob.glob('%s/*/*counter_collection.csv' % d)[0]
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '').strip()
            agg[k][0] += float(r['Counter_Value'])
            agg[k][1] += 1
    return agg


def main():
    fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k, (v, n) in fe.items():
        w, wn = wr.get(k, [0.0, 1])
        rd = 2.0 * v / n * 1024.0
        wb = w / max(1, wn) * 1024.0
        out[k] = dict(launches=n, read_bytes_per_launch=rd, write_bytes_per_launch=wb, hbm_bytes_per_launch=rd + wb)
    json.dump(dict(method='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 '
                          '--warmup 1 --no-cpu-baseline --no-events`; KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction)',
                   kernels=out, **build_id()), open(sys.argv[3], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print('%-50s %6d launches  %8.1f MB/launch' % (k[:50], v['launches'], v['hbm_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
