"""Parse one rocprofv3 PMC pass into per-kernel MFMA-pipe utilisation.

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv \
      -d gpurun_out/pmc_mfma -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --train-only
  python scripts/collect_mfma.py gpurun_out/pmc_mfma profiles/r1_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts the cycles an MFMA occupies a SIMD's matrix pipe, summed over all SIMDs (16 per
v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 is the
kernel's duration in shader cycles; 1,024 SIMDs (256 CUs x 4).  utilisation = busy / (cycles x 1024): the fraction of the
matrix pipes' cycles spent multiplying AT THE CLOCK THE KERNEL RAN AT (the chip lowers its clock under MFMA load, so the
fraction of the 2.4 GHz nominal peak is lower by clock / 2.4 GHz).
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


def main():
    f = glob.glob('%s/*/*counter_collection.csv' % sys.argv[1])[0]
    per = defaultdict(lambda: defaultdict(float))      # (kernel, dispatch id) -> counter -> value
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '').strip()
        per[(k, r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for (k, _), c in per.items():
        a = agg[k]
        a[0] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        a[1] += c.get('GRBM_GUI_ACTIVE', 0.0)
        a[2] += c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0)
        a[3] += 1
    out = {}
    for k, (busy, gui, mops, n) in agg.items():
        if busy <= 0 or gui <= 0:
            continue
        cyc = gui / 8.0
        out[k] = dict(launches=n, mfma_busy_cycles_per_launch=busy / n, kernel_cycles_per_launch=cyc / n,
                      mfma_pipe_utilisation=busy / (cyc * 1024.0), mfma_mops_bf16_per_launch=mops / n)
    json.dump(dict(method='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE over `bench.py --steps 3 '
                          '--warmup 1 --no-cpu-baseline --no-events --train-only`; utilisation = busy / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)',
                   kernels=out, **build_id()), open(sys.argv[2], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['mfma_busy_cycles_per_launch'] * kv[1]['launches'])[:14]:
        print('%-46s %5d launches  %9.0f cycles/launch  MFMA pipe %5.1f %%' % (k[:46], v['launches'], v['kernel_cycles_per_launch'],
                                                                               100 * v['mfma_pipe_utilisation']))


if __name__ == '__main__':
    main()
