"""Per-op roofline table of one training step (SURVEY.md hard-part 1): for every conv launch of `bench.py --dump-ops ops.json`
the achieved TFLOP/s against min(MFMA peak, arithmetic intensity x HBM bandwidth), and the MAC-weighted fractions.
    python bench.py --dump-ops gpurun_out/ops.json ...;  python scripts/layer_roofline.py gpurun_out/ops.json out.md"""
import json
import sys

PEAK, BW = 2500.0, 6.29            # TF/s dense bf16 MFMA; TB/s measured copy bandwidth (MI355X_MICROARCH.md)
rows = json.load(open(sys.argv[1]))
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
conv = [r for r in rows if r['gflop'] > 0 and r['op'].startswith('conv')]
tot_ms = sum(r['ms'] for r in rows)
print('# per-op roofline of one training step (batch 256, every op alone on the GPU, one lane)\n', file=out)
print('step total (sum of isolated ops): %.2f ms; conv ops: %.2f ms, %.1f TFLOP/s = %.1f %% of %d TF/s\n'
      % (tot_ms, sum(r['ms'] for r in conv), sum(r['gflop'] for r in conv) / sum(r['ms'] for r in conv),
         100 * sum(r['gflop'] for r in conv) / sum(r['ms'] for r in conv) / PEAK, PEAK), file=out)
print('| op | layer | kernel | GFLOP | MB (min) | AI flop/B | bound TF/s = min(peak, AI*BW) | ms | achieved TF/s | of bound | of MFMA peak |', file=out)
print('|---|---|---|---|---|---|---|---|---|---|---|', file=out)
wsum = wb = 0.0
for r in conv:
    ai = r['gflop'] * 1e3 / max(r['mbytes'], 1e-9)
    bound = min(PEAK, ai * BW)
    ach = r['gflop'] / r['ms']
    wsum += r['gflop']
    wb += r['gflop'] / bound
    print('| %s | %s | %s | %.1f | %.1f | %.0f | %.0f | %.3f | %.0f | %.2f | %.3f |'
          % (r['op'], r['tag'][:44], r['kernel'], r['gflop'], r['mbytes'], ai, bound, r['ms'], ach, ach / bound, ach / PEAK), file=out)
best_ms = wb                       # sum flop / bound = the time a roofline-perfect step would need for the conv ops
print('\nroofline-perfect time of the conv ops: %.2f ms (measured %.2f ms): MAC-weighted achieved / bound = %.3f'
      % (best_ms, sum(r['ms'] for r in conv), best_ms / sum(r['ms'] for r in conv)), file=out)
other = {}
for r in rows:
    if r not in conv:
        k = r['kernel'] or r['op']
        o = other.setdefault(k, [0.0, 0.0, 0])
        o[0] += r['ms']; o[1] += r['mbytes']; o[2] += 1
print('\n| non-conv kernel | launches | ms | min MB | TB/s vs minimum bytes |\n|---|---|---|---|---|', file=out)
for k, (ms, mb, n) in sorted(other.items(), key=lambda kv: -kv[1][0]):
    print('| %s | %d | %.3f | %.0f | %.2f |' % (k, n, ms, mb, mb / ms / 1e3 if ms > 0 else 0), file=out)
