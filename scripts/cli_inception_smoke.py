"""dev tool: neuston_net TRAIN + RUN with inception_v3 on a small synthetic PNG dataset (the test suite's CLI round trip uses
resnet18): exercises the 299-pixel resize, the auxiliary head, partial last batches and the hipGraph eval forward of RUN."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from ifcb_classifier_amd import neuston_net as nn_

root = tempfile.mkdtemp()
src = os.path.join(root, 'data')
rng = np.random.default_rng(7)
for cls, mean in (('a_dark', 80), ('b_mid', 128), ('c_bright', 176)):
    os.makedirs(os.path.join(src, cls))
    for i in range(37):
        h, w = rng.integers(32, 200, 2)
        Image.fromarray(np.clip(rng.normal(mean, 24, (h, w)), 0, 255).astype(np.uint8), 'L').save(os.path.join(src, cls, '%s_%03d.png' % (cls, i)))
out = os.path.join(root, 'out')


def cli(argv):
    args = nn_.argparse_nn().parse_args(argv)
    nn_.argparse_nn_runtimeparams(args)
    nn_.main(args)


cli(['--batch', '32', '--loaders', '0', 'TRAIN', src, 'inception_v3', 'incsmoke', '--untrain', '--seed', '3', '--emax', '4', '--emin', '1',
     '--estop', '0', '--outdir', out, '--results', 'results.json', 'output_scores', 'f1_macro'])
rows = open(os.path.join(out, 'epochs.csv')).read().strip().splitlines()
print('\n'.join(rows))
tl = [float(r.split(',')[2]) for r in rows[1:]]
assert tl[-1] < tl[0], tl
run_out = os.path.join(root, 'run')
cli(['--batch', '50', '--loaders', '0', 'RUN', src + os.sep, os.path.join(out, 'incsmoke.ptl'), 'runid', '--type', 'img', '--outdir', run_out,
     '--outfile', 'img_results.json'])
res = json.load(open(os.path.join(run_out, 'img_results.json')))
sc = np.array(res['output_scores'])
print('RUN:', sc.shape, 'row sums', sc.sum(1).min(), sc.sum(1).max())
assert sc.shape == (111, 3) and np.allclose(sc.sum(1), 1.0, atol=1e-4)
print('inception CLI smoke ok')
