import sys, os, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_gpu_cli import _make_dataset, _cli
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, 'training-data'); os.makedirs(src); _make_dataset(src)
outdir = os.path.join(tmp, 'out')
_cli(['--batch', '16', '--loaders', '0', 'TRAIN', src, 'resnet18', 'smoke', '--untrain', '--seed', '1', '--emax', '6', '--emin', '1', '--estop', '0', '--outdir', outdir, '--results', 'results.json', 'image_basenames', 'output_scores'])
print(open(os.path.join(outdir, 'epochs.csv')).read())
res = json.load(open(os.path.join(outdir, 'results.json')))
print('val scores', np.array(res['output_scores'])[:5], res['input_classes'][:10], res['output_classes'][:10])
from ifcb_classifier_amd.neuston_models import NeustonModel
from ifcb_classifier_amd.neuston_data import ImageDataset, collate_rois, rois_to_device
m = NeustonModel.load_from_checkpoint(os.path.join(outdir, 'smoke.ptl'), max_batch=16)
paths = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(src) for f in fs)
ds = ImageDataset(paths[:8] + paths[-8:], resize=224)
batch, ids = collate_rois([ds[i] for i in range(16)])
probs, _ = m.eval_batch(rois_to_device(batch, 'cuda', ds.transform))
print('run probs', probs.cpu().numpy().round(3))
ck = torch.load(os.path.join(outdir, 'smoke.ptl'), weights_only=False)
print('ckpt epoch', ck['epoch'], 'bn1 running_mean', ck['state_dict']['model.bn1.running_mean'][:4], 'nbt', ck['state_dict']['model.bn1.num_batches_tracked'])
