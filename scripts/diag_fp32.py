import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
from local_parity import check_plan
from oracle import ops as O
from ifcb_classifier_amd.neuston_models import get_namebrand_model
torch.manual_seed(0)
B, nc, S = 4, 10, 299
hip = get_namebrand_model('inception_v3', nc, max_batch=B, dtype='fp32')
g = torch.Generator().manual_seed(11)
x = torch.rand(B, 3, S, S, generator=g); y = torch.randint(0, nc, (B,), generator=g)
mask = torch.rand(B, 2048, generator=g) > 0.5
hip.set_dropout_mask(mask.cuda()); hip.train()
out = hip(x.cuda())
loss = F.cross_entropy(out.logits, y.cuda()) + 0.4 * F.cross_entropy(out.aux_logits, y.cuda())
loss.backward()
O.set_storage('fp32')
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    worst = check_plan(hip, B, mask, verbose=True)
lines = [l for l in buf.getvalue().splitlines() if float(l.split()[-1]) > 1e-4]
print('\n'.join(lines[:40])); print(worst)
