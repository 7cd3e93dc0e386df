import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ifcb_classifier_amd.neuston_models import get_namebrand_model
from local_parity import check_plan
import io, contextlib
torch.manual_seed(0)
B, nc = 4, 10
hip = get_namebrand_model('inception_v3', nc, max_batch=B)
x = torch.rand(B, 3, 299, 299); y = torch.randint(0, nc, (B,))
mask = torch.rand(B, 2048) > 0.5
hip.set_dropout_mask(mask.cuda()); hip.train()
out = hip(x.cuda())
loss = F.cross_entropy(out.logits, y.cuda()) + 0.4 * F.cross_entropy(out.aux_logits, y.cuda())
loss.backward()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    worst = check_plan(hip, B, mask, verbose=True)
for line in buf.getvalue().splitlines():
    parts = line.split()
    try:
        if float(parts[-1]) > 1e-2 and parts[0] in ('dgamma', 'dbeta', 'dW', 'dx'):
            print(line)
    except Exception:
        pass
print(worst)
