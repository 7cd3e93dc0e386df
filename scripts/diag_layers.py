"""dev tool: per-layer activation / gradient distance between the HIP path and the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ifcb_classifier_amd.neuston_models import get_namebrand_model
from oracle import tv_models

name = sys.argv[1] if len(sys.argv) > 1 else 'inception_v3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
storage = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
S = 299 if name == 'inception_v3' else 224
nc = 10
torch.manual_seed(0)
hip = get_namebrand_model(name, nc, max_batch=B)
sd = {k: v.detach().cpu().clone() for k, v in hip.state_dict().items()}
ora = tv_models.get_namebrand_model(name, nc, storage=storage)
ora.load_state_dict(sd)
x = torch.rand(B, 3, S, S); y = torch.randint(0, nc, (B,))
mask = torch.rand(B, 2048) > 0.5
if name == 'inception_v3':
    hip.set_dropout_mask(mask.cuda()); ora.dropout_mask = mask
acts = {}
def hook(nm):
    def f(m, i, o):
        o.retain_grad(); acts[nm] = o
    return f
for nm, m in ora.named_modules():
    if isinstance(m, tv_models.BasicConv2d):
        m.register_forward_hook(hook(nm))
ora.train(); out = ora(x)
loss = F.cross_entropy(out.logits, y) + 0.4 * F.cross_entropy(out.aux_logits, y) if name == 'inception_v3' else F.cross_entropy(out, y)
loss.backward()
hip.train(); oh = hip(x.cuda())
lh = F.cross_entropy(oh.logits, y.cuda()) + 0.4 * F.cross_entropy(oh.aux_logits, y.cuda()) if name == 'inception_v3' else F.cross_entropy(oh, y.cuda())
lh.backward()
eng = hip.engine
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
for n in eng.convs:
    key = n.conv_key[:-5] if n.conv_key.endswith('.conv') else None
    if key is None or key not in acts: continue
    t = eng.act[n.y.buf.id][:B, :, :, n.y.coff:n.y.coff + n.y.C].float().cpu().permute(0, 3, 1, 2)
    g = eng.grad[n.y.buf.id][:B, :, :, n.y.coff:n.y.coff + n.y.C].float().cpu().permute(0, 3, 1, 2)
    go = acts[key].grad
    pg = rel(hip._pmap[n.conv_key + '.weight'].grad.cpu(), dict(ora.named_parameters())[n.conv_key + '.weight'].grad)
    print('%-28s act %.4f  dact %.4f  dW %.4f' % (key, rel(t, acts[key].detach()), rel(g, go) if go is not None else -1, pg))
