"""dev tool: vgg11 fp32, every stored activation and activation gradient of the HIP path against torch autograd"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import torch.nn.functional as F
import test_gpu_families as T

name, nc, B = 'vgg11', 4, 2
hip, ora = T._pair(name, nc, B, 'fp32')
g = torch.Generator().manual_seed(5)
x = torch.rand(B, 3, 224, 224, generator=g)
y = torch.randint(0, nc, (B,), generator=g)
mo, mh = T._masks(name, B, g)
hip.set_dropout_mask(mh)
hip.train()
out_h = hip(x.cuda())
F.cross_entropy(out_h, y.cuda()).backward()
torch.cuda.synchronize()
# oracle by hand, keeping every tensor
acts = {}
t = x
mods = list(ora.features)
k = 0
while k < len(mods):
    m = mods[k]
    if isinstance(m, torch.nn.MaxPool2d):
        t = F.max_pool2d(t, 2, 2); t.retain_grad(); acts['features.%d' % k] = t; k += 1
    else:
        t = F.relu(F.conv2d(t, m.weight, m.bias, m.stride, m.padding)); t.retain_grad(); acts['features.%d:y' % k] = t; k += 2
f = torch.flatten(t, 1)
c = ora.classifier
h1 = F.relu(F.linear(f, c[0].weight, c[0].bias)); d1 = h1 * mo['classifier.drop0'].float() * 2
h2 = F.relu(F.linear(d1, c[3].weight, c[3].bias)); d2 = h2 * mo['classifier.drop1'].float() * 2
lo = c[6](d2)
F.cross_entropy(lo, y).backward()
eng = hip.engine
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
for b in eng.net.bufs:
    if b.name in acts:
        a = acts[b.name]
        ah = eng.act[b.id][:B].float().cpu().permute(0, 3, 1, 2)
        line = '%-16s act %.1e' % (b.name, rel(ah, a.detach()))
        if b.id in eng.grad:
            gh = eng.grad[b.id][:B].float().cpu().permute(0, 3, 1, 2)
            gref = a.grad * (a.detach() > 0) if b.name.endswith(':y') else a.grad
            e = (gh - gref).abs()
            line += '  grad %.1e  (wrong > 1e-3 max: %d of %d)' % (rel(gh, gref), int((e > 1e-3 * float(gref.abs().max())).sum()), e.numel())
        print(line)
