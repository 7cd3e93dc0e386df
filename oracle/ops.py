"""Node-level CPU oracle: the torch-CPU primitives the reference's hot path executes
(``F.conv2d``, ``F.batch_norm``, ``F.max_pool2d``, ``F.avg_pool2d``, ``F.linear``, ``F.cross_entropy``,
``torch.optim.Adam``; call sites ``/root/reference/neuston_models.py:55,63-78,99,156`` via torchvision's
graphs), wrapped so that a test can check ONE graph node of the HIP plan at a time on the HIP path's own
inputs ("teacher forcing").  Random-init BatchNorm networks amplify a 1e-6 input perturbation ~1000x over
inception_v3's 47-conv depth, so only node-local comparisons can be tight; the end-to-end tests bound the
global distance separately.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).
"""
import torch
import torch.nn.functional as F


STORAGE = 'bf16'      # 'bf16': emulate the HIP path's bf16 storage rounding; 'fp32': the reference's own arithmetic


def set_storage(kind):
    global STORAGE
    assert kind in ('bf16', 'fp32')
    STORAGE = kind


def bf16_round(x):
    if STORAGE == 'fp32':
        return x
    return x.to(torch.bfloat16).to(torch.float32)


def conv_raw(x, w, stride, pad):
    """bf16-stored conv output: fp32 accumulate of bf16 operands, rounded once."""
    return bf16_round(F.conv2d(x, bf16_round(w), None, stride, pad))


def bn_act_fwd(raw, gamma, beta, eps, relu, residual=None):
    """train-mode BN (+residual) (+ReLU) on the stored raw conv output; returns (y_bf16, mean, var_biased)."""
    mean = raw.mean((0, 2, 3))
    var = raw.var((0, 2, 3), unbiased=False)
    y = F.batch_norm(raw, None, None, gamma, beta, True, 0.1, eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return bf16_round(y), mean, var


def bn_act_bwd(raw, gamma, beta, eps, relu, residual, gy, y_for_mask=None):
    """backward of bn_act_fwd at (raw, residual) for upstream gradient gy.
    returns d_raw (bf16-rounded, as the HIP path stores it), dgamma, dbeta, dresidual (or None).
    ``y_for_mask``: take the ReLU mask from this (the checked path's own) activation instead of recomputing it --
    an element whose pre-activation is zero to within rounding may legitimately land on either side of the ReLU
    kink under a different but equally valid evaluation order, and a few such elements dominate a 256-sample
    channel sum; the forward comparison already bounds the activation itself."""
    raw = raw.clone().requires_grad_(True)
    g = gamma.clone().requires_grad_(True)
    b = beta.clone().requires_grad_(True)
    res = residual.clone().requires_grad_(True) if residual is not None else None
    y = F.batch_norm(raw, None, None, g, b, True, 0.1, eps)
    if res is not None:
        y = y + res
    if relu:
        y = y * (y_for_mask > 0).to(y.dtype) if y_for_mask is not None else F.relu(y)
    y.backward(gy)
    return bf16_round(raw.grad), g.grad, b.grad, (res.grad if res is not None else None)


def conv_bwd(x, w, d_raw, stride, pad, need_dx=True):
    wb = bf16_round(w)
    dw = torch.nn.grad.conv2d_weight(x, w.shape, d_raw, stride, pad)
    dx = torch.nn.grad.conv2d_input(x.shape, wb, d_raw, stride, pad) if need_dx else None
    return dw, dx


def pool_fwd(kind, x, k, stride, pad):
    if kind == 'max':
        return F.max_pool2d(x, k, stride, pad)
    return bf16_round(F.avg_pool2d(x, k, stride, pad))


def pool_bwd(kind, x, k, stride, pad, gy):
    x = x.clone().requires_grad_(True)
    y = F.max_pool2d(x, k, stride, pad) if kind == 'max' else F.avg_pool2d(x, k, stride, pad)
    y.backward(gy)
    return x.grad


def head_fwd(x, W, b, mask=None):
    """adaptive_avg_pool2d(1) -> dropout(0.5) with an explicit keep mask -> linear."""
    feat = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    if mask is not None:
        feat = feat * (mask.to(feat.dtype) * 2.0)
    return feat, F.linear(feat, W, b)


def head_bwd(x, W, b, mask, dlogits):
    x = x.clone().requires_grad_(True)
    W = W.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    feat, logits = head_fwd(x, W, b, mask)
    logits.backward(dlogits)
    return bf16_round(x.grad), W.grad, b.grad


def xent(logits, target, weight=1.0):
    """returns (weight*mean CE, d/dlogits)"""
    l = logits.clone().requires_grad_(True)
    loss = weight * F.cross_entropy(l, target)
    loss.backward()
    return loss.detach(), l.grad


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's update (torch 1.7.1 formula: denom = sqrt(v)/sqrt(bc2) + eps)."""
    p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps)
    if step > 1:
        opt.state[p] = dict(step=torch.tensor(float(step - 1)), exp_avg=m.clone(), exp_avg_sq=v.clone())
    p.grad = g.clone()
    opt.step()
    st = opt.state[p]
    return p.detach(), st['exp_avg'], st['exp_avg_sq']
