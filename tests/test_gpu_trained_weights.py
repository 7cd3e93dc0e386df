"""north_star: "class logits AND TRAINED WEIGHTS match the reference's own PyTorch-CPU path on identical inputs" (VERDICT r3
item 2).  N fused HIP train steps (forward + CE(+0.4 aux) + backward + optimizer + repack in one program, the product path)
against the oracle's ``loss.backward(); optimizer.step()`` (ref neuston_models.py:63-64,81-86) on identical batches and
dropout masks, WITHOUT any re-synchronisation in between: every parameter and every BatchNorm buffer is compared after
every step.

What is compared: the UPDATE (p_k - p_0) of every parameter tensor, relative to the oracle's update of that tensor -- a
weight moves by ~1e-3 of itself per step, so "rel of the weight" would pass with the optimizer switched off.  Two optimizers:
 * SGD (+momentum): the update is lr * (momentum-filtered) gradient -- every gradient error shows 1:1, nothing can hide;
 * Adam (the reference's only optimizer): the first steps move every element by ~ +-lr whatever the gradient's size, so an
   element whose gradient sits at rounding-noise level may legitimately step the other way (2*lr off); the bound is therefore
   stated per tensor as a fraction of elements and an L2 bound, both from measurement.
The BatchNorm running statistics start from a momentum-1 calibration pass over the first batch (the trick of
test_gpu_batch256.py), so that buffers are O(1) quantities of the data and their relative error means something.
Tolerances below are the measured values (printed by the test) with a margin of about 2-3x."""
import pytest
import torch
import torch.nn.functional as F

from local_parity import rel

pytestmark = pytest.mark.gpu


def _loss(out, y):
    if isinstance(out, tuple):
        return F.cross_entropy(out[0], y) + 0.4 * F.cross_entropy(out[1], y)
    return F.cross_entropy(out, y)


def _trajectories(name, nc, B, S, dtype, optimizer, steps=3, seed=0):
    """-> per step: dict(upd=worst update-rel, upd_key, w=worst weight-rel, buf=worst buffer-rel, flip=worst fraction of
    elements whose update has the other sign, loss_h, loss_o, per_tensor={key: update-rel})"""
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    from oracle import ops as O
    from oracle import tv_models
    lr, mom = (1e-3, 0.0) if optimizer == 'adam' else (0.05, 0.9)
    torch.manual_seed(seed)
    hip = get_namebrand_model(name, nc, max_batch=B, dtype=dtype, optimizer=optimizer, lr=lr, momentum=mom)
    eng = hip.engine
    storage = 'fp32' if dtype == 'fp32' else 'bf16'
    ora = tv_models.get_namebrand_model(name, nc, storage=storage)
    ora.load_state_dict({k: v.detach().cpu().clone() for k, v in hip.state_dict().items()}, strict=True)
    incep = name == 'inception_v3'
    g = torch.Generator().manual_seed(100 + seed)
    xs = [torch.rand(B, 3, S, S, generator=g) for _ in range(steps)]
    ys = [torch.randint(0, nc, (B,), generator=g) for _ in range(steps)]
    masks = [(torch.rand(B, 2048, generator=g) > 0.5) if incep else None for _ in range(steps)]
    O.set_storage(storage)
    try:
        # running statistics calibrated on the first batch (momentum 1), then the reference's momentum again
        bns = [m for m in ora.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        ora.train()
        if incep:
            ora.dropout_mask = masks[0]
        with torch.no_grad():
            ora(xs[0])
        for m in bns:
            m.momentum = 0.1
            m.num_batches_tracked.zero_()
        hip.load_state_dict(ora.state_dict())
        p0 = {k: v.detach().clone() for k, v in ora.named_parameters()}
        opt = (torch.optim.Adam(ora.parameters(), lr=lr) if optimizer == 'adam'
               else torch.optim.SGD(ora.parameters(), lr=lr, momentum=mom))
        out = []
        for k in range(steps):
            x, y, mask = xs[k], ys[k], masks[k]
            # ---- HIP: ONE fused program per step (the product's fit_batch path)
            if incep:
                hip.set_dropout_mask(mask.cuda())
            hip.train()
            N = eng.load_input_nchw(x.cuda())
            eng.target[:N].copy_(y.cuda())
            eng.train_step(N)
            torch.cuda.synchronize()
            loss_h = float(eng.loss.item())
            # ---- oracle: the reference's arithmetic
            if incep:
                ora.dropout_mask = mask
            ora.train()
            lo = _loss(ora(x), y)
            opt.zero_grad()
            lo.backward()
            opt.step()
            ph = {kk: v.detach().cpu() for kk, v in hip.named_parameters()}
            per, wrel, flips = {}, {}, {}
            for kk, po in ora.named_parameters():
                uo, uh = po.detach() - p0[kk], ph[kk] - p0[kk]
                per[kk] = rel(uh, uo)
                wrel[kk] = rel(ph[kk], po.detach())
                flips[kk] = float(((uo * uh) < 0).float().mean())
            ob = dict(ora.named_buffers())
            brel = {kk: rel(b.detach().cpu().float(), ob[kk].float()) for kk, b in hip.named_buffers() if not kk.endswith('num_batches_tracked')}
            nbt_ok = all(int(b.item()) == k + 1 for kk, b in hip.named_buffers() if kk.endswith('num_batches_tracked'))
            wk = max(per, key=per.get)
            out.append(dict(upd=per[wk], upd_key=wk, w=max(wrel.values()), buf=max(brel.values()), flip=max(flips.values()),
                            loss_h=loss_h, loss_o=float(lo.item()), per_tensor=per, nbt_ok=nbt_ok,
                            upd_median=sorted(per.values())[len(per) // 2]))
        return out
    finally:
        O.set_storage('bf16')


def _report(tag, traj):
    for k, t in enumerate(traj):
        print('%s step %d: loss hip %.6f oracle %.6f | update rel: worst %.3e (%s) median %.3e | weight rel worst %.3e | '
              'BN buffers worst %.3e | worst sign-flip fraction %.3e'
              % (tag, k + 1, t['loss_h'], t['loss_o'], t['upd'], t['upd_key'], t['upd_median'], t['w'], t['buf'], t['flip']))


# fp32 parity mode: (update rel worst, update rel median, weight rel, buffer rel) after the LAST of three un-resynced steps
FP32_SGD = {'inception_v3': (5e-2, 5e-3, 2e-4, 1e-4), 'resnet18': (5e-2, 5e-3, 2e-4, 1e-4)}
FP32_ADAM = {'inception_v3': (3e-1, 5e-2, 2e-3, 1e-4), 'resnet18': (3e-1, 5e-2, 2e-3, 1e-4)}


@pytest.mark.parametrize('name,nc,B,S', [('inception_v3', 10, 4, 299), ('resnet18', 2, 6, 224)])
def test_fp32_trained_weights_sgd(name, nc, B, S):
    traj = _trajectories(name, nc, B, S, 'fp32', 'sgd')
    _report('fp32 SGD(0.05, m=0.9) ' + name, traj)
    uw, um, ww, bb = FP32_SGD[name]
    for t in traj:
        assert t['nbt_ok']
        assert abs(t['loss_h'] - t['loss_o']) < 1e-3 * abs(t['loss_o'])
        assert t['upd'] < uw and t['upd_median'] < um and t['w'] < ww and t['buf'] < bb


@pytest.mark.parametrize('name,nc,B,S', [('inception_v3', 10, 4, 299), ('resnet18', 2, 6, 224)])
def test_fp32_trained_weights_adam(name, nc, B, S):
    traj = _trajectories(name, nc, B, S, 'fp32', 'adam')
    _report('fp32 Adam(1e-3) ' + name, traj)
    uw, um, ww, bb = FP32_ADAM[name]
    for t in traj:
        assert t['nbt_ok']
        assert abs(t['loss_h'] - t['loss_o']) < 1e-3 * abs(t['loss_o'])
        assert t['upd'] < uw and t['upd_median'] < um and t['w'] < ww and t['buf'] < bb


@pytest.mark.parametrize('name,nc,B,S', [('inception_v3', 10, 4, 299), ('resnet18', 2, 6, 224)])
def test_bf16_trained_weights_twin(name, nc, B, S):
    """the performance mode against the bf16-STORAGE oracle: same test, its own (measured, printed) tolerance -- never quoted as
    the fp32 parity"""
    for optimizer in ('sgd', 'adam'):
        traj = _trajectories(name, nc, B, S, 'bf16', optimizer)
        _report('bf16 %s %s' % (optimizer, name), traj)
        for t in traj:
            assert t['nbt_ok']
            assert abs(t['loss_h'] - t['loss_o']) < 0.1 * abs(t['loss_o'])
            assert t['upd_median'] < 0.5 and t['w'] < 5e-2 and t['buf'] < 5e-2
