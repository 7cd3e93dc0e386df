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

How tight can this be?  A random-init BatchNorm network amplifies rounding noise: the fp32 oracle's own end-to-end gradients
move by ~1e-2 when its arithmetic is done in fp64 instead (inception's AuxLogits.conv1 normalises over N x 1 x 1 values: the
batch must not be tiny, 16 here).  The test therefore runs a THIRD trajectory, the oracle in fp64, as the arbiter: the HIP
fp32 mode must be as close to the fp64 trajectory as the fp32 reference arithmetic itself is (factor ARB), and close to the
fp32 reference in absolute terms (measured values, printed, with a margin of about 2-3x)."""
import pytest
import torch
import torch.nn.functional as F

from local_parity import rel

pytestmark = pytest.mark.gpu


def _loss(out, y):
    if isinstance(out, tuple):
        return F.cross_entropy(out[0], y) + 0.4 * F.cross_entropy(out[1], y)
    return F.cross_entropy(out, y)


ARB = 1.5      # HIP-vs-fp64 distance allowed, in units of the oracle's own distance to the fp64 oracle (measured: 1.03-1.28)


def _trajectories(name, nc, B, S, dtype, optimizer, steps=3, seed=0, arbiter=False):
    """-> per step: dict(upd=worst update-rel, upd_key, w=worst weight-rel, buf=worst buffer-rel, flip=worst fraction of
    elements whose update has the other sign, loss_h, loss_o, per_tensor={key: update-rel})"""
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    from oracle import ops as O
    from oracle import tv_models
    lr, mom = (1e-3, 0.0) if optimizer == 'adam' else (0.005, 0.9)
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
        mk = (lambda ps: torch.optim.Adam(ps, lr=lr)) if optimizer == 'adam' else (lambda ps: torch.optim.SGD(ps, lr=lr, momentum=mom))
        opt = mk(ora.parameters())
        o64 = opt64 = None
        if arbiter:
            o64 = tv_models.get_namebrand_model(name, nc, storage='fp32').double()
            o64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ora.state_dict().items()})
            opt64 = mk(o64.parameters())
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
            p64 = None
            if o64 is not None:
                if incep:
                    o64.dropout_mask = mask
                o64.train()
                l64 = _loss(o64(x.double()), y)
                opt64.zero_grad()
                l64.backward()
                opt64.step()
                p64 = {kk: v.detach() for kk, v in o64.named_parameters()}
            per, wrel, flips, arb_h, arb_o = {}, {}, {}, {}, {}
            for kk, po in ora.named_parameters():
                uo, uh = po.detach() - p0[kk], ph[kk] - p0[kk]
                per[kk] = rel(uh, uo)
                wrel[kk] = rel(ph[kk], po.detach())
                flips[kk] = float(((uo * uh) < 0).float().mean())
                if p64 is not None:
                    u64 = p64[kk] - p0[kk].double()
                    arb_h[kk], arb_o[kk] = rel(uh.double(), u64), rel(uo.double(), u64)
            ob = dict(ora.named_buffers())
            hb = {kk: b.detach().cpu().float() for kk, b in hip.named_buffers() if not kk.endswith('num_batches_tracked')}
            brel = {kk: rel(b, ob[kk].float()) for kk, b in hb.items()}
            barb_h = barb_o = None
            if o64 is not None:
                b64 = dict(o64.named_buffers())
                barb_h = max(rel(b.double(), b64[kk].double()) for kk, b in hb.items())
                barb_o = max(rel(ob[kk].double(), b64[kk].double()) for kk in hb)
            nbt_ok = all(int(b.item()) == k + 1 for kk, b in hip.named_buffers() if kk.endswith('num_batches_tracked'))
            wk = max(per, key=per.get)
            out.append(dict(upd=per[wk], upd_key=wk, w=max(wrel.values()), buf=max(brel.values()), flip=max(flips.values()),
                            loss_h=loss_h, loss_o=float(lo.item()), per_tensor=per, nbt_ok=nbt_ok,
                            upd_median=sorted(per.values())[len(per) // 2],
                            arb_h=arb_h, arb_o=arb_o, barb_h=barb_h, barb_o=barb_o))
        return out
    finally:
        O.set_storage('bf16')


def _med(d):
    v = sorted(d.values())
    return v[len(v) // 2]


def _report(tag, traj):
    for k, t in enumerate(traj):
        print('%s step %d: loss hip %.6f oracle %.6f | update rel: worst %.3e (%s) median %.3e | weight rel worst %.3e | '
              'BN buffers worst %.3e | worst sign-flip fraction %.3e'
              % (tag, k + 1, t['loss_h'], t['loss_o'], t['upd'], t['upd_key'], t['upd_median'], t['w'], t['buf'], t['flip']))
        if t['arb_h']:
            print('%s step %d: update distance to the fp64 oracle: HIP median %.3e worst %.3e | oracle median %.3e worst %.3e | '
                  'BN buffers: HIP %.3e oracle %.3e'
                  % (tag, k + 1, _med(t['arb_h']), max(t['arb_h'].values()), _med(t['arb_o']), max(t['arb_o'].values()),
                     t['barb_h'], t['barb_o']))


def _arbitrated(t):
    """HIP no farther from the fp64 trajectory than ARB x the fp32 reference arithmetic is (median and worst tensor)"""
    # (+5e-4 of the update: where the oracle itself sits 1e-4 from its fp64 twin, a ratio of two rounding-noise figures says nothing)
    return (_med(t['arb_h']) <= ARB * _med(t['arb_o']) + 5e-4 and max(t['arb_h'].values()) <= ARB * max(t['arb_o'].values()) + 5e-4
            and t['barb_h'] <= ARB * t['barb_o'] + 1e-6)


# fp32 parity mode, FIRST step (before the trajectories' own chaos compounds): update rel (worst tensor, median), weight rel, BN
# buffers.  Measured on MI355X at the default sizes below (HIP vs fp32 oracle | the fp32 oracle ITSELF vs its fp64 twin):
#   SGD  inception_v3 (batch 8)   3.4e-2 / 2.3e-2 / 3.4e-2 / 3.5e-6 | 3.7e-2 / 2.0e-2      resnet18 (batch 16)  7.4e-3 / 3.7e-3 / 5.4e-3 / 7.7e-8 | 1.1e-2 / 2.9e-3
#   Adam inception_v3             4.3e-1 / 1.6e-1                   | 3.5e-1 / 1.4e-1      resnet18             2.5e-1 / 9.3e-4                   | 2.5e-1 / 1.2e-4
# SGD: absolute first-step bounds at 2 x the measured values.  Adam gets NO absolute bound (round 4's (0.9, 0.3, 0.9) passed a 90 %
# update error: vacuous): its first step is -lr * sign(g), so a tensor whose gradient is at rounding-noise level flips whole
# elements in ANY arithmetic -- the fp32 reference itself sits 0.35 from its fp64 twin -- and the only statement with teeth is the
# arbitrated one (_arbitrated: HIP no farther from the fp64 trajectory than ARB x the fp32 reference is), asserted at every step.
# After step 1 the fp32 and fp64 ORACLES drift apart by 0.55 and 0.77 of the update (inception, SGD steps 2 and 3), and so does
# everything else: from step 2 on only the arbitrated bound means anything for either optimizer.  bf16 storage: both the HIP path
# and the bf16-storage oracle sit 1.3 (inception) / 0.4-0.7 (resnet18) of the update away from the fp64 trajectory from the FIRST
# step on -- equal to each other within 5 %.
STEP1 = {('sgd', 'inception_v3'): (7e-2, 4.6e-2, 7e-2, 1e-5), ('sgd', 'resnet18'): (1.5e-2, 7.5e-3, 1.1e-2, 2e-7)}
# Default sizes keep the GPU suite short (every HIP step costs an fp32 AND an fp64 oracle step on the host): inception_v3 at batch 8
# for two steps, resnet18 at batch 16 for three -- in fp32 mode (SGD, Adam) AND in bf16 (SGD; round 5: inception_v3 too).
# IFCBK_FULL_TESTS=1 adds nothing new in kind: batch 16, three steps, Adam for the bf16 twin as well (4.5 minutes).
import os
FULL = os.environ.get('IFCBK_FULL_TESTS', '0') != '0'
CASES = [('inception_v3', 10, 16 if FULL else 8, 299, 3 if FULL else 2), ('resnet18', 2, 16, 224, 3)]


def _ratios(tag, traj):
    """the arbitrated ratios as one line per step (what GPUTEST_rNN shows): HIP-to-fp64 distance / oracle-to-fp64 distance"""
    import conftest
    for k, t in enumerate(traj):
        mo, wo = _med(t['arb_o']), max(t['arb_o'].values())
        ln = ('trained weights, %s step %d: update distance to the fp64 trajectory, HIP %.3e / oracle %.3e (median over tensors), ratio %.3f; '
              'worst tensor ratio %.3f; BN buffers ratio %.3f [bound %.1f]; loss hip %.6f oracle %.6f'
              % (tag, k + 1, _med(t['arb_h']), mo, _med(t['arb_h']) / max(mo, 1e-30), max(t['arb_h'].values()) / max(wo, 1e-30),
                 t['barb_h'] / max(t['barb_o'], 1e-30), ARB, t['loss_h'], t['loss_o']))
        print(ln)
        conftest.MEASURED.append(ln)


def _check(opt, name, traj, loss_tol):
    t = traj[0]
    if (opt, name) in STEP1:
        uw, um, ww, bb = STEP1[(opt, name)]
        assert t['upd'] < uw and t['upd_median'] < um and t['w'] < ww and t['buf'] < bb
    assert abs(t['loss_h'] - t['loss_o']) < 1e-4 * abs(t['loss_o'])          # same weights, same batch: measured 6e-7
    for t in traj:
        assert t['nbt_ok']
        # (from step 2 on the loss is evaluated on weights that already differ by the chaotic update distances above: 9e-3 at batch 8)
        assert abs(t['loss_h'] - t['loss_o']) < loss_tol * abs(t['loss_o'])
        assert _arbitrated(t)


@pytest.mark.parametrize('name,nc,B,S,steps', CASES)
def test_fp32_trained_weights_sgd(name, nc, B, S, steps):
    traj = _trajectories(name, nc, B, S, 'fp32', 'sgd', steps=steps, arbiter=True)
    _report('fp32 SGD(0.005, m=0.9) ' + name, traj)
    _ratios('fp32 SGD ' + name, traj)
    _check('sgd', name, traj, 4e-2)


@pytest.mark.parametrize('name,nc,B,S,steps', CASES)
def test_fp32_trained_weights_adam(name, nc, B, S, steps):
    traj = _trajectories(name, nc, B, S, 'fp32', 'adam', steps=steps, arbiter=True)
    _report('fp32 Adam(1e-3) ' + name, traj)
    _ratios('fp32 Adam ' + name, traj)
    _check('adam', name, traj, 4e-2)


@pytest.mark.parametrize('name,nc,B,S,steps', CASES)
def test_bf16_trained_weights_twin(name, nc, B, S, steps):
    """the performance mode -- the HEADLINE configuration's model and dtype: inception_v3 in bf16 runs by default --
    against the bf16-STORAGE oracle, arbitrated by the same fp64 trajectory: the HIP bf16 path must be as close to it as the
    oracle's own bf16-storage arithmetic is -- its own (measured, printed) numbers, never quoted as the fp32 parity"""
    for optimizer in (('sgd', 'adam') if FULL else ('sgd',)):
        traj = _trajectories(name, nc, B, S, 'bf16', optimizer, steps=steps if FULL else 2, arbiter=True)
        _report('bf16 %s %s' % (optimizer, name), traj)
        _ratios('bf16 %s %s' % (optimizer, name), traj)
        for t in traj:
            assert t['nbt_ok']
            assert abs(t['loss_h'] - t['loss_o']) < 0.1 * abs(t['loss_o'])
            assert _arbitrated(t)
