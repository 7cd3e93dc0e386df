"""Parity at the BENCHMARKED configuration (BASELINE.json configs[1] / configs[3]): inception_v3, bf16, batch 256.

The small-batch suites (N <= 8) take another path through the conv dispatcher than the benchmark does: at batch 256 the
17x17 / 35x35 layers run the wide-tile kernels, the weight gradients plan their split-K over whole rounds of resident
blocks, the stem runs the row-streaming kernels over 147x147x256 pixels and the persistent kernels walk several tiles per
block.  Here ONE training step (forward + loss + backward through the same op tables ``train_step`` fuses) is checked
node by node against the CPU oracle (``tests/local_parity.py``: every conv output, BatchNorm statistic, activation,
pooling result, parameter gradient and summed input gradient of the 112-node plan), and the test asserts -- through
``ifcbk_op_kernel`` -- that the kernels which dominate the benchmark are the ones that ran.
The RUN twin replays the batch-256 eval forward as a hipGraph and compares a subset of images with the oracle's eval
forward (reference: ``neuston_net.py:324`` --batch, ``neuston_models.py:81-103,152-157``)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from local_parity import check_plan, rel

pytestmark = pytest.mark.gpu

B, NC = 256, 100


def _kernels(eng, prog):
    names = set()
    buf = C.create_string_buffer(256)
    for k in range(prog.n):
        eng.ctx.lib.ifcbk_op_kernel(C.byref(prog.arr[k]), buf, 256)
        if buf.value:
            names.add(buf.value.decode())
    return names


def _has(names, prefix):
    return any(n.startswith(prefix) for n in names)


def test_batch256_train_step_node_parity_with_production_dispatch():
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    torch.manual_seed(1)
    hip = get_namebrand_model('inception_v3', NC, max_batch=B)
    eng = hip.engine
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, 3, 299, 299, generator=g)
    y = torch.randint(0, NC, (B,), generator=g)
    mask = torch.rand(B, 2048, generator=g) > 0.5
    hip.set_dropout_mask(mask.cuda())
    hip.train()
    out = hip(x.cuda())
    loss = F.cross_entropy(out.logits, y.cuda()) + 0.4 * F.cross_entropy(out.aux_logits, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item()
    # ---- the dispatch the benchmark measures
    pl = eng.plan(B)
    names = _kernels(eng, pl.step)
    print('kernels of the batch-256 step:', sorted(names))
    # the kernels that carry the benchmark, by name: the wide-tile ping-pong forward / input gradient and weight gradient, the
    # flat-image kernel of the 35x35 stage, the row-streaming stem kernels, the persistent kernel of the 8x8 layers
    for must in ('conv_pp2<', 'conv_slab<', 'conv_wgrad_pp<', 'conv_flat<', 'conv_wgrad_rows<', 'conv_wgrad_stem', 'conv_rows3x3<', 'conv_ws<', 'bn_bwd', 'bn_apply_kernel'):
        assert _has(names, must), (must, sorted(names))
    # ---- every node of the plan against the oracle, on the HIP path's own inputs
    worst = check_plan(hip, B, mask)
    print('batch-256 node-local worst rel errors:', {k: '%.2e' % v for k, v in worst.items()})
    assert worst['raw'] < 3e-3 and worst['y'] < 3e-3 and worst['pool'] < 3e-3
    assert worst['raw_cp'] < 6e-3 and worst['dW_cp'] < 2e-2
    assert worst['stats'] < 1e-4
    assert worst['head'] < 1e-4
    assert worst['dW'] < 1e-2 and worst['dgamma'] < 1e-2 and worst['dbeta'] < 1e-2
    assert worst['dx'] < 1.5e-2
    for k, b in hip.named_buffers():
        if k.endswith('num_batches_tracked'):
            assert int(b.item()) == 1


def test_resnet50_train_step_node_parity_at_the_batch_where_the_wide_tile_plans_fire():
    """the conv dispatcher is shape-driven, not model-driven: at batch 256 resnet50's layer2 / layer3 convolutions (K = 256 ... 1024
    output channels over 50,176 ... 200,704 pixels) are routed to the wide-tile kernels that inception_v3 tunes -- every node of
    ONE training step against the oracle, with the dispatch asserted (reference: neuston_models.py:37-39, neuston_net.py:324)"""
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    torch.manual_seed(5)
    Br, nc = 256, 7
    hip = get_namebrand_model('resnet50', nc, max_batch=Br)
    eng = hip.engine
    g = torch.Generator().manual_seed(6)
    x = torch.rand(Br, 3, 224, 224, generator=g)
    y = torch.randint(0, nc, (Br,), generator=g)
    hip.train()
    loss = F.cross_entropy(hip(x.cuda()), y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item()
    names = _kernels(eng, eng.plan(Br).step)
    print('kernels of the resnet50 batch-256 step:', sorted(names))
    for must in ('conv_pp2<', 'conv_wgrad_pp<'):
        assert _has(names, must), (must, sorted(names))
    worst = check_plan(hip, Br, None)
    print('resnet50 batch-256 node-local worst rel errors:', {k: '%.2e' % v for k, v in worst.items()})
    assert worst['raw'] < 3e-3 and worst['y'] < 3e-3 and worst['pool'] < 3e-3
    assert worst['stats'] < 1e-4 and worst['head'] < 1e-4
    assert worst['dW'] < 1e-2 and worst['dgamma'] < 1e-2 and worst['dbeta'] < 1e-2
    assert worst['dx'] < 1.5e-2


@pytest.mark.parametrize('B', [256, 768])        # 768: the batch the RUN headline is quoted at (bench.py run_mode)
def test_batch256_eval_hipgraph_matches_oracle_on_a_subset(B):
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    from oracle import tv_models
    torch.manual_seed(3)
    hip = get_namebrand_model('inception_v3', NC, max_batch=B)
    eng = hip.engine
    ora = tv_models.get_namebrand_model('inception_v3', NC, storage='bf16')
    ora.load_state_dict({k: v.detach().cpu().clone() for k, v in hip.state_dict().items()}, strict=True)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, 3, 299, 299, generator=g)
    sub = [0, 1, B // 2 - 1, B // 2, B - 2, B - 1]
    # calibrate the running statistics on a few images (momentum 1) so that eval activations stay O(1)
    for m in ora.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    ora.train()
    with torch.no_grad():
        ora(x[:8])
    hip.load_state_dict(ora.state_dict())
    hip.eval(); ora.eval()
    assert eng.graph_eval, 'RUN mode replays the eval forward as a hipGraph by default'
    with torch.no_grad():
        hip(x.cuda())                          # first pass captures the graph
        eh = hip(x.cuda()).cpu()               # second pass replays it
        eo = ora(x[sub])
    assert 'fwd_eval' in eng.plan(B).graphs
    o32 = tv_models.get_namebrand_model('inception_v3', NC, storage='fp32')
    o32.load_state_dict(ora.state_dict())
    o32.eval()
    with torch.no_grad():
        e32 = o32(x[sub])
    env = rel(eo, e32)
    r, r32 = rel(eh[sub], eo), rel(eh[sub], e32)
    print('batch-%d eval logits (6 of %d images): rel vs bf16-storage oracle %.3e, vs fp32 oracle %.3e, envelope %.3e'
          % (B, B, r, r32, env))
    assert r < 0.5 * env + 2e-3 and r32 < 1.25 * env + 2e-3
    # an image's logits do not depend on its neighbours in the batch (fixed statistics): the same 6 images alone go through
    # the small-grid kernels (another summation order, the same envelope)
    with torch.no_grad():
        alone = hip(x[sub].cuda()).cpu()
    assert rel(alone, eh[sub]) < 0.5 * env + 2e-3
    names = _kernels(eng, eng.plan(B).fwd_eval)
    assert _has(names, 'conv_igemm<') or _has(names, 'conv_pp2<')
