"""Host-side checks (no GPU) of the backbone families beyond inception_v3 / resnet: the oracle restatement of torchvision 0.8.2's
alexnet / vgg / squeezenet1_1 / densenet graphs is pinned by the published parameter totals, the engine's graphs register
exactly the oracle's parameters (names, shapes, order = optimizer-state order) and buffers, and the REAL op tables of each family
build without a device (plan_only) and cover the flat gradient buffer for the data-parallel bucket plan.
Reference: neuston_models.py:22-45."""
import math

import pytest
import torch

# torchvision's published parameter counts at 1000 classes
PUBLISHED = {'alexnet': 61100840, 'vgg11': 132863336, 'vgg13': 133047848, 'vgg16': 138357544, 'vgg19': 143667240,
             'vgg11_bn': 132868840, 'vgg13_bn': 133053736, 'vgg16_bn': 138365992, 'vgg19_bn': 143678248,
             'squeezenet': 1235496, 'densenet121': 7978856, 'densenet161': 28681000, 'densenet169': 14149480,
             'densenet201': 20013928}


@pytest.mark.parametrize('name', sorted(PUBLISHED))
def test_oracle_and_engine_graph_agree_with_the_published_architecture(name):
    from ifcb_classifier_amd import graph
    from oracle import tv_models
    m = tv_models.get_namebrand_model(name, 1000)
    assert sum(p.numel() for p in m.parameters()) == PUBLISHED[name]
    net = graph.build(name, 1000)
    assert [(k, tuple(s)) for k, s, _kind, _n in net.params] == [(k, tuple(p.shape)) for k, p in m.named_parameters()]
    ob = [k for k, _ in m.named_buffers() if not k.endswith('num_batches_tracked')]
    assert [k for k, _s, _n in net.buffers] == ob
    assert sum(math.prod(s) for _k, s, _kind, _n in net.params) == PUBLISHED[name]


def test_head_replacement_follows_the_reference():
    """neuston_models.py:27-42: classifier[6] / classifier[1] (a 1x1 conv) / classifier are rebuilt for num_o_classes"""
    from ifcb_classifier_amd import graph
    shapes = lambda name: {k: tuple(s) for k, s, _kind, _n in graph.build(name, 7).params}
    assert shapes('alexnet')['classifier.6.weight'] == (7, 4096)
    assert shapes('vgg16')['classifier.6.weight'] == (7, 4096)
    assert shapes('squeezenet')['classifier.1.weight'] == (7, 512, 1, 1)
    assert shapes('densenet121')['classifier.weight'] == (7, 1024)
    with pytest.raises(KeyError, match='model unknown!'):
        graph.build('googlenet', 7)


@pytest.mark.parametrize('name,B', [('alexnet', 4), ('vgg11', 2), ('vgg11_bn', 2), ('squeezenet', 4), ('densenet121', 2)])
def test_op_tables_build_and_cover_the_gradient_buffer(name, B):
    from ifcb_classifier_amd import _lib, graph
    from ifcb_classifier_amd.engine import Engine
    eng = Engine(graph.build(name, 10), 0, max_batch=B, plan_only=True)
    pl = eng.plan(B)
    assert pl.fwd_train.n and pl.fwd_eval.n and pl.bwd.n
    segs = eng.ddp_segments(pl)
    # the buckets tile the flat gradient buffer from its tail to its head, every backward op belongs to one segment
    assert segs[0][3] == eng.nparam_padded and segs[-1][2] == 0
    for a, b in zip(segs, segs[1:]):
        assert a[2] == b[3]
    assert sum(s[0].n for s in segs) == pl.bwd.n
    kinds = [pl.bwd.arr[k].kind for k in range(pl.bwd.n)]
    if name in ('alexnet', 'vgg11', 'squeezenet'):
        assert _lib.OP_BIAS_RELU_BWD in kinds and _lib.OP_DROPOUT in kinds and _lib.OP_BN_BWD not in kinds
    if name == 'densenet121':
        # 58 dense layers + 3 transitions + norm5 run a BatchNorm in FRONT of their conv; all but the block-closing ones accumulate
        bnr = [k for k in range(pl.bwd.n) if pl.bwd.arr[k].kind == _lib.OP_BN_BWD and pl.bwd.arr[k].u.bn.ldx != pl.bwd.arr[k].u.bn.C]
        assert len(bnr) >= 58 - 4
        assert sum(1 for k in bnr if pl.bwd.arr[k].flags & 8) == len(bnr)
