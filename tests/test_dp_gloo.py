"""world_size-2 gloo tests (CPU) of the data-parallel path: bucket planning, the overlapped all-reduce launch
order, mean-of-gradients semantics vs a CPU emulation of "k shards, local BN, mean grads" (what the reference's
Lightning ddp computes), dataset sharding and rank-0 validation gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_segment_plan_covers_buffer_in_tail_buckets():
    from ifcb_classifier_amd.dp import segment_plan
    # 6 tensors laid out forward; backward finishes them from the tail, with one local reorder (like resnet's
    # downsample branch) and an op that finishes two tensors (BN weight+bias)
    sizes = [40, 8, 8, 100, 12, 32]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += n
    total = o
    padded = dict(zip(offs, sizes))
    ops = [[offs[5]], [], [offs[3]], [offs[4]], [offs[1], offs[2]], [offs[0]]]
    segs = segment_plan(ops, padded, total, nseg=3)
    assert segs[0][2:] == (offs[5], total) or segs[0][3] == total
    covered = []
    prev_lo = total
    for b0, b1, lo, hi in segs:
        assert hi == prev_lo and lo < hi
        prev_lo = lo
        covered.append((b0, b1))
    assert prev_lo == 0
    assert covered[0][0] == 0 and covered[-1][1] == len(ops)
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    # a cut is never placed where the finished set is not a contiguous tail (after op 2 tensor 4 is missing)
    assert all(b1 != 3 for _, b1, _, _ in segs)
    with pytest.raises(RuntimeError, match='does not cover'):
        segment_plan(ops[:-1], padded, total, nseg=3)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ifcb_classifier_amd.dp import segment_plan, run_overlapped
    from oracle import tv_models
    import torch.nn.functional as F
    torch.manual_seed(0)                                   # identical replicas
    torch.set_num_threads(2)
    model = tv_models.get_namebrand_model('resnet18', 3)
    model.train()
    g = torch.Generator().manual_seed(42)
    X = torch.rand(8, 3, 64, 64, generator=g)
    Y = torch.randint(0, 3, (8,), generator=g)
    xs, ys = X[rank::world], Y[rank::world]                # this rank's shard (per-GPU batch)
    params = list(model.parameters())
    sizes = [(p.numel() + 3) // 4 * 4 for p in params]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += n
    flat = torch.zeros(o)
    loss = F.cross_entropy(model(xs), ys)
    loss.backward()
    # "backward op" k finishes parameter len-1-k (reverse registration order), like the HIP backward list
    ops = [[offs[i]] for i in reversed(range(len(params)))]
    segs = segment_plan(ops, dict(zip(offs, sizes)), o, nseg=4)
    assert 2 <= len(segs) <= 5
    launched = []

    def run_segment(seg):
        for k in range(seg[0], seg[1]):
            i = len(params) - 1 - k
            flat[offs[i]:offs[i] + params[i].numel()] = params[i].grad.flatten()
        launched.append(seg[2:])

    n = run_overlapped(segs, run_segment, flat, lambda t: dist.all_reduce(t, async_op=True))
    assert n == len(segs)
    flat /= world                                          # Adam's grad_scale = 1/world
    # emulation: every shard on one process with local BN statistics, gradients averaged
    ref = torch.zeros(o)
    for r in range(world):
        torch.manual_seed(0)
        m2 = tv_models.get_namebrand_model('resnet18', 3)
        m2.train()
        F.cross_entropy(m2(X[r::world]), Y[r::world]).backward()
        for i, p in enumerate(m2.parameters()):
            ref[offs[i]:offs[i] + p.numel()] += p.grad.flatten() / world
    assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-7)
    # dataset sharding: ranks partition a (padded) permutation; val order is natural
    from ifcb_classifier_amd.neuston_net import ShardedLoader

    class DS:
        def __len__(self):
            return 11
    tr = ShardedLoader(DS(), 4, True, 0, rank, world, seed=5)
    tr.set_epoch(3)
    mine = tr.indices()
    allidx = [None] * world
    dist.all_gather_object(allidx, mine)
    if rank == 0:
        flat_idx = [i for part in allidx for i in part]
        assert len(flat_idx) == 12 and set(flat_idx) == set(range(11))
        assert len(allidx[0]) == len(allidx[1]) == 6
        va = ShardedLoader(DS(), 4, False, 0, 1, 2, seed=5).indices()
        assert va == [1, 3, 5, 7, 9, 0]
    # rank-0 gather of validation outputs, wrap-around duplicates dropped
    from ifcb_classifier_amd.neuston_net import Trainer
    t = Trainer.__new__(Trainer)
    t.dist, t.rank, t.world = dist, rank, world
    steps = [dict(val_batch_loss=torch.tensor(0.5 + rank), val_outputs=torch.full((3, 2), float(rank)),
                  val_input_classes=torch.tensor([rank] * 3), val_input_srcs=['img%d' % i for i in (rank, rank + 2, 4)])]
    out = t._gather_val(steps, 5)
    if rank == 0:
        srcs = [p for s in out for p in s['val_input_srcs']]
        assert srcs == ['img0', 'img2', 'img4', 'img1', 'img3']
        assert sum(len(s['val_outputs']) for s in out) == 5
    else:
        assert out == []
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_protocol_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def _worker_engine_plan(rank, world, port):
    """the REAL backward op list of inception_v3 (Engine(plan_only=True): the op tables the GPU runs, built on the host)
    drives the bucket plan; the gradients come from the CPU oracle replica on this rank's shard"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn.functional as F
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.dp import run_overlapped
    from ifcb_classifier_amd.engine import Engine
    from oracle import tv_models
    torch.set_num_threads(3)
    nc = 7
    eng = Engine(graph.build('inception_v3', nc), max_batch=2, plan_only=True)
    pl = eng.plan(2)
    segs = eng.ddp_segments(pl)
    ops = pl.bwd_list.ops
    # the segments partition the backward list in order and their buckets tile the flat buffer from the tail
    assert segs[0][0].n + sum(s[0].n for s in segs[1:]) == len(ops)
    prev_lo, done = eng.nparam_padded, 0
    for prog, b1, lo, hi in segs:
        assert hi == prev_lo and lo < hi and b1 == done + prog.n
        prev_lo, done = lo, b1
    assert prev_lo == 0 and done == len(ops)
    assert segs[-1][3] - segs[-1][2] <= eng.nparam_padded / 20      # the exposed last bucket is the small stem end
    key_at = {o: key for key, (o, n, shape, kind, node) in eng.poff.items()}
    torch.manual_seed(0)
    model = tv_models.get_namebrand_model('inception_v3', nc, storage='fp32')
    model.train()
    g = torch.Generator().manual_seed(7)
    X = torch.rand(2 * world, 3, 299, 299, generator=g)
    Y = torch.randint(0, nc, (2 * world,), generator=g)
    model.dropout_mask = torch.ones(2, 2048, dtype=torch.bool)

    def grads_of(shard):
        model.zero_grad()
        out = model(X[shard::world])
        (F.cross_entropy(out.logits, Y[shard::world]) + 0.4 * F.cross_entropy(out.aux_logits, Y[shard::world])).backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    mine = grads_of(rank)
    filled = set()

    def run_segment(seg):
        b0 = seg[1] - seg[0].n
        for k in range(b0, seg[1]):
            for off in eng._op_param_offsets(ops[k]):
                key = key_at[off]
                eng.gviews[key].copy_(mine[key])              # what this backward op leaves in the flat gradient buffer
                filled.add(key)

    n = run_overlapped(segs, run_segment, eng.G, lambda t: dist.all_reduce(t, async_op=True))
    assert n == len(segs) and filled == set(mine)
    # emulation of "k shards, local BN statistics, mean of the gradients" (what the reference's ddp computes)
    ref = {k: v / world for k, v in grads_of(0).items()}
    for r in range(1, world):
        for k, v in grads_of(r).items():
            ref[k] += v / world
    for key in ref:
        assert torch.allclose(eng.gviews[key] / world, ref[key], rtol=1e-5, atol=1e-8), key
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_buckets_from_the_engines_own_backward_list_world2_gloo():
    port = _free_port()
    mp.spawn(_worker_engine_plan, args=(2, port), nprocs=2, join=True)


def _worker_exchange(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ifcb_classifier_amd.dp import make_exchange, run_overlapped
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [4, 1, 7, 4096 + 3, 2, 100001]            # buckets with and without a leftover (len % world != 0), one shorter than world
    total = sum(sizes)
    base = torch.randn(total, generator=g)
    res = {}
    for mode in ('allreduce', 'rsag'):
        ex, name = make_exchange(dist, mode)
        assert name == mode
        flat = base.clone()
        segs, hi = [], total
        for k, n in enumerate(sizes):
            segs.append((k, k + 1, hi - n, hi))
            hi -= n
        marks = []
        n = run_overlapped(segs, lambda seg: None, flat, ex, mark=marks.append)
        assert n == len(sizes) and marks == ['before_wait', 'after_wait']
        res[mode] = flat
    # reduce-scatter + all-gather is the same sum as the all-reduce: bit-equal at world 2 (one addition per element)
    assert torch.equal(res['allreduce'], res['rsag'])
    both = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(both, base)
    assert torch.equal(res['rsag'], both[0] + both[1])
    with pytest.raises(ValueError):
        make_exchange(dist, 'ring-of-fire')
    dist.barrier()
    dist.destroy_process_group()


def test_rsag_exchange_equals_allreduce_world2_gloo():
    """IFCBK_DP_EXCHANGE=rsag (reduce-scatter + all-gather per bucket) against all_reduce on the same buckets"""
    port = _free_port()
    mp.spawn(_worker_exchange, args=(2, port), nprocs=2, join=True)
