"""The lane scheduler (engine.schedule_lanes) must order every pair of conflicting ops: read-after-write, write-after-read
and write-after-write on overlapping channel ranges, plus full barriers.  Checked on random programs by replaying the
runner's semantics (ctx.hip::run_lanes): an op starts after the previous op of its lane and after everything queued so
far on the lanes of its wait mask."""
import random

from ifcb_classifier_amd.engine import schedule_lanes, _overlap


def _happens_before(sched):
    n = len(sched)
    last_on_lane = {}
    preds = [set() for _ in range(n)]
    queued = {l: [] for l in range(8)}
    for i, (lane, wait) in enumerate(sched):
        if lane in last_on_lane:
            preds[i].add(last_on_lane[lane])
        for l in range(8):
            if wait >> l & 1 and queued[l]:
                preds[i].add(queued[l][-1])          # the tail of lane l (stream order covers everything before it)
        last_on_lane[lane] = i
        queued[lane].append(i)
    # transitive closure (small n)
    reach = [set() for _ in range(n)]
    for i in range(n):
        for p in preds[i]:
            reach[i].add(p)
            reach[i] |= reach[p]
    return reach


def _random_program(rng, n, nl):
    meta = []
    for _ in range(n):
        if rng.random() < 0.06:
            meta.append((0, None, None))             # barrier op
            continue
        lane = rng.randrange(nl)

        def res():
            key = rng.choice(['a', 'g', 'r'])
            buf = rng.randrange(4)
            lo = rng.choice([0, 0, 64, 128])
            hi = lo + rng.choice([64, 128, 256])
            return (key, buf, lo, hi)
        reads = [res() for _ in range(rng.randrange(0, 3))]
        writes = [res() for _ in range(rng.randrange(0, 2))]
        meta.append((lane, reads, writes))
    return meta


def test_schedule_orders_every_conflict():
    rng = random.Random(1234)
    for trial in range(60):
        nl = rng.choice([1, 2, 3, 4, 6, 8])
        meta = _random_program(rng, rng.randrange(5, 70), nl)
        sched = schedule_lanes(meta)
        assert len(sched) == len(meta)
        reach = _happens_before(sched)
        for j, (lj, rj, wj) in enumerate(meta):
            for i in range(j):
                li, ri, wi = meta[i]
                barrier = (ri is None and wi is None) or (rj is None and wj is None)
                conflict = barrier
                if not barrier:
                    for w in wi:
                        conflict |= any(_overlap(w, x) for x in rj) or any(_overlap(w, x) for x in wj)
                    for w in wj:
                        conflict |= any(_overlap(w, x) for x in ri)
                if conflict:
                    assert i in reach[j], (trial, i, j, meta[i], meta[j], sched[i], sched[j])


def test_single_lane_program_has_no_waits():
    meta = [(0, [('a', 0, 0, 8)], [('a', 1, 0, 8)]), (0, [('a', 1, 0, 8)], [('a', 2, 0, 8)]), (0, None, None)]
    assert schedule_lanes(meta) == [(0, 0), (0, 0), (0, 0)]


def test_weight_gradients_have_their_own_lane_and_wait_for_their_operands():
    """round 4: in the REAL backward list of inception_v3 (Engine(plan_only=True)) every weight gradient -- single launches, the
    sibling GEMMs' segmented ones, the grouped launches -- sits on the last lane, every layer's d(raw) lives in a buffer of its own,
    and the schedule orders each weight gradient behind the BatchNorm backward that writes its dy (and a grouped one behind ALL
    its members').  ref neuston_models.py:81-86: autograd computes the same gradients wherever it likes before optimizer.step()."""
    from ifcb_classifier_amd import graph, _lib
    from ifcb_classifier_amd.engine import Engine
    eng = Engine(graph.build('inception_v3', 7), max_batch=2, plan_only=True)
    assert eng.NL == 4 and eng.wgrad_lane == 1
    pl = eng.plan(2)
    ops, meta = pl.bwd_list.ops, pl.bwd_list.meta
    sched = schedule_lanes(meta)
    reach = _happens_before(sched)
    wkinds = (_lib.OP_CONV_WGRAD, _lib.OP_CONV_WGRAD_SEG, _lib.OP_CONV_WGRAD_GROUP, _lib.OP_STEM_U8_WGRAD)
    nw = 0
    for j, o in enumerate(ops):
        if o.kind in wkinds:
            nw += 1
            assert sched[j][0] == eng.NL - 1, pl.bwd_list.tags[j]
            # every d(raw) resource it reads was written by an earlier op that happens-before it
            for r in meta[j][1]:
                if r[0] in ('drawn', 'dg'):
                    writers = [i for i in range(j) if meta[i][2] and any(_overlap(w, r) for w in meta[i][2])]
                    assert writers, (pl.bwd_list.tags[j], r)
                    assert all(i in reach[j] for i in writers), (pl.bwd_list.tags[j], r)
        else:
            assert sched[j][0] != eng.NL - 1 or o.kind == 0, pl.bwd_list.tags[j]
    assert nw >= 40
    # no two layers share a d(raw) buffer any more
    ptrs = [t.data_ptr() for t in eng.draw_own.values()]
    assert len(ptrs) == len(set(ptrs)) and len(ptrs) >= 60          # (the sibling GEMMs' members share their group's merged tensor)


def test_plan_with_the_side_lane_switch_builds(monkeypatch):
    """IFCBK_WGRAD_SIDE=1 (still documented in scripts/README.md) raised AttributeError in plan() -- the lane expression
    read an attribute an earlier diff had removed"""
    from ifcb_classifier_amd import graph, _lib
    from ifcb_classifier_amd.engine import Engine
    monkeypatch.setenv('IFCBK_WGRAD_SIDE', '1')
    eng = Engine(graph.build('inception_v3', 7), max_batch=2, plan_only=True)
    assert eng.wgrad_side_lane and eng.wgrad_lane == 0
    pl = eng.plan(2)
    ops, meta = pl.bwd_list.ops, pl.bwd_list.meta
    sched = schedule_lanes(meta)
    lanes = {sched[j][0] for j, o in enumerate(ops) if o.kind == _lib.OP_CONV_WGRAD}
    assert len(lanes) > 1          # the weight gradients ride on their layers' NEIGHBOURING lanes, not on one lane of their own


def test_program_lane_default_follows_the_job_size_not_the_construction_order(monkeypatch):
    """an Engine built before init_process_group silently took the single-GPU lane count.  The world size now comes from
    the argument or the launcher's WORLD_SIZE, and train_step_ddp refuses a world the lanes were not chosen for."""
    import pytest
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    monkeypatch.delenv('IFCBK_LANES', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    net = graph.build('resnet18', 2)
    assert Engine(net, max_batch=2, plan_only=True).NL == 4
    assert Engine(net, max_batch=2, plan_only=True, dp_world=8).NL == 2
    monkeypatch.setenv('WORLD_SIZE', '8')
    e = Engine(net, max_batch=2, plan_only=True)
    assert e.NL == 2 and e.dp_world == 8 and e.wgrad_lane == 1
    monkeypatch.delenv('WORLD_SIZE')
    e1 = Engine(net, max_batch=2, plan_only=True)
    with pytest.raises(RuntimeError, match='dp_world'):
        e1.train_step_ddp(2, 8, lambda *a, **k: None)


def test_bucketed_optimizer_waits_for_its_gradients_and_the_repack_for_the_readers_of_the_shadows():
    """round 5: the fused train step runs the optimizer and the bf16 repack PER BUCKET of the flat gradient buffer, on the
    weight-gradient lane, while backward is still running (engine._bucketed_update).  On the frozen step program of inception_v3
    (lanes and wait masks as ctx.hip::run_lanes will see them): the buckets tile the parameter buffer; every op that writes a
    gradient happens-before the optimizer launch of its bucket; a bucket's repack happens-after its optimizer launch AND after
    every input-gradient op that still reads the bf16 shadows it rewrites.  ref neuston_models.py:63-64,81-86: optimizer.step()
    after loss.backward() -- per parameter that order is all the reference's arithmetic depends on."""
    from ifcb_classifier_amd import graph, _lib
    from ifcb_classifier_amd.engine import Engine
    eng = Engine(graph.build('inception_v3', 7), max_batch=2, plan_only=True)
    pl = eng.plan(2)
    arr, n, tags = pl.step.arr, pl.step.n, pl.step.tags
    sched = [((arr[k].flags >> 8) & 7, (arr[k].flags >> 12) & 0xff) for k in range(n)]
    reach = _happens_before(sched)
    base = eng.G.data_ptr()
    adam = pl.step_adam_idxs
    assert len(adam) >= 2 and adam == sorted(adam)
    rng = [((arr[k].p[1] - base) // 4, int(arr[k].i[0])) for k in adam]
    lo_sorted = sorted(rng)
    assert lo_sorted[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(lo_sorted, lo_sorted[1:]))
    assert lo_sorted[-1][0] + lo_sorted[-1][1] == eng.nparam_padded

    def bucket_of(off):
        for b, (lo, cnt) in enumerate(rng):
            if lo <= off < lo + cnt:
                return b
        raise AssertionError(off)
    # ---- every gradient writer happens-before its bucket's optimizer launch
    nwriters, off_of_layer = 0, {}
    for j in range(n):
        offs = eng._op_param_offsets(arr[j])
        for off in offs:
            b = bucket_of(off)
            assert j < adam[b] and j in reach[adam[b]], (tags[j], b)
            nwriters += 1
        # conv weight gradients: remember which parameter offset belongs to which layer name
        k = arr[j].kind
        if k == _lib.OP_CONV_WGRAD or k == _lib.OP_STEM_U8_WGRAD:
            off_of_layer[tags[j]] = offs[0]
        elif k in (_lib.OP_CONV_WGRAD_SEG, _lib.OP_CONV_WGRAD_GROUP):
            for name, off in zip(tags[j].split('+'), offs):
                off_of_layer[name] = off
    assert nwriters > 250 and len(off_of_layer) >= 94
    # ---- the repack of bucket b: after its optimizer launch, after every reader of the shadows it rewrites
    packs = pl.step.find(_lib.OP_WEIGHT_PACK_MULTI)
    assert len(packs) == len(adam)
    for b, (a, pk) in enumerate(zip(adam, packs)):
        assert a < pk and a in reach[pk], b
    dkinds = (_lib.OP_CONV_DGRAD, _lib.OP_CONV_DGRAD_BNSTAT, _lib.OP_CONV_DGRAD_BNSTAT_TAB)
    nreaders = 0
    for j in range(n):
        if arr[j].kind in dkinds:
            for name in tags[j].split('+'):
                pk = packs[bucket_of(off_of_layer[name])]
                assert j < pk and j in reach[pk], (tags[j], name)
                nreaders += 1
    assert nreaders >= 90
