"""Data-parallel gradient exchange: bucket planning and launch order (device agnostic, so the N>1 path is
exercised on CPU with gloo; on MI355X the same code drives RCCL over xGMI).

The reference reaches multi-GPU only through Lightning's ddp_spawn -> torch DDP (``neuston_net.py:101-107``):
gradients averaged with bucketed all-reduce overlapped with backward, per-rank BatchNorm statistics, ``--batch``
per GPU.  Here the flat gradient buffer is laid out in forward (registration) order and backward finishes it
from the tail, so every backward segment completes a contiguous tail bucket [lo, hi) that can be reduced while
the remaining segments still compute.
"""


def segment_plan(op_offsets, padded_size, total, nseg=8, last_frac=1.0 / 24):
    """op_offsets[k]: flat-buffer element offsets of the parameter tensors finished by backward op k;
    padded_size[offset]: padded element count of that tensor; total: flat buffer length.
    Returns [(op_begin, op_end, lo, hi)] -- ops [op_begin, op_end) finish exactly the elements [lo, hi).
    Buckets are about total/nseg elements, except the LAST one: its all-reduce (and the optimizer behind it) cannot
    overlap any backward work, so the final cut is placed where at most ``last_frac`` of the buffer is left (the network's
    first layers: the stem and first blocks hold few parameters) whenever the backward order offers a cut there."""
    n = len(op_offsets)
    cuts, lo_min, produced = [], total, 0           # (ops consumed, lo) wherever the finished set is a contiguous tail
    for k, offs in enumerate(op_offsets):
        for off in offs:
            lo_min = min(lo_min, off)
            produced += padded_size[off]
        if produced == total - lo_min and (not cuts or cuts[-1][1] != lo_min):
            cuts.append((k + 1, lo_min))
    if not cuts or cuts[-1][1] != 0 or produced != total:
        raise RuntimeError('backward op list does not cover the flat gradient buffer (covered %d of %d from %d)'
                           % (produced, total, lo_min))
    cuts[-1] = (n, 0)
    target = total / float(max(1, nseg))
    small = [c for c in cuts[:-1] if 0 < c[1] <= last_frac * total]
    final_cut = max(small, key=lambda c: c[1]) if small else None
    segs, start, prev_lo = [], 0, total
    for c in cuts:
        k1, lo = c
        last = c is cuts[-1]
        if last or c is final_cut or (prev_lo - lo >= target and (final_cut is None or lo > final_cut[1])):
            segs.append((start, k1, lo, prev_lo))
            start, prev_lo = k1, lo
    return segs


def run_overlapped(segments, run_segment, grad_flat, all_reduce, mark=None):
    """launch backward segment by segment; right after a segment is enqueued, start the asynchronous
    all-reduce (sum) of the bucket it completed, then wait for all of them.  ``all_reduce(tensor)`` returns a
    work handle with ``.wait()`` (torch.distributed async_op=True) or None.  ``mark(name)``: called with
    'before_wait' once every segment and exchange is enqueued and with 'after_wait' behind the last wait -- bench.py
    records an event pair there: the time the compute stream is blocked on the exchange (its exposed part)."""
    works = []
    for seg in segments:
        run_segment(seg)
        lo, hi = seg[2], seg[3]
        if hi > lo:
            works.append(all_reduce(grad_flat[lo:hi]))
    if mark is not None:
        mark('before_wait')
    for w in works:
        if w is not None:
            w.wait()
    if mark is not None:
        mark('after_wait')
    return len(works)


EXCHANGES = ('allreduce', 'rsag')


class _Chain:
    """work handle of an exchange made of several collectives (reduce-scatter -> all-gather [+ tail all-reduce])"""

    def __init__(self, works, then=None):
        self.works, self.then = list(works), then

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        if self.then is not None:
            nxt, self.then = self.then(), None
            self.works = []
            if nxt is not None:
                nxt.wait()
        return True


def exchange_mode(mode=None):
    import os
    mode = (mode or os.environ.get('IFCBK_DP_EXCHANGE', 'allreduce')).lower()
    if mode not in EXCHANGES:
        raise ValueError('IFCBK_DP_EXCHANGE must be one of %s, got %r' % ('|'.join(EXCHANGES), mode))
    return mode


def make_exchange(dist, mode=None, group=None):
    """-> (callable(bucket) -> work handle, mode): the gradient exchange of one bucket (sum over ranks, in place).

    allreduce  one ``all_reduce`` per bucket (torch ProcessGroupNCCL = RCCL; ring-shaped by default: every byte crosses
               ONE xGMI link at a time, 2 (w-1)/w * bytes per link)
    rsag       ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the bucket, in place (rank r owns elements
               [r*c, (r+1)*c) of it, c = len // world; the < world leftover elements go through a small all_reduce):
               on the fully connected 8-GPU xGMI node each rank exchanges 1/world of the bucket with every peer at once
               (all seven links busy instead of one).  Sum order per element differs from the ring's for
               world > 2 (deterministic for a fixed world), identical at world 2.
    Both are asynchronous on the backend's own stream (RCCL) and overlap the backward segments launched behind them;
    on gloo (CPU tests) the two phases are chained at wait(): its worker threads do not order work items.
    The reference reaches this exchange through Lightning's ddp (neuston_net.py:102)."""
    mode = exchange_mode(mode)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if mode == 'allreduce' or world == 1:
        return (lambda t: dist.all_reduce(t, group=group, async_op=True)), mode
    ordered = dist.get_backend(group) != 'gloo'       # stream-ordered backends run a group's collectives in issue order

    def rsag(t):
        n = t.numel()
        c = n // world
        m = c * world
        tail = (lambda: dist.all_reduce(t[m:], group=group, async_op=True)) if m < n else (lambda: None)
        if c == 0:
            return tail()
        shard = t[rank * c:(rank + 1) * c]
        rs = dist.reduce_scatter_tensor(shard, t[:m], group=group, async_op=True)
        if ordered:
            return _Chain([rs, dist.all_gather_into_tensor(t[:m], shard, group=group, async_op=True), tail()])
        return _Chain([rs, tail()], then=lambda: dist.all_gather_into_tensor(t[:m], shard, group=group, async_op=True))
    return rsag, mode
