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


def run_overlapped(segments, run_segment, grad_flat, all_reduce):
    """launch backward segment by segment; right after a segment is enqueued, start the asynchronous
    all-reduce (sum) of the bucket it completed, then wait for all of them.  ``all_reduce(tensor)`` returns a
    work handle with ``.wait()`` (torch.distributed async_op=True) or None."""
    works = []
    for seg in segments:
        run_segment(seg)
        lo, hi = seg[2], seg[3]
        if hi > lo:
            works.append(all_reduce(grad_flat[lo:hi]))
    for w in works:
        if w is not None:
            w.wait()
    return len(works)
