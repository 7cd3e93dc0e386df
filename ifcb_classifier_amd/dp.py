"""Data-parallel gradient exchange: bucket planning and launch order (device agnostic, so the N>1 path is
exercised on CPU with gloo; on MI355X the same code drives RCCL over xGMI).

The reference reaches multi-GPU only through Lightning's ddp_spawn -> torch DDP (``neuston_net.py:101-107``):
gradients averaged with bucketed all-reduce overlapped with backward, per-rank BatchNorm statistics, ``--batch``
per GPU.  Here the flat gradient buffer is laid out in forward (registration) order and backward finishes it
from the tail, so every backward segment completes a contiguous tail bucket [lo, hi) that can be reduced while
the remaining segments still compute.
"""


def segment_plan(op_offsets, padded_size, total, nseg=8):
    """op_offsets[k]: flat-buffer element offsets of the parameter tensors finished by backward op k;
    padded_size[offset]: padded element count of that tensor; total: flat buffer length.
    Returns [(op_begin, op_end, lo, hi)] -- ops [op_begin, op_end) finish exactly the elements [lo, hi)."""
    target = total / float(max(1, nseg))
    segs, start, prev_lo, lo_min, produced = [], 0, total, total, 0
    n = len(op_offsets)
    for k, offs in enumerate(op_offsets):
        for off in offs:
            lo_min = min(lo_min, off)
            produced += padded_size[off]
        last = k == n - 1
        closed = produced == total - lo_min           # every tensor at or above lo_min is finished
        if last and not (closed and lo_min == 0):
            raise RuntimeError('backward op list does not cover the flat gradient buffer (covered %d of %d from %d)'
                               % (produced, total, lo_min))
        if (closed and prev_lo - lo_min >= target) or last:
            segs.append((start, k + 1, lo_min, prev_lo))
            start, prev_lo = k + 1, lo_min
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
