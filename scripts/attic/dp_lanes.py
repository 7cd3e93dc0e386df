"""dev tool: the data-parallel step (train_step_ddp: bucketed async all-reduces on the RCCL stream beside the backward lanes) on the
real `nccl` backend at world size 1, batch 256, with 2, 3 and 4 program lanes -- the measurement behind the DP lane default
(VERDICT r2 item 6).  One process, one GPU:
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 python scripts/dp_lanes.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

B = int(os.environ.get('DP_BATCH', 256))


def allred(t):
    return dist.all_reduce(t, async_op=True)


res = {}
for lanes in ('2', '3', '4'):
    os.environ['IFCBK_LANES'] = lanes
    eng = Engine(graph.build('inception_v3', 100), 0, max_batch=B)
    eng.init_weights(seed=1)
    x = torch.rand(B, 3, 299, 299, device='cuda')
    eng.target[:B].copy_(torch.randint(0, 100, (B,)))
    eng.load_input_nchw(x)
    for mode in ('fused', 'ddp'):
        step = (lambda: eng.train_step(B)) if mode == 'fused' else (lambda: eng.train_step_ddp(B, 1, allred))
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        per = []
        for _ in range(20):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            per.append(1e3 * (time.perf_counter() - t0))
        per.sort()
        res[(lanes, mode)] = (sum(per) / len(per), per[0], per[-1])
        print('lanes %s %-5s: mean %.2f ms  min %.2f  max %.2f  (20 steps, each synchronised)' % (lanes, mode, *res[(lanes, mode)]), flush=True)
    del eng
    torch.cuda.empty_cache()
dist.barrier()
dist.destroy_process_group()
