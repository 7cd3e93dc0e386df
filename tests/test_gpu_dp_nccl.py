"""The data-parallel step on the REAL collective backend: torch.distributed 'nccl' (= RCCL on ROCm), world size 1 on the one
GPU a test box has.  The exchange itself is trivial at one rank, but everything around it is the production path:
ProcessGroupNCCL's stream hand-off (its collective stream waits for the caller's current stream, where every program lane
has joined), async work handles, bucket slices of the flat gradient buffer, Adam's 1/world scale, the broadcast of the
initial weights and the rank-0 gather of validation outputs (reference: ``neuston_net.py:101-107`` -> [PL] ddp_spawn).
Runs in a child process so that the process group never leaks into the other tests."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

for name, B, S in (('resnet18', 4, 224), ('inception_v3', 6, 299)):
    engs = []
    for _ in range(2):
        e = Engine(graph.build(name, 5), 0, max_batch=B)
        e.init_weights(seed=11)
        e.dropout_seed = 5
        engs.append(e)
    a, b = engs
    dist.broadcast(b.P, 0); dist.broadcast(b.RB, 0)          # Trainer.fit's start-up exchange
    b.params_changed()
    g = torch.Generator().manual_seed(3)
    calls = []

    def allreduce(t):
        calls.append(t.numel())
        return dist.all_reduce(t, async_op=True)

    for step in range(3):
        x = torch.rand(B, 3, S, S, generator=g).cuda()
        y = torch.randint(0, 5, (B,), generator=g)
        for e in engs:
            e.load_input_nchw(x)
            e.target[:B].copy_(y)
        a.train_step(B)
        b.train_step_ddp(B, 1, allreduce)
        torch.cuda.synchronize()
        assert torch.equal(a.loss, b.loss), (name, step)
        assert torch.equal(a.G, b.G), (name, step)
        assert torch.equal(a.P, b.P), (name, step)
    segs = b.ddp_segments(b.plan(B))
    assert len(calls) == 3 * len(segs) and sum(calls) == 3 * b.nparam_padded, (calls, len(segs))
    assert torch.equal(a.RB, b.RB)
    print(name, 'nccl world-1 ddp step == fused step over 3 steps;', len(segs), 'buckets', flush=True)
    del engs, a, b

# rank-0 gather of validation outputs (Trainer._gather_val) on the nccl backend
out = [None]
dist.gather_object(dict(v=torch.arange(4), s=['a', 'b']), out, dst=0)
assert out[0]['s'] == ['a', 'b'] and out[0]['v'].tolist() == [0, 1, 2, 3]
t = torch.ones(3, device='cuda')
dist.all_reduce(t)
assert t.tolist() == [1.0, 1.0, 1.0]
dist.barrier()
dist.destroy_process_group()
print('NCCL_OK', flush=True)
'''


def test_ddp_step_on_the_nccl_backend_world1(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = tmp_path / 'nccl_child.py'
    script.write_text(CHILD % dict(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    print(r.stderr[-3000:])
    assert r.returncode == 0 and 'NCCL_OK' in r.stdout
