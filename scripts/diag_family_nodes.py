"""dev tool: node-local parity table of one backbone family (tests/local_parity.py, verbose), e.g. densenet121 fp32"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import torch.nn.functional as F
import test_gpu_families as T
from local_parity import check_plan
from oracle import ops as O
name, dtype = sys.argv[1], sys.argv[2]
nc, B = 5, 3
hip, ora = T._pair(name, nc, B, dtype)
g = torch.Generator().manual_seed(5)
x = torch.rand(B, 3, 224, 224, generator=g)
y = torch.randint(0, nc, (B,), generator=g)
mo, mh = T._masks(name, B, g)
if mh is not None:
    hip.set_dropout_mask(mh)
hip.train()
F.cross_entropy(hip(x.cuda()), y.cuda()).backward()
torch.cuda.synchronize()
O.set_storage(dtype)
check_plan(hip, B, verbose=True)
