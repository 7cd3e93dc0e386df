import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import test_gpu_families as T
name, nc, B = sys.argv[1], 4, 2
dtype = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
hip, ora, x, lh, lo, loss_h, loss_o = T._step(name, nc, B, dtype)
errs, dead = T._grad_errors(hip, ora)
for k, v in errs.items():
    print('%-40s %.2e' % (k, v))
